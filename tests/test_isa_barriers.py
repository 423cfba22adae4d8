"""Generated-code check (CPU only: hipcc cross-compiles): the chunk hand-over barriers of the linear-stream resize kernels wait for
the wave's own LDS-DMA (tools/check_isa_barriers.py has the story)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_every_hand_over_barrier_waits_for_the_dma():
    spec = importlib.util.spec_from_file_location("check_isa_barriers", os.path.join(ROOT, "tools", "check_isa_barriers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen, bad = mod.check(mod.assembly())
    assert seen >= 100, seen  # chunk stream 6 + K-split 6 + cropped stream 4 + search 21 instantiations, two or three barriers each
    assert not bad, bad


def test_the_checker_sees_a_missing_wait():
    spec = importlib.util.spec_from_file_location("check_isa_barriers", os.path.join(ROOT, "tools", "check_isa_barriers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    good = """_ZN3vdf31resize_mfma_frame_stream_kernelILi1EEEvv:
\ts_waitcnt vmcnt(0)
\ts_barrier
.LBB0_1:
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\ts_barrier
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:
"""
    assert mod.check(good) == (1, [])
    bad = good.replace("\ts_waitcnt vmcnt(0)\n", "\ts_waitcnt lgkmcnt(0)\n", 1)
    seen, missing = mod.check(bad)
    assert seen == 1 and len(missing) == 1
    other = good.replace("resize_mfma_frame_stream_kernel", "dct_hash_kernel").replace("\ts_waitcnt vmcnt(0)\n", "")
    assert mod.check(other) == (0, [])
