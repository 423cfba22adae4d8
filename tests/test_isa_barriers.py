"""Generated-code checks (CPU only: hipcc cross-compiles the kernels to gfx950 assembly once per session; tools/check_isa_barriers.py has the
stories): the chunk hand-over barriers of the linear-stream resize kernels and of the search kernel wait for the wave's own LDS-DMA; every
instantiation of the per-wave stream kernel holds its explicit block wait; no shipped stream kernel spills registers to scratch."""
import importlib.util
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_hipcc = pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")


def _mod():
    spec = importlib.util.spec_from_file_location("check_isa_barriers", os.path.join(ROOT, "tools", "check_isa_barriers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def isa():
    mod = _mod()
    return mod, mod.assembly()


@needs_hipcc
def test_every_hand_over_barrier_waits_for_the_dma(isa):
    mod, text = isa
    seen, bad = mod.check(text)
    assert seen >= 90, seen  # chunk stream 6 + K-split 6 + cropped stream 4 + search 16 instantiations, two or three barriers each
    assert not bad, bad


@needs_hipcc
def test_every_per_wave_stream_instantiation_waits_for_its_own_block(isa):
    mod, text = isa
    seen, bad = mod.check_wave_waits(text)
    assert seen >= 30, seen  # 5 wave counts x 3 addressing modes x plain / ROWCROP
    assert not bad, bad


@needs_hipcc
def test_every_lds_reuse_keeps_its_barriers(isa):
    """LDS buffers re-used across the iterations of a persistent loop are ordered by barriers; every instantiation of every
    such kernel still holds them in the generated code (round 5's race was one missing LDS-only barrier in two kernel families)."""
    mod, text = isa
    seen, bad = mod.check_reuse_barriers(text)
    assert seen >= 85, seen
    assert not bad, bad


@needs_hipcc
def test_the_fused_letterbox_kernel_waits_for_its_table_dma(isa):
    mod, text = isa
    seen, bad = mod.check_fused_letterbox(text)
    assert seen == 1 and not bad, bad
    good = """_ZN3vdf38letterbox_resize_dct_hash_small_kernelEPKhjj:
\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tds_write_b32 v1, v2
\ts_barrier
\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds
\tds_read_b128 v[4:7], v3
\ts_endpgm
.Lfunc_end0:
"""
    assert mod.check_fused_letterbox(good) == (1, [])
    assert len(mod.check_fused_letterbox(good.replace("\ts_waitcnt vmcnt(0)\n", "\ts_nop 0\n"))[1]) == 1
    moved = good.replace("\tds_write_b32 v1, v2\n", "\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds\n")  # a DMA between wait and barrier
    assert len(mod.check_fused_letterbox(moved)[1]) == 1


def test_the_reuse_check_sees_a_missing_barrier():
    mod = _mod()
    good = """_ZN3vdf31resize_mfma_frame_stream_kernelILi1ELi8ELi0ELb0EEEvv:
\ts_waitcnt vmcnt(0)
\ts_barrier
\ts_barrier
\ts_barrier
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\ts_barrier
\t;;#ASMEND
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\ts_barrier
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:
"""
    assert mod.check_reuse_barriers(good) == (1, [])
    one_gone = good.replace("\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\t;;#ASMEND\n", "", 1)  # the library before ce37e43
    seen, bad = mod.check_reuse_barriers(one_gone)
    assert seen == 1 and len(bad) == 1 and bad[0][1:] == (3, 1, (3, 2))
    assert len(mod.check_reuse_barriers(good.replace("\ts_barrier\n", "", 1))[1]) == 1  # a plain __syncthreads() gone
    other = good.replace("resize_mfma_frame_stream_kernel", "some_other_kernel_of_31_characters"[:31])
    assert mod.check_reuse_barriers(other) == (0, [])


@needs_hipcc
def test_no_stream_kernel_spills(isa):
    """`.vgpr_spill_count` / scratch size of every kernel of the three kernel files, as `llvm-readelf --notes` shows them for the shipped
    code objects: the search kernels of tolerances above 0.357 (K = 14 ... 16 k-steps) spilled 6 - 9 VGPRs into their MFMA stream until
    round 4 moved their row-term vectors to LDS."""
    mod, text = isa
    sp = mod.spills(text)
    search = {k: v for k, v in sp.items() if "hamming_" in k or "resolve_candidates" in k or "expand_fp4" in k}
    assert len([k for k in search if "hamming_mfma2_kernel" in k]) == 16  # CHK 6, 8, 10 .. 14, 16 x 8 / 4 waves
    assert all(v == (0, 0, 0) for v in search.values()), {k: v for k, v in search.items() if v != (0, 0, 0)}
    resize = {k: v for k, v in sp.items() if re.search(r"resize_|dct_hash", k)}
    assert len(resize) >= 50 and all(v[0] == 0 and v[2] == 0 for v in resize.values()), {k: v for k, v in resize.items() if v[0] or v[2]}
    other = {k: v for k, v in sp.items() if (v[0] or v[2]) and k not in search and k not in resize}
    # the letterbox detect: letterbox_kernel (eight waves per SIMD, 64 registers) parked 4 values of its four-row walk in scratch until round 5
    # made its wave index wave-uniform for the compiler (readfirstlane): the per-wave LDS addresses moved to SGPRs
    detect = {k: v for k, v in sp.items() if "letterbox_" in k and "resize" not in k}  # (the fused small-frame kernel counts as a resize kernel)
    assert len(detect) == 4 and all(v[0] == 0 and v[2] == 0 for v in detect.values()), detect
    assert not other, other
    # round 6: the device-side sorts are hand-written (csrc/sort_order.hip) - rocPRIM's radix sort was the only scratch user of the library
    sorts = {k: v for k, v in sp.items() if "radix_" in k or "sort_keys" in k or "gather_hashes" in k}
    assert len(sorts) >= 7 and all(v == (0, 0, 0) for v in sorts.values()), sorts  # keys, gather, histogram x 2 key types, onesweep x 3
    assert all(v[2] == 0 for v in sp.values()), {k: v for k, v in sp.items() if v[2]}  # no kernel of the library uses scratch memory


def test_the_checker_sees_a_missing_wait():
    mod = _mod()
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


def test_the_checker_sees_a_missing_block_wait_and_a_spill():
    mod = _mod()
    good = """_ZN3vdf35resize_mfma_frame_wavestream_kernelILi4EEEvv:
\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds
\tglobal_load_dwordx4 v[2:5], v[6:7], off
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tds_read_b128 v[8:11], v12
\tv_mfma_i32_16x16x64_i8 v[0:3], v[8:11], v[8:11], v[0:3]
\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds
\ts_endpgm
.Lfunc_end0:
"""
    assert mod.check_wave_waits(good) == (1, [])
    assert len(mod.check_wave_waits(good.replace("\ts_waitcnt vmcnt(0)\n", "\ts_nop 0\n"))[1]) == 1
    moved = good.replace("\t;;#ASMEND\n", "\t;;#ASMEND\n\tbuffer_load_dwordx4 v1, s[0:3], 0 offen lds\n", 1)
    assert len(mod.check_wave_waits(moved)[1]) == 1
    meta = """  - .agpr_count:     0
    .name:           _ZN3vdf20hamming_mfma2_kernelILi16ELi8EEEvv
    .private_segment_fixed_size: 40
    .sgpr_spill_count: 0
    .vgpr_count:     256
    .vgpr_spill_count: 9
    .wavefront_size: 64
"""
    assert mod.spills(meta) == {"_ZN3vdf20hamming_mfma2_kernelILi16ELi8EEEvv": (9, 0, 40)}
