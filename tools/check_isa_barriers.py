#!/usr/bin/env python3
"""Every chunk hand-over barrier of the linear-stream resize kernels must be preceded by `s_waitcnt vmcnt(0)`.

The kernels that hand a DMA-filled LDS buffer from four issuing waves to four consuming waves (resize_mfma_frame_stream_kernel,
resize_mfma_frame_ksplit_kernel, resize_mfma_cropped_stream_kernel) need each wave's own LDS-DMA instructions to have landed before it
arrives at the barrier.  The compiler once dropped the wait that the fence of __syncthreads() used to bring along (round 3: wrong hashes
in one instantiation), so the wait is explicit in the source - and this script checks the generated code of every instantiation:
compile csrc/dct_hash.hip and csrc/hamming.hip (whose matrix-core search kernels hand candidate stages over the same way) to gfx950
assembly, walk each of those kernels, and for every `s_barrier` that is not one of the hand-written
LDS-only barriers (`s_waitcnt lgkmcnt(0)` + `s_barrier` inside one inline-asm block) require an `s_waitcnt vmcnt(0)` among the
instructions between the last label and the barrier.

    python tools/check_isa_barriers.py [path/to/kernels.s]      exit code 0 = every hand-over barrier waits
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("resize_mfma_frame_stream_kernel", "resize_mfma_frame_ksplit_kernel", "resize_mfma_cropped_stream_kernel",
           "hamming_mfma2_kernel", "hamming_mfma_kernel")  # the search kernels hand their candidate stages over the same way


def assembly(path=None):
    if path:
        return open(path).read()
    csrc = os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc")
    text = ""
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("dct_hash.hip", "hamming.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S",
                            "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), os.path.join(csrc, src), "-o", out],
                           check=True, stderr=subprocess.DEVNULL)
            text += open(out).read() + "\n"
    return text


def check(text):
    """-> (number of hand-over barriers seen, list of (kernel symbol, line number) without the wait)"""
    lines = text.split("\n")
    kernel, in_asm, asm_has_lgkm, seen, bad = None, False, False, 0, []
    since_label = []
    for no, raw in enumerate(lines, 1):
        line = raw.strip()
        m = re.match(r"^(_ZN3vdf\w+):", raw)
        if m:
            kernel = m.group(1) if any(k in m.group(1) for k in KERNELS) else None
            since_label = []
            continue
        if kernel is None:
            continue
        if line.startswith(".Lfunc_end"):
            kernel = None
            continue
        if re.match(r"^\.LBB\w+:", line):
            since_label = []
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm, asm_has_lgkm = True, False
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";"):
            continue
        if in_asm and "lgkmcnt(0)" in line and "vmcnt" not in line:
            asm_has_lgkm = True
        if line.startswith("s_barrier"):
            if in_asm and asm_has_lgkm:
                continue  # the LDS-only barrier of the K-split reduction / frame end
            seen += 1
            if not any("vmcnt(0)" in x for x in since_label):
                bad.append((kernel, no))
        since_label.append(line)
    return seen, bad


if __name__ == "__main__":
    n, bad = check(assembly(sys.argv[1] if len(sys.argv) > 1 else None))
    for k, no in bad:
        print(f"hand-over barrier without s_waitcnt vmcnt(0): {k} (line {no})")
    print(f"{n} hand-over barriers checked, {len(bad)} without the wait")
    sys.exit(1 if bad or n == 0 else 0)
