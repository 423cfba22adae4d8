#!/usr/bin/env python3
"""Every chunk hand-over barrier of the linear-stream resize kernels must be preceded by `s_waitcnt vmcnt(0)`.

The kernels that hand a DMA-filled LDS buffer from four issuing waves to four consuming waves (resize_mfma_frame_stream_kernel,
resize_mfma_frame_ksplit_kernel, resize_mfma_cropped_stream_kernel) need each wave's own LDS-DMA instructions to have landed before it
arrives at the barrier.  The compiler once dropped the wait that the fence of __syncthreads() used to bring along (round 3: wrong hashes
in one instantiation), so the wait is explicit in the source - and this script checks the generated code of every instantiation:
compile csrc/dct_hash.hip, csrc/hamming.hip (whose matrix-core search kernel hands candidate stages over the same way) and
csrc/cropdetect.hip (for the spill report) to gfx950 assembly, walk each of those kernels, and for every `s_barrier` that is not one of the hand-written
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
           "hamming_mfma2_kernel")  # the search kernel hands its candidate stages over the same way
# The per-wave stream kernel has no workgroup hand-over: every wave waits for its OWN block (an inline-asm `s_waitcnt vmcnt(0)` that also
# reads the two vertical fragments requested behind the DMA) right before the products that read the block from LDS.
WAVE_KERNELS = ("resize_mfma_frame_wavestream_kernel",)


def assembly(path=None):
    if path:
        return open(path).read()
    csrc = os.path.join(ROOT, "vid_dup_finder_lib_amd", "csrc")
    text = ""
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("dct_hash.hip", "hamming.hip", "cropdetect.hip", "sort_order.hip"):
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


def check_wave_waits(text):
    """-> (instantiations seen, [(kernel symbol, why)]): every instantiation of the per-wave stream kernel must hold the explicit
    `s_waitcnt vmcnt(0)` in inline asm, and no LDS-DMA may sit between that wait and the first matrix product behind it."""
    lines = text.split("\n")
    kernel, in_asm, seen, bad = None, False, 0, []
    waits, armed = 0, False
    for raw in lines:
        line = raw.strip()
        m = re.match(r"^(_ZN3vdf\w+):", raw)
        if m:
            kernel = m.group(1) if any(k in m.group(1) for k in WAVE_KERNELS) else None
            waits, armed = 0, False
            if kernel:
                seen += 1
            continue
        if kernel is None:
            continue
        if line.startswith(".Lfunc_end"):
            if waits == 0:
                bad.append((kernel, "no explicit s_waitcnt vmcnt(0) in inline asm"))
            kernel = None
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm and re.match(r"s_waitcnt\s+vmcnt\(0\)\s*$", line):
            waits += 1
            armed = True
            continue
        if armed and re.match(r"buffer_load_dword\w*\s.*\blds\b", line):
            bad.append((kernel, "an LDS-DMA between the block wait and the products that read the block"))
            armed = False
        if armed and line.startswith("v_mfma"):
            armed = False
    return seen, bad


def spills(text):
    """-> {kernel symbol: (vgpr spills, sgpr spills, scratch bytes)} from the code object metadata of the assembly (what
    `llvm-readelf --notes` shows for the shipped library): a spilled register in a stream kernel is a scratch round trip per use."""
    out, name = {}, None
    vg = sg = sc = 0
    for raw in text.split("\n"):
        line = raw.strip()
        m = re.match(r"^\.name:\s+(_ZN3vdf\w+)$", line)
        if m:
            name = m.group(1)
        m = re.match(r"^\.private_segment_fixed_size:\s+(\d+)$", line)
        if m:
            sc = int(m.group(1))
        m = re.match(r"^\.sgpr_spill_count:\s+(\d+)$", line)
        if m:
            sg = int(m.group(1))
        m = re.match(r"^\.vgpr_spill_count:\s+(\d+)$", line)
        if m:
            vg = int(m.group(1))
        if line.startswith("- .") or line.startswith("- .args") or line == "...":
            pass
        if line.startswith(".wavefront_size:") and name:  # last key of a kernel's record (keys are sorted)
            out[name] = (vg, sg, sc)
            name, vg, sg, sc = None, 0, 0, 0
    return out


if __name__ == "__main__":
    text = assembly(sys.argv[1] if len(sys.argv) > 1 else None)
    n, bad = check(text)
    for k, no in bad:
        print(f"hand-over barrier without s_waitcnt vmcnt(0): {k} (line {no})")
    print(f"{n} hand-over barriers checked, {len(bad)} without the wait")
    nw, badw = check_wave_waits(text)
    for k, why in badw:
        print(f"per-wave stream kernel: {why}: {k}")
    print(f"{nw} per-wave stream instantiations checked, {len(badw)} bad")
    sp = {k: v for k, v in spills(text).items() if v[0] or v[2]}
    for k, v in sorted(sp.items()):
        print(f"spills: {k}: {v[0]} VGPRs, {v[1]} SGPRs, scratch {v[2]} B")
    sys.exit(1 if bad or n == 0 or badw or nw == 0 else 0)
