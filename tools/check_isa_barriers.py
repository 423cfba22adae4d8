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


# LDS buffers that live across iterations of a persistent loop (or across the passes of the DCT) are ordered by workgroup barriers; DESIGN.md
# section 4.4 has the table (buffer, writer, reader, the barrier in the source).  A barrier that disappears from the source - or from the
# generated code - shows up here as a count below what every instantiation of the kernel family has today:
#   family: (plain s_barrier instructions at least, hand-written LDS-only barriers `s_waitcnt lgkmcnt(0); s_barrier` exactly)
REUSE_BARRIERS = {
    "dct_hash_kernel": (5, 0),                                # b / c overlay, words: dct_hash_block's four + the cube's
    "resize_dct_hash_fused_kernel": (5, 0),
    "resize_dct_hash_persistent_kernel": (5, 0),              # sh.cube / sh.b|c / sh.words across clips
    "resize_dct_hash_tiled_kernel": (5, 0),
    "resize_dct_hash_cropped_small_kernel": (10, 0),          # two code paths x 5
    "letterbox_resize_dct_hash_small_kernel": (9, 0),         # prologue 2 + per clip: probes / tables hand-over, detect, cube, DCT 4
    "resize_mfma_frame_stream_kernel": (3, 2),                # chunk hand-over, frame end; LDS-only: K... one-chunk frame (round 5's race), reduction
    "resize_mfma_frame_ksplit_kernel": (3, 2),
    "resize_mfma_cropped_stream_kernel": (5, 2),
    "resize_mfma_frame_wavestream_kernel": (1, 1),
    "hamming_mfma2_kernel": (3, 0),                           # candidate stages
    "letterbox_kernel": (2, 0),
    "letterbox_sides_kernel": (3, 0),                         # s_edge / s_prog across work-list entries
    "radix_onesweep_kernel": (6, 0),                          # ticket, uniform-digit vote, per-wave counts, digit scan (2), bases
}


def check_reuse_barriers(text):
    """-> (kernels checked, list of (kernel symbol, plain barriers, LDS-only barriers, expected)) for every instantiation of REUSE_BARRIERS"""
    counts, kernel, in_asm, asm_lgkm = {}, None, False, False
    for raw in text.split("\n"):
        m = re.match(r"^(_ZN3vdf\w+):", raw)
        if m:
            kernel = m.group(1)
            counts[kernel] = [0, 0]
            continue
        if kernel is None:
            continue
        line = raw.strip()
        if line.startswith(".Lfunc_end"):
            kernel = None
        elif line.startswith(";;#ASMSTART"):
            in_asm, asm_lgkm = True, False
        elif line.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and "lgkmcnt(0)" in line and "vmcnt" not in line:
            asm_lgkm = True
        elif line.startswith("s_barrier"):
            counts[kernel][1 if (in_asm and asm_lgkm) else 0] += 1
    seen, bad = 0, []
    for sym, (plain, lds_only) in counts.items():
        for fam, (want_plain, want_lds) in REUSE_BARRIERS.items():
            if re.search(r"\d+" + fam + r"(I|E)", sym):  # the family's own name, length-prefixed in the mangled symbol
                seen += 1
                if plain < want_plain or lds_only != want_lds:
                    bad.append((sym, plain, lds_only, (want_plain, want_lds)))
    return seen, bad


def check_fused_letterbox(text):
    """letterbox_resize_dct_hash_small_kernel hands its box tables over by LDS-DMA (every wave brings a quarter of them): at the top of the
    persistent loop every wave must wait for its own DMA (`s_waitcnt vmcnt(0)`, hand-written: the compiler knows nothing of the hand-over)
    BEFORE the workgroup barrier behind which the tables are read, and no new DMA may be issued in between.
    -> (kernels seen, list of reasons)"""
    seen, bad = 0, []
    for m in re.finditer(r"^(_ZN3vdf\d+letterbox_resize_dct_hash_small_kernel\w*):", text, flags=re.M):
        seen += 1
        body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
        events, in_asm = [], False
        for raw in body:
            line = raw.strip()
            if line.startswith(";;#ASMSTART"):
                in_asm = True
            elif line.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and "vmcnt(0)" in line:
                events.append("wait")
            elif line.startswith("s_barrier"):
                events.append("barrier")
            elif line.startswith("buffer_load") and line.endswith("lds"):
                events.append("dma")
        if "dma" not in events:
            bad.append((m.group(1), "no LDS-DMA found: the table hand-over changed - update this check"))
        elif "wait" not in events:
            bad.append((m.group(1), "no explicit s_waitcnt vmcnt(0) in front of the hand-over barrier"))
        else:
            after = events[events.index("wait") + 1:]
            if not after or after[0] != "barrier":
                bad.append((m.group(1), f"the wait is followed by {after[:1]} instead of the hand-over barrier"))
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
    nr, badr = check_reuse_barriers(text)
    for k, plain, lds_only, want in badr:
        print(f"re-use barriers: {k}: {plain} s_barrier + {lds_only} LDS-only, expected at least {want[0]} + exactly {want[1]}")
    print(f"{nr} kernels with LDS re-use checked, {len(badr)} short of their barriers")
    nf, badf = check_fused_letterbox(text)
    for k, why in badf:
        print(f"fused letterbox kernel: {why}: {k}")
    print(f"{nf} fused letterbox kernel(s) checked, {len(badf)} bad")
    sp = {k: v for k, v in spills(text).items() if v[0] or v[2]}
    for k, v in sorted(sp.items()):
        print(f"spills: {k}: {v[0]} VGPRs, {v[1]} SGPRs, scratch {v[2]} B")
    sys.exit(1 if bad or n == 0 or badw or nw == 0 or badr or nr == 0 or badf or nf == 0 else 0)
