#!/usr/bin/env python3
"""One sweep over the frame-size classes of the VideoHash path: every VDF_RESIZE_MODE on every size, TB/s of frame bytes
(tools/bench_hash.py's figure).  Feeds the size-class table of DESIGN.md 4.1 and the pruning of resize_dispatch.cpp.
Usage (on the GPU box): python tools/sweep_resize_modes.py [--modes 0,4,6] > gpurun_out/.../resize_sweep.txt"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vid_dup_finder_lib_amd as vdf

SIZES = [(16, 16), (32, 32), (48, 32), (64, 64), (60, 44), (96, 96), (128, 128), (100, 300), (160, 200), (176, 144), (240, 426),
         (320, 240), (426, 240), (480, 270), (500, 300), (640, 360), (640, 480), (720, 480), (720, 576), (768, 432), (854, 480),
         (960, 540), (1024, 576), (1152, 648), (1280, 720), (1366, 768), (1440, 1080), (1536, 864), (1600, 900), (1680, 1050), (1792, 1008),
         (1920, 1080), (2000, 1125), (2048, 1152), (2560, 1440), (3840, 2160), (4200, 2200)]
ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="0,4,6")
ap.add_argument("--mb", type=int, default=1500, help="frame bytes per launch")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--sizes", default="", help="WxH,WxH,... instead of the built-in list")
args = ap.parse_args()
if args.sizes:
    SIZES = [tuple(int(v) for v in s.split("x")) for s in args.sizes.split(",")]
modes = [int(m) for m in args.modes.split(",")]
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
g = torch.Generator(device=dev)
g.manual_seed(1)
print("# w x h: clips | " + " | ".join(f"mode {m}" for m in modes) + "   (TB/s of frame bytes; - = mode refused the size)")
for w, h in SIZES:
    n = max(96, min(400_000, args.mb * 1_000_000 // (16 * w * h)))  # at least 1536 frames: six workgroups per CU
    frames = torch.empty((n, 16, h, w), dtype=torch.uint8, device=dev)
    chunk = max(1, (1 << 30) // (16 * h * w))
    for c0 in range(0, n, chunk):
        frames[c0:c0 + chunk] = torch.randint(0, 256, (min(chunk, n - c0), 16, h, w), dtype=torch.uint8, device=dev, generator=g)
    out = torch.zeros((n, 16), dtype=torch.int64, device=dev)
    ref = None
    cells = []
    for m in modes:
        if m == 1 and w * h > 128 * 128:
            cells.append("  .  ")
            continue
        os.environ["VDF_RESIZE_MODE"] = str(m)
        eng = vdf.Engine(0)
        os.environ.pop("VDF_RESIZE_MODE")
        try:
            eng.hash_frames_device(frames.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            elif not torch.equal(ref, out):
                cells.append("WRONG")
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                eng.hash_frames_device(frames.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            cells.append(f"{n * 16 * (w * h + 8) / ms / 1e9:5.2f}")
        except vdf.VdfError:
            cells.append("  -  ")
        finally:
            eng.close()
    print(f"{w:5d} x {h:4d}: {n:6d} | " + " | ".join(cells), flush=True)
    del frames, out
