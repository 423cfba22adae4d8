#!/usr/bin/env python3
"""Focused timing of the VideoHash construction kernels (for rocprofv3 runs): frames resident in HBM."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vid_dup_finder_lib_amd as vdf

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=100000)
ap.add_argument("--w", type=int, default=64)
ap.add_argument("--h", type=int, default=64)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda", 0)
eng = vdf.Engine(0)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
g = torch.Generator(device=dev)
g.manual_seed(20250617)
frames = torch.randint(0, 256, (args.clips, 16, args.h, args.w), dtype=torch.uint8, device=dev, generator=g)
out = torch.zeros((args.clips, 16), dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.hash_frames_device(frames.data_ptr(), args.clips, 16, args.w, args.h, out.data_ptr(), stream=st.cuda_stream)
torch.cuda.synchronize()
e0.record()
for _ in range(args.steps):
    eng.hash_frames_device(frames.data_ptr(), args.clips, 16, args.w, args.h, out.data_ptr(), stream=st.cuda_stream)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.steps
nbytes = args.clips * 16 * (args.w * args.h + 8)
print(f"{args.clips} clips of 16x{args.h}x{args.w}: {ms:.3f} ms/step, {args.clips * 16 / ms * 1e3:.4g} frames/s, "
      f"{nbytes / ms / 1e6:.1f} GB/s algorithmic ({nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s)")
