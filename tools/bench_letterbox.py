#!/usr/bin/env python3
"""Timing of detect + crop + hash (vdf_hash_frames_u8_letterbox_device) on HBM-resident clips."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=20000)
ap.add_argument("--w", type=int, default=64)
ap.add_argument("--h", type=int, default=64)
ap.add_argument("--bars", type=float, default=0.12, help="letterbox bar height as a fraction of H (0 = none)")
ap.add_argument("--side", type=float, default=0.0, help="pillarbox bar width as a fraction of W (0 = none)")
ap.add_argument("--black", type=float, default=0.0, help="fraction of clips whose frame 0 is uniformly black (a fade-in: every strip of every edge is letterbox)")
ap.add_argument("--mix", action="store_true", help="a mixed batch: 70 %% of the clips without bars, 20 %% with --bars top / bottom, 10 %% with --side bars")
ap.add_argument("--noise", type=int, default=0, help="bars are 16 + U{0..noise} per pixel instead of one value (what a lossy codec leaves of a black bar)")
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
eng = vdf.Engine(0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
frames = torch.randint(0, 256, (a.clips, 16, a.h, a.w), dtype=torch.uint8, device=dev, generator=g)
bar = int(a.h * a.bars)
side = int(a.w * a.side)
tb = slice(None) if not a.mix else slice(7, None, 10)   # clips 7, 17, ...  (and 8, 18, ... below): 20 %
tb2 = slice(0, 0) if not a.mix else slice(8, None, 10)
sd = slice(None) if not a.mix else slice(9, None, 10)   # 10 %
if bar:
    for sl in (tb, tb2):
        frames[sl, :, :bar, :] = 16
        frames[sl, :, a.h - bar:, :] = 16
if side:
    frames[sd, :, :, :side] = 16
    frames[sd, :, :, a.w - side:] = 16
if a.noise > 0:
    nz = torch.randint(0, a.noise + 1, (a.clips, 16, a.h, a.w), dtype=torch.uint8, device=dev, generator=g)
    frames = torch.where(frames == 16, 16 + nz, frames)
    del nz
if a.black > 0:
    frames[:: max(1, int(round(1 / a.black))), 0] = 16
out = torch.zeros((a.clips, 16), dtype=torch.int64, device=dev)
crops_d = torch.zeros((a.clips, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("detect only", lambda: eng.cropdetect_letterbox_device(frames.data_ptr(), a.clips, 16, a.w, a.h, crops_d.data_ptr(), stream=st.cuda_stream)),
                 ("detect+crop+hash", lambda: eng.hash_frames_letterbox_device(frames.data_ptr(), a.clips, 16, a.w, a.h, out.data_ptr(), stream=st.cuda_stream))):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(a.steps): r = fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    print(f"{name}: {a.clips} clips 16x{a.h}x{a.w} bars={bar} side={side} black={a.black}: {ms:.3f} ms, {a.clips*16/ms*1e3:.4g} frames/s, {a.clips*16*a.w*a.h/ms/1e6:.1f} GB/s of frames")
print("crop[0] =", r[0] if r is not None else None)
