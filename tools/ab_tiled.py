#!/usr/bin/env python3
"""Frames of up to 128 x 128: the tiled persistent kernel against the one-workgroup-per-clip kernel (VDF_HASH_NO_PERSISTENT=1) - TB/s of frame
bytes and equality of every hash.  Usage (GPU box): python tools/ab_tiled.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vid_dup_finder_lib_amd as vdf
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
print("# w x h: clips | per-clip kernel | tiled persistent   (TB/s of frame bytes)")
SIZES = [(80, 48), (96, 64), (80, 80), (96, 96), (112, 112), (128, 72), (128, 96), (128, 128), (64, 128), (64, 96), (48, 100), (128, 64), (112, 63), (32, 128), (16, 100), (128, 17),
         (144, 81), (160, 90), (160, 120), (176, 99), (192, 108), (192, 64), (192, 80), (160, 64), (144, 128), (192, 128),
         (208, 117), (224, 126), (256, 64), (256, 96), (256, 128), (240, 100), (208, 80)]
if len(sys.argv) > 1:
    SIZES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1].split(",")]
for w, h in SIZES:
    n = max(96, min(400_000, 1500 * 1_000_000 // (16 * w * h)))
    frames = torch.randint(0, 256, (n, 16, h, w), dtype=torch.uint8, device=dev, generator=g)
    torch.cuda.synchronize()
    cells, ref = [], None
    for env in ({"VDF_HASH_NO_PERSISTENT": "1"}, {}):
        os.environ.update(env)
        eng = vdf.Engine(0)
        for k in env: os.environ.pop(k)
        out = torch.zeros((n, 16), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        eng.hash_frames_device(frames.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        elif not torch.equal(ref, out):
            cells.append("WRONG"); eng.close(); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.hash_frames_device(frames.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
        e1.record(); torch.cuda.synchronize()
        cells.append(f"{n * 16 * w * h / (e0.elapsed_time(e1) / 5) / 1e9:5.2f}")
        eng.close()
    print(f"{w:5d} x {h:4d}: {n:6d} | " + " | ".join(cells), flush=True)
