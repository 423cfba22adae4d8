#!/usr/bin/env python3
"""Small letterboxed frames: detect + crop + hash through the three routes of vdf_hash_frames_u8_letterbox_device[_async]
(default = boxes stay on the device, fused kernel for frames of at most 64 x 64; VDF_NO_LB_FUSED = detect kernels + cropped kernel reading
the boxes on the device; VDF_LB_HOST_PLAN = round 5: boxes to the host, host plan) beside the plain hash of the same clips.
Every route's hashes and boxes are compared with the oracle on the first --check clips of every pattern.
    python tools/bench_letterbox_small.py [--clips 20000] [--w 64] [--h 64]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=20000)
ap.add_argument("--w", type=int, default=64)
ap.add_argument("--h", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--check", type=int, default=300)
ap.add_argument("--child", action="store_true")
a = ap.parse_args()

if not a.child:
    for label, env in (("default (device boxes, fused <= 64x64)", {}), ("VDF_NO_LB_FUSED", {"VDF_NO_LB_FUSED": "1"}),
                       ("VDF_LB_HOST_PLAN (round 5)", {"VDF_LB_HOST_PLAN": "1"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--clips", str(a.clips), "--w", str(a.w), "--h", str(a.h),
                            "--steps", str(a.steps), "--check", str(a.check)], env=e, capture_output=True, text=True)
        print(f"== {label}: {a.clips} clips of 16 x {a.h} x {a.w}")
        print(r.stdout.strip())
        if r.returncode:
            print("FAILED", r.stderr[-2000:])
    sys.exit(0)

import numpy as np, torch
import vid_dup_finder_lib_amd as vdf
from oracle import vdf_oracle as orc

dev = torch.device("cuda", 0)
eng = vdf.Engine(0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
g = torch.Generator(device=dev); g.manual_seed(1)
n, w, h = a.clips, a.w, a.h
base = torch.randint(0, 256, (n, 16, h, w), dtype=torch.uint8, device=dev, generator=g)
bar_t, bar_s = max(1, int(h * 0.12)), max(1, int(w * 0.125))


def pattern(name):
    fr = base.clone()
    if name in ("top_bottom", "both", "noisy_bars"):
        fr[:, :, :bar_t, :] = 16; fr[:, :, h - bar_t:, :] = 16
    if name in ("side", "both"):
        fr[:, :, :, :bar_s] = 16; fr[:, :, :, w - bar_s:] = 16
    if name == "noisy_bars":
        nz = torch.randint(0, 4, fr.shape, dtype=torch.uint8, device=dev, generator=g)
        fr = torch.where(fr == 16, 16 + nz, fr)
    if name == "smooth":  # video-like: a gradient + mild noise, bars on a third of the clips - the strip tests' undecided cases
        yy = torch.arange(h, device=dev).view(1, 1, h, 1).float(); xx = torch.arange(w, device=dev).view(1, 1, 1, w).float()
        ph = torch.rand((n, 16, 1, 1), device=dev, generator=g) * 40
        img = 90 + 50 * torch.sin(xx / w * 3.0 + ph / 9) + 30 * torch.cos(yy / h * 2.0) + torch.randn((n, 16, h, w), device=dev, generator=g) * 3
        fr = img.clamp(0, 255).to(torch.uint8)
        fr[::3, :, :bar_t, :] = 20; fr[::3, :, h - bar_t:, :] = 20
        fr[1::3, :, :, :bar_s] = 18; fr[1::3, :, :, w - bar_s:] = 18
    return fr


out = torch.zeros((n, 16), dtype=torch.int64, device=dev)
dcr = torch.zeros((n, 4), dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    fn(); torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for _ in range(3):
        e0.record()
        for _ in range(a.steps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        best = min(best, ms); tot += ms
    return best


res = {}
for name in ("no_bars", "top_bottom", "side", "both", "noisy_bars", "smooth"):
    fr = base if name == "no_bars" else pattern(name)
    k = min(a.check, n)
    want_h, want_c = [], []
    host = fr[:k].cpu().numpy()
    for c in range(k):
        _, hh, _, cc = orc.hash_clip_letterbox(host[c])
        want_h.append(hh); want_c.append(cc)
    want_h, want_c = np.array(want_h), np.array(want_c)
    out.zero_(); dcr.fill_(-1)
    crops = eng.hash_frames_letterbox_device(fr.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    ok_sync = np.array_equal(out[:k].cpu().numpy().view(np.uint64), want_h) and np.array_equal(crops[:k], want_c)
    out.zero_()
    eng.hash_frames_letterbox_device(fr.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream, d_crops=dcr.data_ptr())
    torch.cuda.synchronize()
    ok_async = np.array_equal(out[:k].cpu().numpy().view(np.uint64), want_h) and np.array_equal(dcr[:k].cpu().numpy().astype(np.uint32), want_c)
    # the whole batch: both calls agree with each other on every clip
    full = np.array_equal(dcr.cpu().numpy().astype(np.uint32), crops)
    t_plain = timed(lambda: eng.hash_frames_device(fr.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream))
    t_sync = timed(lambda: eng.hash_frames_letterbox_device(fr.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream))
    t_async = timed(lambda: eng.hash_frames_letterbox_device(fr.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream, d_crops=dcr.data_ptr()))
    gb = n * 16 * w * h / 1e9
    print(f"{name:11s} oracle: sync {'ok' if ok_sync else 'MISMATCH'} async {'ok' if ok_async else 'MISMATCH'} same {'ok' if full else 'MISMATCH'} | "
          f"plain {t_plain:.3f} ms | letterbox host boxes {t_sync:.3f} ms | device boxes {t_async:.3f} ms = {gb / t_async:.2f} TB/s of frames | crop[0] {list(map(int, crops[0]))}")
    if fr is not base:
        del fr
