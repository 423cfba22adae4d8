#!/usr/bin/env python3
"""vdf_hash_frames_u8_letterbox_device (detect -> clip-by-clip deal to the row-range / column-range / cropped kernels -> one DCT launch) on large
mixed batches against the long way round: vdf_cropdetect_letterbox_device, the boxes brought to the host, vdf_hash_frames_u8_cropped_device on the
GENERAL kernels (VDF_RESIZE_MODE=4, VDF_NO_ROWCROP).  First launch of fresh contexts; random sizes; six bar layouts per case dealt to the clips
at random (none, top / bottom, sides, all four, short boxes, noisy bars).
Usage (GPU box): python tools/diff_sweep_letterbox_hash.py [--cases 100] [--mb 600] [--seed 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vid_dup_finder_lib_amd as vdf

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=100)
ap.add_argument("--mb", type=int, default=600)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(a.seed)


def engine_with(env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        return vdf.Engine(0)
    finally:
        for k in env:
            os.environ.pop(k, None)


n_bad = 0
for case in range(a.cases):
    w = int(rng.choice([64, 128, 176, 240, 256, 320, 426, 480, 512, 640, 720, 854, 960, 1024, 1280, 1366, 1440, 1600, 1920, 2048, 2560])) if case % 2 else int(rng.integers(64, 2100))
    h = int(rng.choice([int(rng.integers(40, 130)), int(rng.integers(130, 400)), int(rng.integers(400, 1100))]))
    if rng.random() < 0.75 and (w * h) % 16:
        h = (h + 15) // 16 * 16
    n = int(max(40, min(20000, a.mb * 1_000_000 // (16 * w * h))))
    frames = torch.randint(30, 256, (n, 16, h, w), dtype=torch.uint8, device=dev, generator=g)  # picture: never within 16 of the bars' 0 ... 12
    layouts = [(0, 0, 0, 0)]
    for _ in range(5):
        t, b = (int(rng.integers(0, h // 3)) if rng.random() < 0.7 else 0 for _ in range(2))
        l, r = (int(rng.integers(0, w // 3)) if rng.random() < 0.5 else 0 for _ in range(2))
        if rng.random() < 0.25:  # a short box
            keep = int(rng.integers(2, 70))
            if h - t - b > keep: b = h - t - keep
        layouts.append((l, r, t, b))
    pick = torch.from_numpy(rng.integers(0, len(layouts), n)).to(dev)
    for s, (l, r, t, b) in enumerate(layouts):
        idx = torch.nonzero(pick == s).flatten()
        if len(idx) == 0 or (l, r, t, b) == (0, 0, 0, 0):
            continue
        noisy = s % 2 == 1
        def bar(shape):
            return torch.randint(0, 4, shape, dtype=torch.uint8, device=dev, generator=g) + 8 if noisy else torch.full(shape, 5, dtype=torch.uint8, device=dev)
        if l: frames[idx, :, :, :l] = bar((len(idx), 16, h, l))
        if r: frames[idx, :, :, w - r:] = bar((len(idx), 16, h, r))
        if t: frames[idx, :, :t, :] = bar((len(idx), 16, t, w))
        if b: frames[idx, :, h - b:, :] = bar((len(idx), 16, b, w))
    torch.cuda.synchronize()
    # the long way round
    eng = engine_with({"VDF_RESIZE_MODE": "4", "VDF_NO_ROWCROP": "1"})
    crops_d = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    ref = torch.zeros((n, 16), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # the library's stream does not wait for torch's fills
    eng.cropdetect_letterbox_device(frames.data_ptr(), n, 16, w, h, crops_d.data_ptr())
    torch.cuda.synchronize()
    crops = crops_d.cpu().numpy().astype(np.uint32)
    eng.hash_frames_cropped_device(frames.data_ptr(), n, 16, w, h, crops, ref.data_ptr())
    torch.cuda.synchronize()
    eng.close()
    want_crops = np.array(layouts, np.uint32)[pick.cpu().numpy()]
    crops_ok = np.array_equal(crops, want_crops)
    msgs = []
    for name, env in (("default", {}), ("no_wavestream", {"VDF_NO_WAVESTREAM": "1"})):
        eng = engine_with(env)
        out = torch.zeros((n, 16), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        got_crops = eng.hash_frames_letterbox_device(frames.data_ptr(), n, 16, w, h, out.data_ptr())
        torch.cuda.synchronize()
        eng.close()
        bad = torch.nonzero((out != ref).any(dim=1)).flatten()
        cbad = 0 if got_crops is None else int((np.asarray(got_crops, np.uint32).reshape(n, 4) != crops).any(axis=1).sum())
        if len(bad) or cbad:
            n_bad += 1
            gc = np.asarray(got_crops, np.uint32).reshape(n, 4)
            first = int(np.nonzero((gc != crops).any(axis=1))[0][0]) if cbad else int(bad[0])
            msgs.append(f"{name}: WRONG hashes {len(bad)} {bad[:5].tolist()} crops {cbad}; clip {first}: box from the letterbox call {gc[first].tolist()}, from the detect call {crops[first].tolist()}, planted {want_crops[first].tolist()}")
        else:
            msgs.append(f"{name}: ok")
    print(f"[{case}] {w}x{h} n={n} layouts {layouts[1:]} detect {'as planted' if crops_ok else 'differs from the planted bars (dark picture / wide bars: not an error)'}: " + "; ".join(msgs), flush=True)
    del frames
print(f"== {a.cases} cases, {n_bad} mismatching (variant, case) pairs")
