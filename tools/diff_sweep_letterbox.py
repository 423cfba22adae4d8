#!/usr/bin/env python3
"""Letterbox detect on LARGE batches (hundreds to thousands of clips per launch, first launch of a fresh context) against the oracle: random frame
sizes over all three side-walk kernels, random bars of the fuzz's styles (tests/test_gpu_fuzz.py:_bar) on random subsets of the clips - the work
lists, the persistent side walkers and the atomicMin union under load, which the parity tests' dozen clips per size do not exercise.
Usage (GPU box): python tools/diff_sweep_letterbox.py [--cases 100] [--mb 200] [--seed 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import vid_dup_finder_lib_amd as vdf
from oracle import vdf_oracle as orc
from test_gpu_fuzz import _bar

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=100)
ap.add_argument("--mb", type=int, default=200, help="bytes of PROBED frames (two per clip) per case")
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
n_bad = 0
for case in range(a.cases):
    hk = case % 3
    h = int(rng.integers(17, 256)) if hk == 0 else int(rng.integers(256, 512)) if hk == 1 else int(rng.integers(512, 1100))
    w = int(rng.choice([64, 128, 256, 320, 384, 512, 640, 768, 1024, 1280, 1920])) if case % 2 else int(rng.integers(40, 1500))
    n = int(max(32, min(6000, a.mb * 1_000_000 // (2 * w * h))))
    # only frames 0 and 8 are probed: the clips are two frames long here (frames_per_clip = 9 with stride tricks would need 9 x the bytes)
    probe = np.empty((n, 2, h, w), np.uint8)
    pics = [rng.integers(0, 256, size=(h, w), dtype=np.uint8), rng.integers(0, 30, size=(h, w), dtype=np.uint8),
            (100 + rng.integers(0, 20, size=(h, w))).astype(np.uint8)]
    styles = []
    for _ in range(8):  # eight bar layouts per case, dealt to the clips at random
        f = pics[int(rng.integers(0, 3))].copy()
        t, b = (int(rng.integers(0, h // 3)) if rng.random() < 0.6 else 0 for _ in range(2))
        l, r = (int(rng.integers(0, w // 3)) if rng.random() < 0.6 else 0 for _ in range(2))
        if l: f[:, :l] = _bar(rng, (h, l), 1)
        if r: f[:, w - r:] = _bar(rng, (h, r), 1)
        if t: f[:t, :] = _bar(rng, (t, w), 0)
        if b: f[h - b:, :] = _bar(rng, (b, w), 0)
        styles.append(f)
    styles.append(np.full((h, w), 16, np.uint8))  # a uniform frame
    pick0 = rng.integers(0, len(styles), n)
    pick8 = np.where(rng.random(n) < 0.6, pick0, rng.integers(0, len(styles), n))
    st = np.stack(styles)
    probe[:, 0] = st[pick0]
    probe[:, 1] = st[pick8]
    # the device sees clips of 16 frames with frame stride 0 tricks?  No: lay the clip out as 9 frames (0 .. 8), frames 1 .. 7 never read
    d = torch.zeros((n, 9, h, w), dtype=torch.uint8, device="cuda")
    d[:, 0] = torch.from_numpy(probe[:, 0]).cuda()
    d[:, 8] = torch.from_numpy(probe[:, 1]).cuda()
    crops = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eng = vdf.Engine(0)
    try:
        eng.cropdetect_letterbox_device(d.data_ptr(), n, 9, w, h, crops.data_ptr())
        torch.cuda.synchronize()
    finally:
        eng.close()
    got = crops.cpu().numpy().astype(np.uint32)
    # the oracle per distinct (style of frame 0, style of frame 8) pair
    want = np.zeros((n, 4), np.uint32)
    cache = {}
    for c in range(n):
        key = (int(pick0[c]), int(pick8[c]))
        if key not in cache:
            clip = np.zeros((9, h, w), np.uint8)
            clip[0], clip[8] = st[key[0]], st[key[1]]
            cache[key] = orc.cropdetect_letterbox(clip)
        want[c] = cache[key]
    bad = np.nonzero((got != want).any(axis=1))[0]
    n_bad += len(bad) > 0
    print(f"[{case}] {w}x{h} n={n} distinct pairs {len(cache)}: " + ("ok" if len(bad) == 0 else f"WRONG {len(bad)} clips, first {bad[:4].tolist()} got {got[bad[0]].tolist()} want {want[bad[0]].tolist()}"), flush=True)
print(f"== {a.cases} cases, {n_bad} with mismatches")
