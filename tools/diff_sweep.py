#!/usr/bin/env python3
"""Differential sweep for races: LARGE batches, FIRST launch of a fresh context, every kernel family against the whole-line kernels.
(Round 5 found the one-chunk hand-over race this way: 1 - 2 clips in 30 000, first launch only - far below what the parity tests' few
clips per size can show.)  Uncropped: default dispatch and the forced stream / K-split / fused forms against VDF_RESIZE_MODE=4; cropped:
random boxes (per size: one box for all clips, or per-clip boxes) through the default dispatch against the general kernels.
Usage (GPU box): python tools/diff_sweep.py [--cases 200] [--mb 800] [--seed 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vid_dup_finder_lib_amd as vdf

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--mb", type=int, default=800)
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


def run(env, fn):
    eng = engine_with(env)
    try:
        out = fn(eng)
        torch.cuda.synchronize()
        return out
    finally:
        eng.close()


n_bad = 0
for case in range(a.cases):
    kind = case % 3  # 0: uncropped, 1: one box for all clips, 2: per-clip boxes
    cls = int(rng.integers(0, 6))
    w = int([rng.integers(64, 200), rng.integers(200, 520), rng.integers(520, 1000), rng.integers(1000, 1930), rng.integers(1930, 4100),
             rng.choice([64, 128, 176, 256, 320, 426, 480, 512, 640, 720, 854, 960, 1024, 1280, 1366, 1440, 1600, 1920, 2048, 2560, 3840])][cls])
    h = int(rng.choice([int(rng.integers(17, 130)), int(rng.integers(130, 400)), int(rng.integers(400, 1100))], p=[0.35, 0.4, 0.25]))
    if rng.random() < 0.7:
        h = (h + 15) // 16 * 16 if (w * h) % 16 else h  # mostly frames that end on 16 bytes (the stream kernels' condition)
    n = int(max(48, min(40000, a.mb * 1_000_000 // (16 * w * h))))
    frames = torch.randint(0, 256, (n, 16, h, w), dtype=torch.uint8, device=dev, generator=g)
    torch.cuda.synchronize()
    out_shape = (n, 16)
    if kind == 0:
        def call(eng):
            out = torch.zeros(out_shape, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()  # the library's stream does not wait for torch's fill
            eng.hash_frames_device(frames.data_ptr(), n, 16, w, h, out.data_ptr())
            return out
        ref = run({"VDF_RESIZE_MODE": "4"}, call)
        variants = [("default", {}), ("mode5", {"VDF_RESIZE_MODE": "5"}), ("mode6", {"VDF_RESIZE_MODE": "6"})]
        if h <= 128:
            variants.append(("mode3", {"VDF_RESIZE_MODE": "3"}))
        desc = f"uncropped {w}x{h} n={n}"
    else:
        crops = np.zeros((n, 4), np.uint32)
        def box():
            t, b = (int(rng.integers(0, max(1, h // 2))) for _ in range(2))
            if t + b >= h: b = 0
            if kind == 1 and rng.random() < 0.5 or kind == 2 and rng.random() < 0.4:
                l = r = 0
            else:
                l, r = (int(rng.integers(0, max(1, w // 3))) for _ in range(2))
            if rng.random() < 0.3:  # short boxes: few blocks, one chunk
                keep = int(rng.integers(1, 70))
                if h - t - b > keep: b = h - t - keep
            return (l, r, t, b)
        if kind == 1:
            crops[:] = box()
        else:
            pool = [box() for _ in range(6)] + [(0, 0, 0, 0)]
            crops[:] = np.array(pool, np.uint32)[rng.integers(0, len(pool), n)]
        def call(eng):
            out = torch.zeros(out_shape, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()  # the library's stream does not wait for torch's fill
            eng.hash_frames_cropped_device(frames.data_ptr(), n, 16, w, h, crops, out.data_ptr())
            return out
        ref = run({"VDF_RESIZE_MODE": "4", "VDF_NO_ROWCROP": "1"}, call)
        variants = [("default", {}), ("rowcrop_all", {"VDF_ROWCROP_ALL": "1"}), ("mode5", {"VDF_RESIZE_MODE": "5"})]
        desc = f"cropped({'one box' if kind == 1 else 'per clip'}) {w}x{h} n={n} box0={tuple(int(v) for v in crops[0])}"
    line = []
    for name, env in variants:
        try:
            out = run(env, call)
        except vdf.VdfError as e:
            line.append(f"{name}: refused({e.code})")
            continue
        bad = torch.nonzero((out != ref).any(dim=1)).flatten()
        if len(bad):
            n_bad += 1
            line.append(f"{name}: WRONG {len(bad)} clips {bad[:6].tolist()}")
        else:
            line.append(f"{name}: ok")
    print(f"[{case}] {desc}: " + "; ".join(line), flush=True)
    del frames
print(f"== {a.cases} cases, {n_bad} mismatching (variant, case) pairs")
