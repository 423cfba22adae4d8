#!/usr/bin/env python3
"""Race hunt for the persistent stream kernels: many launches on fresh engines (fresh allocations, cold caches), every result
compared with the whole-line / scalar kernels' on the same frames.  Usage: python tools/stress_stream_kernels.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vid_dup_finder_lib_amd as vdf

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15


def engine(mode):
    os.environ["VDF_RESIZE_MODE"] = str(mode)
    try:
        return vdf.Engine(0)
    finally:
        os.environ.pop("VDF_RESIZE_MODE", None)


def letterboxed(rng, n, h, w):
    f = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
    for c in range(n):
        t, b = int(rng.integers(0, h // 5)), int(rng.integers(0, h // 5))
        l, r = (int(rng.integers(0, w // 6)), int(rng.integers(0, w // 6))) if c % 2 else (0, 0)
        if t: f[c, :, :t] = 16
        if b: f[c, :, h - b:] = 16
        if l: f[c, :, :, :l] = 16
        if r: f[c, :, :, w - r:] = 16
    return f


rng = np.random.default_rng(7)
bad_total = 0
for name, h, w, n, ref_mode, letterbox in (("ksplit 2048", 300, 2048, 40, 4, False), ("ksplit 3840", 200, 3840, 20, 4, False),
                                           ("stream 1024 (re-pitched)", 300, 1024, 40, 4, False), ("stream 422 (shifted)", 240, 422, 40, 1, False),
                                           ("stream band 1920", 300, 1920, 30, 4, False),
                                           ("cropped stream 854", 480, 854, 30, 4, True), ("cropped stream 480", 270, 480, 40, 4, True)):
    frames = letterboxed(rng, n, h, w) if letterbox else rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    call = (lambda e: e.hash_frames_letterbox(frames)[0]) if letterbox else (lambda e: e.hash_frames(frames))
    e = engine(ref_mode); want = call(e); e.close()
    bad = 0
    for rep in range(reps):
        e = engine(0)
        for _ in range(2):
            got = call(e)
            bad += int(not np.array_equal(got, want))
        e.close()
    bad_total += bad
    print(f"{name}: {bad} bad launches of {2 * reps}", flush=True)
sys.exit(1 if bad_total else 0)
