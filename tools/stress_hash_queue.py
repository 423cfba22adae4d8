#!/usr/bin/env python3
"""Stress of the batching queue (vdf_hash_queue_*): many threads, thousands of submissions, random pauses, small and odd batch sizes, short waits
- every returned hash and crop box against the batch call's on the same clip.  (The parity test does 96 submissions from 12 threads.)
Usage (GPU box): python tools/stress_hash_queue.py [--seconds 20]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vid_dup_finder_lib_amd as vdf
from vid_dup_finder_lib_amd.engine import HashQueue

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=20.0)
a = ap.parse_args()
rng = np.random.default_rng(3)
total_bad = 0
for (h, w, letterbox, max_batch, wait_us, n_threads, devices) in [(48, 64, True, 32, 200, 48, None), (64, 64, False, 7, 50, 64, None), (90, 160, True, 64, 2000, 32, None),
                                                                  (270, 480, True, 16, 500, 24, None), (64, 64, True, 256, 100, 96, [0, 0, 0]), (120, 200, False, 5, 0, 40, [0, 0])]:
    eng = vdf.Engine(devices=devices) if devices else vdf.Engine(0)
    n = 512
    frames = rng.integers(40, 220, size=(n, 16, h, w), dtype=np.uint8)
    frames[::2, :, :h // 8, :] = 16
    frames[::3, :, :, -(w // 9):] = 17
    frames[::5, :, :, :w // 10] = (16 + rng.integers(0, 4, size=(len(range(0, n, 5)), 16, h, w // 10))).astype(np.uint8)
    want_h, want_c = eng.hash_frames_letterbox(frames) if letterbox else (eng.hash_frames(frames), np.zeros((n, 4), np.uint32))
    q = HashQueue(eng, w, h, max_batch=max_batch, max_wait_us=wait_us, letterbox=letterbox)
    stop = time.time() + a.seconds / 6
    counts, bad, errs = [0] * n_threads, [0] * n_threads, []

    def worker(t):
        r = np.random.default_rng(100 + t)
        try:
            while time.time() < stop:
                i = int(r.integers(0, n))
                hsh, crop = q.submit(frames[i])
                counts[t] += 1
                if not np.array_equal(hsh, want_h[i]) or tuple(crop) != tuple(int(x) for x in want_c[i]):
                    bad[t] += 1
                if r.random() < 0.2:
                    time.sleep(float(r.random()) * 0.002)
        except Exception as e:
            errs.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=300)
    nb, nc = q.stats()
    q.close(); eng.close()
    total_bad += sum(bad) + len(errs)
    print(f"{w}x{h} letterbox={letterbox} max_batch={max_batch} wait={wait_us}us threads={n_threads} devices={devices}: {sum(counts)} submissions in {nb} batches, {sum(bad)} wrong, errors {errs[:2]}", flush=True)
print(f"== {total_bad} wrong results or errors")
