#!/usr/bin/env python3
"""Whole-call wall time of the search entry points over sizes, tolerances and duration shapes nobody benchmarks: vdf_search_self on a
device-resident database (one-device shards call: kernel + download + replay), vdf_search_refs_device with few and many references."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf
from bench import make_hashes

eng = vdf.Engine(devices=[0])
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)

def self_call(words, dur, tol, reps=3):
    n = len(words)
    dw = torch.from_numpy(words.view(np.int64)).to(dev); dd = torch.from_numpy(dur.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    eng.search_self_shards([dw.data_ptr()], [dd.data_ptr()], [n], tol, as_arrays=True)
    best, g = 1e9, None
    for _ in range(reps):
        t = time.perf_counter(); g = eng.search_self_shards([dw.data_ptr()], [dd.data_ptr()], [n], tol, as_arrays=True); best = min(best, time.perf_counter() - t)
    tm = eng.last_timing()
    return best * 1e3, tm, g

print("== self search, all durations equal (full triangle), tolerance 350: whole call")
for n in (1000, 10_000, 100_000, 300_000, 1_000_000):
    w = make_hashes(n, 1); ms, tm, g = self_call(w, np.zeros(n, np.uint32), 350)
    print(f"n={n:8d}: {ms:8.3f} ms  (stream {tm['stream_ms']:.3f} resolve {tm['resolve_ms']:.3f} download {tm['download_ms']:.3f} replay {tm['replay_ms']:.3f}) {n*(n-1)/2/ms/1e9:.1f} T pairs/s")
print("== 300 k hashes, tolerance sweep (random hashes: hits only from the planted copies)")
w = make_hashes(300_000, 2)
for tol in (0, 100, 350, 400, 450, 480, 500):
    ms, tm, g = self_call(w, np.zeros(len(w), np.uint32), tol, reps=2)
    print(f"tol={tol:4d}: {ms:9.3f} ms  suspects {tm['suspects']}  hits_filtered {tm['hits_filtered']}  (stream {tm['stream_ms']:.2f} resolve {tm['resolve_ms']:.2f} download {tm['download_ms']:.2f} replay {tm['replay_ms']:.2f})")
print("== 1 M hashes, duration shapes, tolerance 350")
n = 1_000_000; w = make_hashes(n, 3)
shapes = {"log-uniform 5..7200 s": np.exp(rng.uniform(np.log(5), np.log(7200), n)).astype(np.uint32),
          "all distinct (0..n-1)": np.arange(n, dtype=np.uint32),
          "10 % zero (unknown length), rest log-uniform": np.where(rng.random(n) < 0.1, 0, np.exp(rng.uniform(np.log(5), np.log(7200), n))).astype(np.uint32),
          "two values (60 s, 61 s)": (60 + (rng.random(n) < 0.5)).astype(np.uint32)}
for name, d in shapes.items():
    order = np.argsort(d, kind="stable")
    ms, tm, g = self_call(w[order], d[order], 350, reps=2)
    print(f"{name:48s}: {ms:9.3f} ms (stream {tm['stream_ms']:.2f})")
print("== references against 1 M candidates (log-uniform durations), whole vdf_search_refs_device call")
eng1 = vdf.Engine(0)  # the device-pointer entry points take a single-device context
d = np.sort(shapes["log-uniform 5..7200 s"]); cw = torch.from_numpy(w.view(np.int64)).to(dev); cd = torch.from_numpy(d.view(np.int32)).to(dev)
for nr in (1, 10, 100, 1000, 10_000, 100_000, 1_000_000):
    idx = rng.integers(0, n, nr); rw = torch.from_numpy(w[idx].view(np.int64)).to(dev); rd = torch.from_numpy(d[idx].view(np.int32)).to(dev)
    cap = max(1 << 16, 4 * nr)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); r = eng1.search_refs_device(cw.data_ptr(), cd.data_ptr(), n, rw.data_ptr(), rd.data_ptr(), nr, 350, capacity=cap); best = min(best, time.perf_counter() - t)
    print(f"n_ref={nr:8d}: {best*1e3:9.3f} ms  hits {r[1] if isinstance(r, tuple) else len(r)}")
print("== the same with the candidate database pinned (vdf_ctx_pin_database): whole call, and its phases")
eng1.pin_database(cw.data_ptr(), n)
for nr in (1, 10, 100, 1000, 10_000, 100_000):
    idx = rng.integers(0, n, nr); rw = torch.from_numpy(w[idx].view(np.int64)).to(dev); rd = torch.from_numpy(d[idx].view(np.int32)).to(dev)
    cap = max(1 << 16, 4 * nr)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); r = eng1.search_refs_device(cw.data_ptr(), cd.data_ptr(), n, rw.data_ptr(), rd.data_ptr(), nr, 350, capacity=cap); dt = time.perf_counter() - t
        if dt < best: best, tm = dt, eng1.last_timing()
    print(f"n_ref={nr:8d}: {best*1e3:9.3f} ms  hits {r[1] if isinstance(r, tuple) else len(r)}  " + " ".join(f"{k} {v:.3f}" if isinstance(v, float) else f"{k} {v}" for k, v in tm.items()))
eng1.pin_database(0, 0)
print("== few candidates against 1 M references (the incremental case the other way round)")
for nc in (1, 10, 100, 1000):
    idx = np.sort(rng.integers(0, n, nc)); cw2 = torch.from_numpy(w[idx].view(np.int64)).to(dev); cd2 = torch.from_numpy(d[idx].view(np.int32)).to(dev)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t = time.perf_counter(); r = eng1.search_refs_device(cw2.data_ptr(), cd2.data_ptr(), nc, cw.data_ptr(), cd.data_ptr(), n, 350, capacity=1 << 22); dt = time.perf_counter() - t
        if dt < best: best, tm = dt, eng1.last_timing()
    print(f"n_cand={nc:7d}: {best*1e3:9.3f} ms  hits {r[1] if isinstance(r, tuple) else len(r)}  " + " ".join(f"{k} {v:.3f}" if isinstance(v, float) else f"{k} {v}" for k, v in tm.items()))
