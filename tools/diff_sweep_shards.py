#!/usr/bin/env python3
"""search() on ONE device against the sharded forms on contexts of 2, 3 and 4 slots of the same GPU ({0,0}, {0,0,0}, {0,0,0,0}: the multi-GPU code
path - thread per slot, replication, row tiles dealt cyclically, the round-5 cross-shard replay filter with its bitmap exchange, the merged host
replay) on random databases: sparse, clustered around the tolerance, and duplicate-DENSE (clusters of up to 300 identical-duration near-copies,
where the filter drops most of the adjacency).  Groups must be identical, member for member.  Host-array calls and device-shard calls.
Usage (GPU box): python tools/diff_sweep_shards.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vid_dup_finder_lib_amd as vdf
from vid_dup_finder_lib_amd import distributed as vd

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-n", type=int, default=200_000)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
engines = {1: vdf.Engine(devices=[0]), 2: vdf.Engine(devices=[0, 0]), 3: vdf.Engine(devices=[0, 0, 0]), 4: vdf.Engine(devices=[0, 0, 0, 0])}


def flip(words, idx_src, idx_dst, nflips):
    """words[idx_dst] = words[idx_src] with nflips random bits (of the 1024) flipped."""
    for s, d, k in zip(idx_src, idx_dst, nflips):
        bits = np.unpackbits(words[s].view(np.uint8), bitorder="little")
        bits[rng.choice(1024, size=int(k), replace=False)] ^= 1
        words[d] = np.packbits(bits, bitorder="little").view(np.uint64)


n_bad = 0
for case in range(a.cases):
    n = int(np.exp(rng.uniform(np.log(300), np.log(a.max_n))))
    tol = int(rng.choice([250, 300, 350, 350, 400]))
    words = rng.integers(0, 2 ** 64, size=(n, 16), dtype=np.uint64)
    words[:, 15] &= np.uint64((1 << 40) - 1)
    style = case % 3
    dur = np.zeros(n, np.uint32)
    if style == 0:    # sparse: a per cent of near-copies
        k = max(1, n // 100)
        flip(words, rng.integers(0, n, k), rng.integers(0, n, k), rng.integers(max(0, tol - 30), tol + 30, k))
        dur = np.exp(rng.uniform(0, np.log(7200), n)).astype(np.uint32)
    elif style == 1:  # clustered: a tenth in clusters of 2 ... 20 around random centres, distances on both sides of the tolerance
        members = rng.choice(n, size=max(2, n // 10), replace=False)
        centres = rng.choice(members, size=max(1, len(members) // 8), replace=False)
        flip(words, rng.choice(centres, len(members)), members, rng.integers(0, tol // 2 + 40, len(members)))
        dur = rng.integers(0, 12, n).astype(np.uint32)
    else:             # dense: clusters of up to 300 near-identical entries with one duration
        pos = 0
        while pos < n // 5:
            size = int(rng.integers(2, 300))
            idx = rng.choice(n, size=size, replace=False)
            flip(words, np.full(size - 1, idx[0]), idx[1:], rng.integers(0, 60, size - 1))
            dur[idx] = int(rng.integers(1, 50))
            pos += size
    order = np.argsort(dur, kind="stable")
    words, dur = np.ascontiguousarray(words[order]), np.ascontiguousarray(dur[order])
    ref = engines[1].search_self_sorted(words, dur, tol)
    # search_with_references: references in the caller's order (unsorted durations), drawn from the database and from nowhere
    nr = int(rng.integers(1, max(2, n // 10)))
    ridx = rng.integers(0, n, nr)
    rw, rd = words[ridx].copy(), dur[ridx].copy()
    fresh = rng.random(nr) < 0.3
    rw[fresh] = rng.integers(0, 2 ** 64, size=(int(fresh.sum()), 16), dtype=np.uint64)
    ref_r = engines[1].search_refs_sorted(words, dur, rw, rd, tol)
    line = []
    for G in (2, 3, 4):
        got = engines[G].search_self_sorted(words, dur, tol)
        ok = got == ref
        # device-shard form
        sw, sd, sizes = [], [], []
        for k in range(G):
            lo, hi = vd.split_range(n, k, G)
            sw.append(torch.from_numpy(words[lo:hi].view(np.int64)).to(dev))
            sd.append(torch.from_numpy(dur[lo:hi].view(np.int32)).to(dev))
            sizes.append(hi - lo)
        torch.cuda.synchronize()
        got2 = engines[G].search_self_shards([t.data_ptr() for t in sw], [t.data_ptr() for t in sd], sizes, tol)
        ok2 = got2 == ref
        tm = engines[G].last_timing()
        ok3 = engines[G].search_refs_sorted(words, dur, rw, rd, tol) == ref_r
        n_bad += (not ok) + (not ok2) + (not ok3)
        line.append(f"G={G}: host {'same' if ok else 'DIFFERENT'}, shards {'same' if ok2 else 'DIFFERENT'} (filtered {tm['hits_filtered']}), refs {'same' if ok3 else 'DIFFERENT'}")
    print(f"[{case}] n={n} tol={tol} style {('sparse', 'clustered', 'dense')[style]}: groups {len(ref)} members {sum(len(g) for g in ref)}, {nr} refs -> {len(ref_r)} groups: " + "; ".join(line), flush=True)
print(f"== {a.cases} cases, {n_bad} differing")
