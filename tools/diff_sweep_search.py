#!/usr/bin/env python3
"""Differential sweep of the two Hamming backends (fp4 Gram matrix on the matrix cores against XOR + popcount on the VALU) on random databases:
fresh contexts, first launch, sizes 1 k ... 400 k, clustered near-duplicates around the tolerance, every duration shape, random tolerances -
search() groups and search_with_references() hit lists must be identical.  (The parity tests compare each backend with the oracle at sizes the
oracle finishes in seconds; this covers the sizes in between those and the 1 M / 10 M property tests.)
Usage (GPU box): python tools/diff_sweep_search.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vid_dup_finder_lib_amd as vdf

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-n", type=int, default=400_000)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(a.seed)


def engine_with(env, **kw):
    for k, v in env.items():
        os.environ[k] = v
    try:
        return vdf.Engine(**kw)
    finally:
        for k in env:
            os.environ.pop(k, None)


def database(n, tol):
    w = torch.randint(-(2 ** 63), 2 ** 63 - 1, (n, 16), dtype=torch.int64, device=dev, generator=g)
    w[:, 15] &= (1 << 40) - 1
    # near-duplicates: a tenth of the entries are copies of other entries with a random number of flipped bits around the tolerance
    k = max(1, n // 10)
    src = torch.randint(0, n, (k,), device=dev, generator=g)
    dst = torch.randint(0, n, (k,), device=dev, generator=g)
    flips = torch.randint(max(0, tol - 40), tol + 40, (k,), device=dev, generator=g).clamp(min=0)
    # each of the 1024 bit positions flips with probability flips / 1024 (so distances scatter on both sides of the tolerance)
    p = (flips.float() / 1024.0)[:, None, None]
    mask_bits = (torch.rand((k, 16, 64), device=dev, generator=g) < p)
    weights = (1 << torch.arange(63, device=dev, dtype=torch.int64))
    mask = (mask_bits[:, :, :63].long() * weights).sum(dim=2) | (mask_bits[:, :, 63].long() << 63)
    w[dst] = w[src] ^ mask
    kind = int(rng.integers(0, 5))
    if kind == 0:
        d = torch.zeros(n, dtype=torch.int64, device=dev)
    elif kind == 1:
        d = torch.randint(0, 12, (n,), device=dev, generator=g)
    elif kind == 2:
        d = torch.exp(torch.rand(n, device=dev, generator=g) * float(np.log(2e5))).long()
    elif kind == 3:
        d = torch.randint(4_000_000_000, 2 ** 32, (n,), device=dev, generator=g)
    else:
        d = torch.tensor([7, 8, 100, 109, 110, 111, 1000, 1100], device=dev)[torch.randint(0, 8, (n,), device=dev, generator=g)]
    d, order = torch.sort(d, stable=True)
    d32 = torch.where(d >= 2 ** 31, d - 2 ** 32, d).to(torch.int32)  # the u32 bit pattern in torch's int32
    return w[order].contiguous(), d32.contiguous(), kind


n_bad = 0
for case in range(a.cases):
    n = int(np.exp(rng.uniform(np.log(1000), np.log(a.max_n))))
    tol = int(rng.choice([0, 100, 250, 300, 350, 350, 400]))
    w, d, kind = database(n, tol)
    nr = int(rng.integers(1, max(2, n // 20)))
    ridx = torch.randint(0, n, (nr,), device=dev, generator=g)
    rw, rd = w[ridx].contiguous(), d[ridx].contiguous()
    torch.cuda.synchronize()
    res = {}
    for name, env in (("mfma", {"VDF_SEARCH_BACKEND": "mfma"}), ("valu", {"VDF_SEARCH_BACKEND": "valu"})):
        eng = engine_with(env, devices=[0])
        offs, members = eng.search_self_shards([w.data_ptr()], [d.data_ptr()], [n], tol, as_arrays=True)
        eng.close()
        eng1 = engine_with(env, device=0)
        hits, nh = eng1.search_refs_device(w.data_ptr(), d.data_ptr(), n, rw.data_ptr(), rd.data_ptr(), nr, tol, capacity=1 << 20)
        eng1.close()
        res[name] = (np.asarray(offs).copy(), np.asarray(members).copy(), np.asarray(hits[:nh]).copy())
    same = all(np.array_equal(x, y) for x, y in zip(res["mfma"], res["valu"]))
    n_bad += not same
    print(f"[{case}] n={n} tol={tol} durations kind {kind} refs={nr}: groups {len(res['mfma'][0]) - 1} members {len(res['mfma'][1])} ref hits {len(res['mfma'][2])}: "
          + ("same" if same else "DIFFERENT"), flush=True)
print(f"== {a.cases} cases, {n_bad} differing")
