#!/usr/bin/env python3
"""search_with_references at the BASELINE configs[4] shape (1 M candidates x 100 k references, log-uniform durations,
+-5 % windows, tolerance 350): kernel time and waste ratio (pairs the tiles evaluated / pairs the windows admit) for
the knobs in the environment (VDF_MFMA_REFS_ROWS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vid_dup_finder_lib_amd as vdf

n_cand, n_ref = 1_000_000, 100_000
rng = np.random.default_rng(20250615)
cw = rng.integers(0, 2**64, size=(n_cand, 16), dtype=np.uint64); cw[:, 15] &= np.uint64((1 << 40) - 1)
cd = np.sort(np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n_cand))).astype(np.uint32))
src = rng.choice(n_cand, size=n_ref // 2, replace=False)
rw = np.concatenate([cw[src].copy(), rng.integers(0, 2**64, size=(n_ref - n_ref // 2, 16), dtype=np.uint64)])
rd = np.concatenate([cd[src], np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n_ref - n_ref // 2))).astype(np.uint32)])
perm = rng.permutation(n_ref); rw, rd = rw[perm], rd[perm]
t = [torch.from_numpy(a).cuda() for a in (cw.view(np.int64), cd.view(np.int32), rw.view(np.int64), rd.view(np.int32))]
torch.cuda.synchronize()
eng = vdf.Engine(0)
import time
ks, wall = [], []
for i in range(6):
    t0 = time.perf_counter()
    hits, n_hits = eng.search_refs_device(t[0].data_ptr(), t[1].data_ptr(), n_cand, t[2].data_ptr(), t[3].data_ptr(), n_ref, 350)
    if i: wall.append(time.perf_counter() - t0)
    st = eng.last_stats()
    if i: ks.append(st["kernel_ms"])
print(f"refs_rows {os.environ.get('VDF_MFMA_REFS_ROWS','256')}: call {np.mean(wall) * 1e3:.2f} ms, kernel_ms mean {np.mean(ks):.3f} min {np.min(ks):.3f}, "
      f"pairs {st['pairs']:.4g}, computed {st['pairs_computed']:.4g}, waste ratio {st['pairs_computed']/st['pairs']:.3f}, hits {n_hits}, tiles {st['n_tiles']}")
