#!/usr/bin/env python3
"""Projected multi-GPU balance measured on ONE GPU: run every shard of the N-GPU bench problem in turn."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf
from bench import make_hashes

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=8)
a = ap.parse_args()
n = int(round(1_000_000 * a.gpus ** 0.5))
words = make_hashes(n, 20250613)
dw = torch.from_numpy(words.view(np.int64)).cuda(); dd = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
eng = vdf.Engine(0)
eng.search_self_device(dw.data_ptr(), dd.data_ptr(), n, 350, shard_index=0, shard_count=a.gpus)
ms, pairs, hits = [], [], 0
for r in range(a.gpus):
    h, nh, ov = eng.search_self_device(dw.data_ptr(), dd.data_ptr(), n, 350, shard_index=r, shard_count=a.gpus)
    st = eng.last_stats()
    ms.append(st["kernel_ms"]); pairs.append(st["pairs"]); hits += nh
print(f"n={n} shards={a.gpus}: kernel ms per shard {[round(m,1) for m in ms]}")
print(f"pairs per shard min/max = {min(pairs):.4g}/{max(pairs):.4g} (imbalance {max(pairs)/np.mean(pairs)-1:.3%}); "
      f"sum = {sum(pairs)} == n(n-1)/2 = {n*(n-1)//2}: {sum(pairs)==n*(n-1)//2}; hits {hits}")
print(f"projected {a.gpus}-GPU rate = {sum(pairs)/max(ms)*1e3:.4g} pairs/s (kernel only; + all-gather of {n*128/1e6:.0f} MB)")
