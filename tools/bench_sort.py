#!/usr/bin/env python3
"""The device-side sorts of a search call on their own: Search::sort order (vdf_sort_order_device, with and without path ranks) and
the (row, col) sort of a dense hit list as dup_heavy makes it.  Every result is checked against numpy's stable sort."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf

eng = vdf.Engine(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
for n in (1000, 100_000, 1_000_000, 10_000_000):
    dur = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n))).astype(np.uint32)
    rank = rng.permutation(n).astype(np.uint32)
    k = len(rank[1::7])
    rank[::7][:k] = rank[1::7]  # equal ranks: stability decides
    d_d, d_r = torch.from_numpy(dur.view(np.int32)).to(dev), torch.from_numpy(rank.view(np.int32)).to(dev)
    perm = torch.zeros(n, dtype=torch.int32, device=dev)
    for with_rank in (False, True):
        want = np.lexsort((rank, dur)) if with_rank else np.argsort(dur, kind="stable")
        eng.sort_order_device(d_d.data_ptr(), n, perm.data_ptr(), d_r.data_ptr() if with_rank else 0)
        torch.cuda.synchronize()
        ok = np.array_equal(perm.cpu().numpy().view(np.uint32), want.astype(np.uint32))
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            eng.sort_order_device(d_d.data_ptr(), n, perm.data_ptr(), d_r.data_ptr() if with_rank else 0)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"sort_order n={n} rank={with_rank}: {'ok' if ok else 'MISMATCH'} median {sorted(ts)[4]:.3f} ms (min {min(ts):.3f}, max {max(ts):.3f})")
