#!/usr/bin/env python3
"""search_self kernel time for several database sizes (all durations equal) - for A/B of tile/chunk defaults."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf
from bench import make_hashes
eng = vdf.Engine(0)
for n in (10_000, 50_000, 100_000, 300_000, 1_000_000):
    words = make_hashes(n, 1)
    dw = torch.from_numpy(words.view(np.int64)).cuda(); dd = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        eng.search_self_device(dw.data_ptr(), dd.data_ptr(), n, 350)
        best = min(best, eng.last_stats()["kernel_ms"])
    st = eng.last_stats()
    print(f"n={n}: kernel {best:.3f} ms, {st['pairs'] / best * 1e3:.3g} pairs/s, {st['n_tiles']} workgroups")
