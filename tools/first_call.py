#!/usr/bin/env python3
"""What a ONE-SHOT caller pays (the app runs one search per process): context creation, the first host-level search() over 1 M
hashes (lazy allocations, pinned staging, upload) and the second one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process)
import vid_dup_finder_lib_amd as vdf
from bench import make_hashes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = make_hashes(n, 20250613)
d = np.zeros(n, np.uint32)
t0 = time.perf_counter(); eng = vdf.Engine(0); t1 = time.perf_counter()
g = eng.search_self_sorted(w, d, 350); t2 = time.perf_counter()
tm1 = eng.last_timing()
g = eng.search_self_sorted(w, d, 350); t3 = time.perf_counter()
tm2 = eng.last_timing()
print(f"n={n}: ctx {1e3*(t1-t0):.1f} ms, first search {1e3*(t2-t1):.1f} ms (library total {tm1['total_ms']:.1f}), second {1e3*(t3-t2):.1f} ms (library total {tm2['total_ms']:.1f}); {len(g)} groups")
