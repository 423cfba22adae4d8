import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf
from vid_dup_finder_lib_amd import engine as ve
from bench import make_hashes
n = 1_000_000
words = make_hashes(n, 20250613)
dw = torch.from_numpy(words.view(np.int64)).cuda(); dd = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
eng = vdf.Engine(0)
for it in range(3):
    t0 = time.perf_counter()
    hits, n_hits, ov = eng.search_self_device(dw.data_ptr(), dd.data_ptr(), n, 350)
    t1 = time.perf_counter()
    matched = np.zeros(n, np.uint8)
    g = ve.replay_self(n, hits, matched, 0, n)
    t2 = time.perf_counter()
    groups = ve.finish_self(g)
    t3 = time.perf_counter()
    st = eng.last_stats()
    print(f"device call {1e3*(t1-t0):.2f} ms (kernel {st['kernel_ms']:.2f}), replay {1e3*(t2-t1):.2f}, finish {1e3*(t3-t2):.2f}, hits {n_hits}, groups {len(groups)}")
