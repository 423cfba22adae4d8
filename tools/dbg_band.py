import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf
n = int(os.environ.get("N", "300000")); tol = int(os.environ.get("TOL", "470"))
rng = np.random.default_rng(1)
words = rng.integers(0, 2**64, size=(n, 16), dtype=np.uint64); words[:, 15] &= np.uint64((1 << 40) - 1)
d_w = torch.from_numpy(words.view(np.int64)).cuda(); d_d = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
eng = vdf.Engine(0)
lo, hi = 1000, 1064
band, nh, ov = eng.search_self_device(d_w.data_ptr(), d_d.data_ptr(), n, tol, row_begin=lo, row_end=hi)
st = eng.last_stats()
want = set()
for i in range(lo, hi):
    dist = np.unpackbits((words[i + 1:] ^ words[i]).view(np.uint8), axis=1).sum(axis=1)
    want |= {(i, i + 1 + int(j)) for j in np.nonzero(dist <= tol)[0]}
got = {(int(a), int(b)) for a, b in band}
missing = sorted(want - got); extra = sorted(got - want)
print("n", n, "tol", tol, "want", len(want), "got", len(got), "missing", len(missing), "extra", len(extra), "n_hits", nh, "overflow", hex(ov), st)
if missing:
    m = np.array(missing)
    print("missing cols min/max", m[:,1].min(), m[:,1].max())
    print("by chunk(65536):", np.unique(m[:,1] // 65536, return_counts=True))
    print("by row-32 block:", np.unique((m[:,0] - (m[:,0]//512)*512) // 32, return_counts=True))
    print("by sub (col%128//32):", np.unique(m[:,1] % 128 // 32, return_counts=True))
    print("by lane (col%32):", np.unique(m[:,1] % 32, return_counts=True)[1])
    print("by reg-row (row%32):", np.unique(m[:,0] % 32, return_counts=True))
    g = np.array(sorted(got)); print("got cols per chunk:", np.unique(g[:,1]//65536, return_counts=True))
    for c in np.unique(m[:,1]//65536)[:3]:
        mm = m[m[:,1]//65536 == c]; gg = g[g[:,1]//65536 == c]
        print("chunk", c, "missing col range", mm[:,1].min(), mm[:,1].max(), "got col range", gg[:,1].min() if len(gg) else None, gg[:,1].max() if len(gg) else None)
