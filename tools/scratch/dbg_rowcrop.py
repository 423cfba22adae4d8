import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import vid_dup_finder_lib_amd as vdf
from oracle import vdf_oracle as orc
eng = vdf.Engine(0)
def run(h, w, crops, n, tag):
    rng = np.random.default_rng(h + w)
    frames = rng.integers(0, 256, size=(n, 16, h, w), dtype=np.uint8)
    d = torch.from_numpy(frames).cuda()
    out = torch.zeros((n, 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    eng.hash_frames_cropped_device(d.data_ptr(), n, 16, w, h, crops, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    bad = []
    for c in range(n):
        l, r, t, b = (int(x) for x in crops[c])
        rc, want = orc.hash_clip(np.ascontiguousarray(frames[c][:, t:h - b, l:w - r]))[:2]
        if not np.array_equal(got[c], want):
            bad.append((c, int(sum(bin(int(x)).count("1") for x in (got[c] ^ want)))))
    print(tag, "bad clips (index, bits):", bad, flush=True)
for n in (24, 48, 6):
    crops = np.tile(np.array((0, 0, 31, 17), np.uint32), (n, 1)); crops[::3] = 0
    run(300, 720, crops, n, f"mixed n={n}")
    crops = np.tile(np.array((0, 0, 31, 17), np.uint32), (n, 1))
    run(300, 720, crops, n, f"all cropped n={n}")
    crops = np.zeros((n, 4), np.uint32); crops[1] = (0, 0, 1, 0)
    run(300, 720, crops, n, f"all but one uncropped n={n}")
