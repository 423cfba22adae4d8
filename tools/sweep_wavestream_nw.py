#!/usr/bin/env python3
"""Per-wave stream kernel: waves per workgroup (VDF_WAVESTREAM_NW = 4, 5, 6, 8) against the default dispatch, per frame width.
TB/s of frame bytes, frames resident in HBM, hashes checked against the default's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vid_dup_finder_lib_amd as vdf

dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
sizes = [(256, 144), (320, 180), (426, 240), (480, 270), (640, 360), (768, 432), (854, 480), (896, 504), (960, 540), (1024, 576), (1152, 648), (1280, 720),
         (1366, 768), (1440, 810), (1536, 864), (1600, 900), (1680, 1050), (1792, 1008), (1904, 1071), (1920, 1080)]
if len(sys.argv) > 1:
    sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
print("   size      | default |   NW=4 |   NW=5 |   NW=6 |   NW=8   (TB/s)")
for w, h in sizes:
    n = max(32, int(2.0e9 / (16 * w * h)))
    g = torch.Generator(device=dev); g.manual_seed(w)
    frames = torch.randint(0, 256, (n, 16, h, w), dtype=torch.uint8, device=dev, generator=g)
    row, ref = [], None
    for nw in (0, 4, 5, 6, 8):
        if nw: os.environ["VDF_WAVESTREAM_NW"] = str(nw)
        else: os.environ.pop("VDF_WAVESTREAM_NW", None)
        eng = vdf.Engine(0)
        out = torch.zeros((n, 16), dtype=torch.int64, device=dev)
        try:
            eng.hash_frames_device(frames.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                eng.hash_frames_device(frames.data_ptr(), n, 16, w, h, out.data_ptr(), stream=st.cuda_stream)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 4
            if ref is None: ref = out.clone()
            ok = torch.equal(ref, out)
            row.append(f"{n * 16 * w * h / ms / 1e9:6.2f}{'' if ok else '!'}")
        except vdf.VdfError as ex:
            row.append("   err")
        eng.close()
    os.environ.pop("VDF_WAVESTREAM_NW", None)
    print(f"{w:5d} x {h:4d}: " + " | ".join(row), flush=True)
