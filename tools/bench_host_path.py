#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points (never the headline `value`): vdf_hash_frames_u8 and
vdf_search_self called with pageable numpy memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vid_dup_finder_lib_amd as vdf
from bench import make_hashes

eng = vdf.Engine(0)
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, size=(20000, 16, 64, 64), dtype=np.uint8)
eng.hash_frames(frames[:100])
eng.hash_frames(frames)  # allocations (pinned staging, device batch buffers) happen on the first full-size call
dts = []
for _ in range(5):
    t0 = time.perf_counter(); eng.hash_frames(frames); dts.append(time.perf_counter() - t0)
dt = min(dts)
print("  all runs (ms):", " ".join(f"{x*1e3:.1f}" for x in dts), "| VDF_COPY_THREADS", os.environ.get("VDF_COPY_THREADS"), "VDF_HOST_CHUNK_MB",
      os.environ.get("VDF_HOST_CHUNK_MB"), "VDF_HOST_DIRECT", os.environ.get("VDF_HOST_DIRECT"))
print(f"vdf_hash_frames_u8 (host, pageable): {len(frames)} clips in {dt*1e3:.1f} ms = {len(frames)*16/dt:.4g} frames/s, {frames.nbytes/dt/1e9:.1f} GB/s over PCIe")
if os.environ.get("HOST_PATH_SKIP_SEARCH"):
    sys.exit(0)
w = make_hashes(1_000_000, 20250613); d = np.zeros(len(w), np.uint32)
eng.search_self_sorted(w[:1000], d[:1000], 350)
t0 = time.perf_counter(); g = eng.search_self_sorted(w, d, 350); dt = time.perf_counter() - t0
print(f"vdf_search_self (host arrays): 1 M hashes in {dt*1e3:.1f} ms = {len(w)*(len(w)-1)/2/dt:.4g} pairs/s incl. 128 MB upload ({len(g)} groups)")
