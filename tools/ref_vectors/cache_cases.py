"""The entries of the hash-cache hand-off (cache_dump.rs writes them with the APP's own cache writer; tests/test_cache_format.py
compares what vdf_cache_decode reads from that file with this list).  Deterministic, no RNG state outside this function."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")

KIND_OK, KIND_NOT_VIDEO, KIND_VID_PROC, KIND_NOT_ENOUGH_FRAMES = 0, 1, 2, 3


def cache_cases():
    """[(path, kind, words[16] u64, duration, mtime_secs, mtime_nanos, message)]"""
    s = np.load(os.path.join(G, "search_golden.npz"))
    words, dur = s["hashes"], s["durations"]
    rng = np.random.default_rng(20250620)
    out = []
    for i in range(200):
        d = int(dur[i]) if i % 7 else int(rng.choice([0, 250, 251, 65535, 65536, 2**32 - 1]))  # varint boundaries
        path = f"/videos/dir{i % 5}/sub.{i % 3}/clip é{i}.mkv" if i % 4 else f"rel/clip{i}.webm"
        secs = int(rng.integers(0, 2**40)) if i % 5 else int(rng.choice([0, 250, 251, 2**32]))
        out.append((path, KIND_OK, words[i].astype(np.uint64), d, secs, int(rng.integers(0, 10**9)), ""))
    out.append(("/videos/full.mp4", KIND_OK, np.full(16, 0xFFFFFFFFFFFFFFFF, np.uint64), 7, 1, 2, ""))  # padding bits set
    out.append(("/videos/empty.mp4", KIND_OK, np.zeros(16, np.uint64), 0, 0, 0, ""))
    out.append(("/videos/not_a_video.txt", KIND_NOT_VIDEO, np.zeros(16, np.uint64), 0, 11, 12, ""))
    out.append(("/videos/broken.avi", KIND_VID_PROC, np.zeros(16, np.uint64), 0, 13, 14, "ffmpeg: moov atom not found é"))
    out.append(("/videos/short.gif", KIND_NOT_ENOUGH_FRAMES, np.zeros(16, np.uint64), 0, 15, 16, ""))
    return out
