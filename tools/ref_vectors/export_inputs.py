#!/usr/bin/env python3
"""Flat little-endian copies of the committed fixture INPUTS for the Rust harness (ref_vectors.rs).
hash_inputs.bin:   u32 n_cases, then per case: u32 name_len, name, u32 n_clips, n_frames, h, w, then n_clips*n_frames*h*w bytes.
search_inputs.bin: u32 n, n*16 u64 hash words, n u32 durations, u32 n_ref, n_ref*16 u64, n_ref u32.
resize_probe_inputs.bin: u32 n_sizes, then per axis size: u32 n, u32 n_rand, n_rand * n random bytes (resize_tables.rs: the rows / columns
                   it probes fast_image_resize's U8 path with; the constant, step and impulse probes need no input).
cache_inputs.bin:  u32 n, then per entry: u32 kind (0 Ok, 1 Err(NotVideo), 2 Err(VidProc), 3 Err(NotEnoughFrames)), 16 u64 words,
                   u32 duration, u64 mtime secs, u32 mtime nanos, u32 path_len, path (UTF-8), u32 msg_len, msg (cache_dump.rs)."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")
OUT = os.path.join(os.environ.get("VDF_VECTORS_DIR") or HERE, "inputs")
os.makedirs(OUT, exist_ok=True)

z = np.load(os.path.join(G, "hash_golden.npz"))
names = sorted(k[: -len("_frames")] for k in z.files if k.endswith("_frames"))
with open(os.path.join(OUT, "hash_inputs.bin"), "wb") as f:
    f.write(struct.pack("<I", len(names)))
    for name in names:
        fr = np.ascontiguousarray(z[name + "_frames"], dtype=np.uint8)
        nb = name.encode()
        f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<IIII", *fr.shape))
        f.write(fr.tobytes())
s = np.load(os.path.join(G, "search_golden.npz"))
with open(os.path.join(OUT, "search_inputs.bin"), "wb") as f:
    for h, d in ((s["hashes"], s["durations"]), (s["ref_hashes"], s["ref_durations"])):
        f.write(struct.pack("<I", len(d)))
        f.write(np.ascontiguousarray(h, dtype="<u8").tobytes())
        f.write(np.ascontiguousarray(d, dtype="<u4").tobytes())
# axis sizes whose tables the crate is asked for: every width / height of the hash fixtures, the headline 64, the common video sizes, and
# sizes around the 64-pixel tile boundary and the upscaling side of 16 (tests/test_reference_vectors.py: RESIZE_PROBE_SIZES)
RESIZE_PROBE_SIZES = sorted({int(d) for name in names for d in z[name + "_frames"].shape[2:]} | {8, 12, 17, 48, 63, 64, 65, 90, 128, 270, 360, 480, 640,
                                                                                                  720, 1080, 1280, 1920})
RESIZE_PROBE_RAND = 8
rng = np.random.default_rng(20251004)
with open(os.path.join(OUT, "resize_probe_inputs.bin"), "wb") as f:
    f.write(struct.pack("<I", len(RESIZE_PROBE_SIZES)))
    for n in RESIZE_PROBE_SIZES:
        f.write(struct.pack("<II", n, RESIZE_PROBE_RAND))
        f.write(rng.integers(0, 256, size=(RESIZE_PROBE_RAND, n), dtype=np.uint8).tobytes())
import sys

sys.path.insert(0, HERE)
from cache_cases import cache_cases  # noqa: E402

cases = cache_cases()
with open(os.path.join(OUT, "cache_inputs.bin"), "wb") as f:
    f.write(struct.pack("<I", len(cases)))
    for path, kind, words, dur, secs, nanos, msg in cases:
        pb, mb = path.encode(), msg.encode()
        f.write(struct.pack("<I", kind) + np.ascontiguousarray(words, dtype="<u8").tobytes() + struct.pack("<IQI", dur, secs, nanos))
        f.write(struct.pack("<I", len(pb)) + pb + struct.pack("<I", len(mb)) + mb)
print("wrote", OUT, "cases:", names, "+", len(cases), "cache entries")
