#!/usr/bin/env python3
"""outputs/ref_hash_out.bin + outputs/ref_search_out.bin (written by ref_vectors.rs inside the real crate) ->
tests/golden/ref_hash.npz + tests/golden/ref_search.npz (what tests/test_reference_vectors.py loads).
ref_hash_out.bin:   u32 n_cases; per case: u32 name_len, name, u32 n_clips; per clip: 16 u64 words, 16*256 u8 resized
                    frames (row-major 16x16 each), 1000 f64 coefficients in bit order (100 kt + 10 kx + ky).
ref_search_out.bin: three group lists (search 0.35, search 0.1, search_with_references 0.35); each: u32 n_groups; per
                    group: i64 reference index (-1 = none), u32 n_members, members as u32 indices.
ref_resize_tables.bin (resize_tables.rs): u32 n_sizes; per axis size: u32 n, u32 n_rand; f32 impulse[n][16] (weight of source x in output o);
                    then u8 [rows][16] blocks: constants 0..255 horizontal, the same vertical, step edges k = 0..n horizontal, vertical,
                    the n_rand random rows horizontal, vertical  ->  tests/golden/ref_resize_tables.npz."""
import os
import struct

import numpy as np

HERE = os.environ.get("VDF_VECTORS_DIR") or os.path.dirname(os.path.abspath(__file__))
G = os.environ.get("VDF_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                                      "tests", "golden")


def rd(f, fmt):
    return struct.unpack(fmt, f.read(struct.calcsize(fmt)))


def import_resize_tables():
    p = os.path.join(HERE, "outputs", "ref_resize_tables.bin")
    if not os.path.exists(p):
        return
    t = {}
    with open(p, "rb") as f:
        (n_sizes,) = rd(f, "<I")
        sizes = []
        for _ in range(n_sizes):
            n, n_rand = rd(f, "<II")
            sizes.append(n)
            t[f"s{n}_impulse"] = np.frombuffer(f.read(n * 16 * 4), "<f4").reshape(n, 16).copy()
            for key, rows in (("const_h", 256), ("const_v", 256), ("step_h", n + 1), ("step_v", n + 1), ("rand_h", n_rand), ("rand_v", n_rand)):
                t[f"s{n}_{key}"] = np.frombuffer(f.read(rows * 16), np.uint8).reshape(rows, 16).copy()
        assert f.read() == b"", "trailing bytes in ref_resize_tables.bin"
        t["sizes"] = np.array(sizes, np.uint32)
    np.savez_compressed(os.path.join(G, "ref_resize_tables.npz"), **t)
    print("wrote ref_resize_tables.npz into", G)


import_resize_tables()
out = {}
if not os.path.exists(os.path.join(HERE, "outputs", "ref_hash_out.bin")):  # only the cache dump (or the table dump) was run
    if not os.path.exists(os.path.join(HERE, "outputs", "ref_cache.bin")):
        raise SystemExit(0)
    cache = os.path.join(HERE, "outputs", "ref_cache.bin")
    import shutil

    shutil.copyfile(cache, os.path.join(G, "ref_cache.bin"))
    print("copied ref_cache.bin into", G)
    raise SystemExit(0)
with open(os.path.join(HERE, "outputs", "ref_hash_out.bin"), "rb") as f:
    (n_cases,) = rd(f, "<I")
    for _ in range(n_cases):
        (ln,) = rd(f, "<I")
        name = f.read(ln).decode()
        (n_clips,) = rd(f, "<I")
        words = np.zeros((n_clips, 16), np.uint64)
        small = np.zeros((n_clips, 16, 16, 16), np.uint8)
        coefs = np.zeros((n_clips, 1000), np.float64)
        for c in range(n_clips):
            words[c] = np.frombuffer(f.read(128), "<u8")
            small[c] = np.frombuffer(f.read(4096), np.uint8).reshape(16, 16, 16)
            coefs[c] = np.frombuffer(f.read(8000), "<f8")
        out[name + "_hashes"], out[name + "_resized"], out[name + "_coefs"] = words, small, coefs
np.savez_compressed(os.path.join(G, "ref_hash.npz"), **out)

res = {}
with open(os.path.join(HERE, "outputs", "ref_search_out.bin"), "rb") as f:
    for key in ("self350", "self100", "refs350"):
        (ng,) = rd(f, "<I")
        offs, mem, refs = [0], [], []
        for _ in range(ng):
            (r,) = rd(f, "<q")
            (nm,) = rd(f, "<I")
            mem += list(rd(f, f"<{nm}I"))
            offs.append(len(mem))
            refs.append(r)
        res[key + "_offsets"] = np.array(offs, np.uint64)
        res[key + "_members"] = np.array(mem, np.uint64)
        res[key + "_index"] = np.array(refs, np.int64)
np.savez_compressed(os.path.join(G, "ref_search.npz"), **res)
print("wrote ref_hash.npz, ref_search.npz into", G)
cache = os.path.join(HERE, "outputs", "ref_cache.bin")  # written by cache_dump.rs with the app's own cache writer
if os.path.exists(cache):
    import shutil

    shutil.copyfile(cache, os.path.join(G, "ref_cache.bin"))
    print("copied ref_cache.bin into", G)
