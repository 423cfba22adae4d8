#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself cannot run here (Rust, no cargo) and holds no
known-answer vector for this path, so these vectors pin the ORACLE's behaviour over time and give the GPU box
a reference-free fixture; the C oracle is cross-checked against its independent numpy/scipy twin while
generating.  Fixtures are data only: inputs + expected outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import hashgen as hg  # noqa: E402
from oracle import vdf_oracle as orc  # noqa: E402


def csr(groups):
    offs = np.cumsum([0] + [len(g) for g in groups]).astype(np.uint64)
    mem = np.array([m for g in groups for m in g], dtype=np.uint64)
    return offs, mem


def main():
    rng = np.random.default_rng(20250617)
    out = {}
    # "static64": clips whose coefficients are mathematically zero in bulk - 16 identical frames (every kt >= 1 coefficient),
    # constant frames (every AC coefficient), frames mirrored left-right (every odd kx) or top-bottom (every odd ky).  Their
    # bits are defined by the split-radix structure of rustdct's butterflies (exact +-0.0 -> bit 0), not by rounding noise.
    srng = np.random.default_rng(20250619)
    one = srng.integers(0, 256, size=(64, 64), dtype=np.uint8)
    mir = srng.integers(0, 256, size=(16, 64, 32), dtype=np.uint8)
    static = np.stack([np.broadcast_to(one, (16, 64, 64)), np.full((16, 64, 64), 200, np.uint8),
                       np.full((16, 64, 64), 17, np.uint8), np.full((16, 64, 64), 128, np.uint8),
                       np.concatenate([mir, mir[:, :, ::-1]], axis=2),
                       np.concatenate([mir, mir[:, :, ::-1]], axis=2).transpose(0, 2, 1)]).astype(np.uint8)
    for name, shape in (("f64", (6, 16, 64, 64)), ("f16", (16, 16, 16, 16)), ("f33x47", (2, 16, 33, 47)),
                        ("static64", static.shape)):
        frames = static if name == "static64" else rng.integers(0, 256, size=shape, dtype=np.uint8)
        hashes, coefs = orc.hash_clips_with_coefs(frames)
        # cross-check against the numpy twin (independent resize + vectorised split-radix DCT: same operation order, so
        # the coefficients must agree bit for bit) and against scipy's DCT (a different algorithm: values to 1e-8)
        from scipy.fft import dctn
        for c in range(shape[0]):
            small = np.stack([orc.np_resize_frame(f) for f in frames[c]])
            assert np.array_equal(small, np.stack([orc.resize_frame(f) for f in frames[c]]))
            tw, tc = orc.np_hash_frames16(small, want_coefs=True)
            assert np.array_equal(tc, coefs[c]) and np.array_equal(tw, hashes[c])
            cube = np.transpose(small.astype(np.float64), (0, 2, 1)) - 128.0
            assert np.allclose((dctn(cube, type=2) / 8.0)[:10, :10, :10].reshape(-1), coefs[c], atol=1e-8)
        out[name + "_frames"] = frames
        out[name + "_hashes"] = hashes
        out[name + "_dontcare"] = np.packbits(np.abs(coefs) < 1e-6, axis=1)
    np.savez_compressed(os.path.join(HERE, "hash_golden.npz"), **out)

    rng = np.random.default_rng(20250613)
    words, dur = hg.planted_set(rng, 2000, n_clusters=60, max_copies=6, durations="windowed")
    w, d, _ = hg.sort_by_duration(words, dur)
    g350 = orc.search_self_sorted(w, d, 350)
    g100 = orc.search_self_sorted(w, d, 100)
    pick = rng.choice(len(d), size=150, replace=False)
    rw, rd = w[pick].copy(), d[pick].copy()
    for i in range(0, len(rw), 2):
        bits = np.unpackbits(rw[i].view(np.uint8), bitorder="little")
        bits[rng.choice(1024, size=int(rng.integers(0, 360)), replace=False)] ^= 1
        rw[i] = np.packbits(bits, bitorder="little").view(np.uint64)
    refs = orc.search_refs_sorted(w, d, rw, rd, 350)
    o350, m350 = csr(g350)
    o100, m100 = csr(g100)
    ro, rm = csr([m for _, m in refs])
    np.savez_compressed(os.path.join(HERE, "search_golden.npz"), hashes=w, durations=d, ref_hashes=rw,
                        ref_durations=rd, self350_offsets=o350, self350_members=m350, self100_offsets=o100,
                        self100_members=m100, refs350_offsets=ro, refs350_members=rm,
                        refs350_index=np.array([r for r, _ in refs], np.int64))
    print("groups:", len(g350), len(g100), "ref groups:", len(refs))


if __name__ == "__main__":
    main()
