#!/usr/bin/env python3
"""Host phases of the cache route at the scale it exists for: a synthetic app cache of n entries -> vdf_cache_decode_mt ->
vdf_path_ranks, timed per thread count.  No GPU needed (the search half is bench.py's cache_ingest leg).
    python tools/bench_cache_ingest.py --entries 2000000 --threads 1,2,4,8,0
The cache bytes come from vdf_cache_encode over arrays built with numpy (no per-entry Python strings): paths like
/srv/media/lib_07/show_0412/season_03/clip_00001234.mkv - shared directory prefixes, as a real library has."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synth_paths(n, seed=20250620):
    """(blob u8, offsets u64[n + 1]): n fixed-width paths in random order over ~n/40 directories."""
    rng = np.random.default_rng(seed)
    tmpl = np.frombuffer(b"/srv/media/lib_00/show_0000/season_00/clip_00000000.mkv", np.uint8)
    L = len(tmpl)
    blob = np.tile(tmpl, (n, 1))
    ids = rng.permutation(n).astype(np.int64)
    show = ids // 40
    fields = ((15, 2, show // 2000 % 100), (23, 4, show % 2000 * 5 % 10000), (35, 2, ids // 8 % 5), (43, 8, ids))
    for at, width, val in fields:
        v = val.copy()
        for k in range(width - 1, -1, -1):
            blob[:, at + k] = 48 + v % 10
            v //= 10
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    return blob.reshape(-1), offs


def synth_cache(n, seed=20250620):
    from vid_dup_finder_lib_amd import _capi

    lib = _capi.load()
    rng = np.random.default_rng(seed)
    hashes = rng.integers(0, 2**64, size=(n, 16), dtype=np.uint64)
    hashes[:, 15] &= np.uint64((1 << 40) - 1)
    dur = np.floor(np.exp(rng.uniform(np.log(5), np.log(7200), size=n))).astype(np.uint32)
    blob, offs = synth_paths(n, seed)
    secs = rng.integers(1_600_000_000, 1_760_000_000, size=n, dtype=np.uint64)
    nanos = rng.integers(0, 10**9, size=n, dtype=np.uint32)
    out, out_len = C.c_void_p(), C.c_size_t()
    rc = lib.vdf_cache_encode(n, hashes.ctypes.data, dur.ctypes.data, offs.ctypes.data, blob.ctypes.data, secs.ctypes.data,
                              nanos.ctypes.data, C.byref(out), C.byref(out_len))
    assert rc == 0
    data = np.ctypeslib.as_array((C.c_uint8 * out_len.value).from_address(out.value)).copy()
    lib.vdf_buffer_free(out)
    return data, hashes, dur, blob, offs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entries", type=int, default=2_000_000)
    ap.add_argument("--threads", default="1,2,4,8,0")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from vid_dup_finder_lib_amd import _capi
    from vid_dup_finder_lib_amd._capi import VdfCacheSoa

    lib = _capi.load()
    t0 = time.perf_counter()
    data, hashes, dur, blob, offs = synth_cache(args.entries)
    print(f"{args.entries} entries, {data.size / 1e6:.0f} MB of cache bytes, generated in {time.perf_counter() - t0:.1f} s; "
          f"{os.cpu_count()} host threads")
    for nt in [int(x) for x in args.threads.split(",")]:
        best = 1e9
        for _ in range(args.reps):
            soa = VdfCacheSoa()
            t0 = time.perf_counter()
            rc = lib.vdf_cache_decode_mt(data.ctypes.data, data.size, nt, C.byref(soa))
            dt = time.perf_counter() - t0
            assert rc == 0 and soa.n_ok == args.entries
            best = min(best, dt)
            lib.vdf_cache_free(C.byref(soa))
        print(f"decode   threads={nt or 'auto':>4}: {best * 1e3:8.1f} ms  ({data.size / best / 1e9:.2f} GB/s)")
    rank = np.zeros(args.entries, np.uint32)
    for nt in [int(x) for x in args.threads.split(",")]:
        best = 1e9
        for _ in range(max(1, args.reps - 1)):
            t0 = time.perf_counter()
            rc = lib.vdf_path_ranks(blob.ctypes.data, offs.ctypes.data, args.entries, rank.ctypes.data, nt)
            dt = time.perf_counter() - t0
            assert rc == 0
            best = min(best, dt)
        print(f"pathrank threads={nt or 'auto':>4}: {best * 1e3:8.1f} ms  ({best / args.entries * 1e6:.3f} us per entry)")


if __name__ == "__main__":
    main()
