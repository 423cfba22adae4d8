#!/usr/bin/env python3
"""Host phases of the cache route at the scale it exists for: a synthetic app cache of n entries -> vdf_cache_decode_mt ->
vdf_path_ranks, timed per thread count.  No GPU needed (the search half is bench.py's cache_ingest leg).
    python tools/bench_cache_ingest.py --entries 2000000 --threads 1,2,4,8,0
The cache bytes come from bench.py's synth_cache (vdf_cache_encode over arrays built with numpy, no per-entry Python strings)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


from bench import synth_cache, synth_paths  # noqa: E402,F401  (one generator for the tool and bench.py's cache_ingest leg)


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
    data, hashes, dur, blob, offs, _ = synth_cache(args.entries)
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
