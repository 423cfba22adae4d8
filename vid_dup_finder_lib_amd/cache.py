"""The reference app's on-disk hash cache <-> SoA arrays; thin wrapper over
vdf_cache_decode / vdf_cache_encode (csrc/cache_format.cpp, which documents the bincode layout)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from ._capi import HASH_WORDS, VdfCacheSoa, VdfError


def decode_cache(data: bytes):
    """bytes of a cache file -> dict(hashes [n,16] u64, durations [n] u32, paths [n] str, mtime_secs, mtime_nanos,
    n_entries, n_err, n_key_differs).  Entries holding Err(..) are counted in n_err and skipped."""
    lib = _capi.load()
    soa = VdfCacheSoa()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else None
    rc = lib.vdf_cache_decode(buf, len(data), C.byref(soa))
    if rc:
        raise VdfError(rc, "malformed cache file")
    try:
        n = int(soa.n_ok)
        hashes = np.ctypeslib.as_array(soa.hashes, shape=(n, HASH_WORDS)).copy() if n else np.zeros((0, HASH_WORDS), np.uint64)
        durs = np.ctypeslib.as_array(soa.durations, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        offs = np.ctypeslib.as_array(soa.path_offsets, shape=(n + 1,)).copy()
        blob = C.string_at(soa.paths, int(offs[-1])) if n else b""
        paths = [blob[int(offs[i]):int(offs[i + 1])].decode("utf-8") for i in range(n)]
        secs = np.ctypeslib.as_array(soa.mtime_secs, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
        nanos = np.ctypeslib.as_array(soa.mtime_nanos, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        return {"hashes": hashes, "durations": durs, "paths": paths, "mtime_secs": secs, "mtime_nanos": nanos,
                "n_entries": int(soa.n_entries), "n_err": int(soa.n_err), "n_key_differs": int(soa.n_key_differs)}
    finally:
        lib.vdf_cache_free(C.byref(soa))


def encode_cache(hashes, durations, paths: Sequence[str], mtime_secs=None, mtime_nanos=None) -> bytes:
    """SoA -> bytes the app's BaseFsCache::load_cache_from_disk accepts (every entry Ok, key = src_path)."""
    lib = _capi.load()
    h = np.ascontiguousarray(hashes, dtype=np.uint64).reshape(-1, HASH_WORDS)
    d = np.ascontiguousarray(durations, dtype=np.uint32)
    n = len(d)
    enc = [p.encode("utf-8") for p in paths]
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc]) if n else []
    blob = b"".join(enc)
    ms = np.ascontiguousarray(mtime_secs, dtype=np.uint64) if mtime_secs is not None else None
    mn = np.ascontiguousarray(mtime_nanos, dtype=np.uint32) if mtime_nanos is not None else None
    out_p, out_len = C.c_void_p(), C.c_size_t(0)
    rc = lib.vdf_cache_encode(n, h.ctypes.data, d.ctypes.data, offs.ctypes.data, blob,
                              ms.ctypes.data if ms is not None else None, mn.ctypes.data if mn is not None else None,
                              C.byref(out_p), C.byref(out_len))
    if rc:
        raise VdfError(rc, "vdf_cache_encode failed")
    try:
        return C.string_at(out_p, out_len.value)
    finally:
        lib.vdf_buffer_free(out_p)


def video_hashes_from_cache(data: bytes):
    """Decode straight into VideoHash objects (API mirror convenience; the SoA form is what scales)."""
    from .api import VideoHash

    c = decode_cache(data)
    return [VideoHash(c["hashes"][i], c["paths"][i], int(c["durations"][i])) for i in range(len(c["paths"]))]
