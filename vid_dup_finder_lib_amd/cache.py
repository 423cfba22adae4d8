"""The reference app's on-disk hash cache <-> SoA arrays; thin wrapper over
vdf_cache_decode / vdf_cache_encode (csrc/cache_format.cpp, which documents the bincode layout)."""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence

import numpy as np

from . import _capi
from ._capi import HASH_WORDS, VdfCacheSoa, VdfError


class PathTable(Sequence):
    """The paths of a decoded cache: one UTF-8 blob + offsets, decoded entry by entry on access (a million Python strings cost more
    than the whole decode; a search needs the paths of the grouped members only).  Compares equal to a list of the same strings."""

    __slots__ = ("_blob", "_offs")

    def __init__(self, blob: bytes, offsets: np.ndarray):
        self._blob, self._offs = blob, offsets

    def __len__(self):
        return len(self._offs) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._blob[int(self._offs[i]):int(self._offs[i + 1])].decode("utf-8")

    def __iter__(self):
        b, o = self._blob, self._offs.tolist()
        return (b[o[k]:o[k + 1]].decode("utf-8") for k in range(len(o) - 1))

    def __eq__(self, other):
        if isinstance(other, (PathTable, list, tuple)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __repr__(self):
        return f"PathTable({len(self)} paths)"

    @property
    def blob(self) -> bytes:
        return self._blob

    @property
    def offsets(self) -> np.ndarray:
        return self._offs


def decode_cache(data: bytes):
    """bytes of a cache file -> dict(hashes [n,16] u64, durations [n] u32, paths (PathTable: a lazy sequence of n str), mtime_secs,
    mtime_nanos, n_entries, n_err, n_key_differs).  Entries holding Err(..) are counted in n_err and skipped."""
    lib = _capi.load()
    soa = VdfCacheSoa()
    data = bytes(data) if not isinstance(data, bytes) else data
    rc = lib.vdf_cache_decode(C.cast(C.c_char_p(data), C.c_void_p) if data else None, len(data), C.byref(soa))  # read in place: no copy
    if rc:
        raise VdfError(rc, "malformed cache file")
    # The arrays are views of the decoder's own buffers (no second copy of 128 B per entry): an owner object frees them when the last
    # view is gone.
    owner = _SoaOwner(lib, soa)
    n = int(soa.n_ok)
    hashes = owner.view(soa.hashes, C.c_uint64, n * HASH_WORDS, np.uint64).reshape(n, HASH_WORDS)
    durs = owner.view(soa.durations, C.c_uint32, n, np.uint32)
    offs = owner.view(soa.path_offsets, C.c_uint64, n + 1, np.uint64)
    blob = C.string_at(soa.paths, int(offs[-1])) if n else b""
    secs = owner.view(soa.mtime_secs, C.c_uint64, n, np.uint64)
    nanos = owner.view(soa.mtime_nanos, C.c_uint32, n, np.uint32)
    return {"hashes": hashes, "durations": durs, "paths": PathTable(blob, offs), "mtime_secs": secs, "mtime_nanos": nanos,
            "n_entries": int(soa.n_entries), "n_err": int(soa.n_err), "n_key_differs": int(soa.n_key_differs)}


class _SoaOwner:
    """Keeps a decoded vdf_cache_soa alive while numpy views of its arrays exist (each view's base holds a reference)."""

    def __init__(self, lib, soa):
        self._lib, self._soa = lib, soa

    def view(self, ptr, ctype, count, dtype):
        if count == 0:
            return np.zeros(0, dtype)
        raw = (ctype * count).from_address(C.addressof(ptr.contents))
        raw._owner = self  # the ctypes array becomes the numpy array's base
        return np.frombuffer(raw, dtype=dtype)

    def __del__(self):
        try:
            self._lib.vdf_cache_free(C.byref(self._soa))
        except Exception:  # interpreter shutdown
            pass


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
