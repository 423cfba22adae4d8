"""The reference app's on-disk hash cache <-> SoA arrays: thin wrappers over vdf_cache_decode / vdf_cache_encode
(csrc/cache_format.cpp, which documents the bincode layout), the metadata sidecar (csrc/cache_metadata.cpp; cache_metadata.rs),
Search::sort's path order for a whole cache (csrc/path_order.cpp) and the one-call route from cache bytes to MatchGroups
(csrc/cache_search.cpp)."""
from __future__ import annotations

import ctypes as C
import os
from collections.abc import Sequence
from dataclasses import dataclass

import numpy as np

from . import _capi
from ._capi import HASH_WORDS, VdfCacheMetadata, VdfCacheSearchTiming, VdfCacheSoa, VdfError, VdfGroups

DEFAULT_VID_HASH_SKIP_FORWARD = 15.0  # vid_dup_finder_lib/src/definitions.rs:18 (CreationOptions::default, video_hash_builder.rs:55-63)


class PathTable(Sequence):
    """The paths of a decoded cache: one UTF-8 blob + offsets, decoded entry by entry on access (a million Python strings cost more
    than the whole decode; a search needs the paths of the grouped members only).  Compares equal to a list of the same strings."""

    __slots__ = ("_blob", "_offs")

    def __init__(self, blob, offsets: np.ndarray):
        """blob: bytes, or a u8 array viewing the decoder's buffer (no copy of the path bytes)."""
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
        return bytes(self._blob[int(self._offs[i]):int(self._offs[i + 1])]).decode("utf-8")

    def __iter__(self):
        b, o = bytes(self._blob), self._offs.tolist()
        return (b[o[k]:o[k + 1]].decode("utf-8") for k in range(len(o) - 1))

    def __eq__(self, other):
        if isinstance(other, (PathTable, list, tuple)):
            return len(self) == len(other) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __repr__(self):
        return f"PathTable({len(self)} paths)"

    @property
    def blob(self):
        return self._blob

    @property
    def offsets(self) -> np.ndarray:
        return self._offs


def decode_cache(data, n_threads: int = 0):
    """bytes of a cache file -> dict(hashes [n,16] u64, durations [n] u32, paths (PathTable: a lazy sequence of n str), mtime_secs,
    mtime_nanos, n_entries, n_err, n_key_differs).  Entries holding Err(..) are counted in n_err and skipped.
    data: bytes, or anything with the buffer protocol (a numpy u8 array, an mmap) - read in place.  n_threads: 0 = all host threads
    for files of a few MB and more (vdf_cache_decode_mt)."""
    lib = _capi.load()
    soa = VdfCacheSoa()
    if isinstance(data, bytes):
        ptr, n = (C.cast(C.c_char_p(data), C.c_void_p) if data else None), len(data)
    else:
        view = np.frombuffer(data, dtype=np.uint8)
        ptr, n = (view.ctypes.data if view.size else None), int(view.size)
    rc = lib.vdf_cache_decode_mt(ptr, n, int(n_threads), C.byref(soa))  # read in place: no copy
    if rc:
        raise VdfError(rc, "malformed cache file")
    # The arrays are views of the decoder's own buffers (no second copy of 128 B per entry): an owner object frees them when the last
    # view is gone.
    owner = _SoaOwner(lib, soa)
    n = int(soa.n_ok)
    hashes = owner.view(soa.hashes, C.c_uint64, n * HASH_WORDS, np.uint64).reshape(n, HASH_WORDS)
    durs = owner.view(soa.durations, C.c_uint32, n, np.uint32)
    offs = owner.view(soa.path_offsets, C.c_uint64, n + 1, np.uint64)
    blob_len = int(offs[-1]) if n else 0
    blob = owner.view(soa.paths, C.c_char, blob_len, np.uint8) if blob_len else np.zeros(0, np.uint8)  # a view too: 600 MB at 10 M entries
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


# ---- the metadata sidecar (cache_metadata.rs; video_hash_filesystem_cache.rs:76-139) ---------------------------------------
_CROP_NAMES = {"none": _capi.VDF_CROPDETECT_NONE, "letterbox": _capi.VDF_CROPDETECT_LETTERBOX, "motion": _capi.VDF_CROPDETECT_MOTION}


def _crop_code(cropdetect) -> int:
    """Cropdetect enum member (api.Cropdetect), its name, or the C code."""
    if isinstance(cropdetect, int):
        return cropdetect
    name = getattr(cropdetect, "value", cropdetect)
    return _CROP_NAMES[str(name).lower()]


class CacheMetadataError(VdfError):
    """VdfCacheError::MetadataValidationError: the sidecar does not parse or does not match the run's options."""


@dataclass
class CacheMetadata:
    """VdfCacheMetadata (cache_metadata.rs:45-51); the codes are include/vdf.h's VDF_CACHE_OS_* / VDF_CACHE_BACKEND_* / VDF_CROPDETECT_*."""
    operating_system: int
    decode_backend: int
    crop: int
    skip_forward_amount: float
    cache_version: int

    def _c(self) -> VdfCacheMetadata:
        return VdfCacheMetadata(self.operating_system, self.decode_backend, self.crop, 0, self.skip_forward_amount, self.cache_version)

    @classmethod
    def _from_c(cls, m: VdfCacheMetadata) -> "CacheMetadata":
        return cls(int(m.operating_system), int(m.decode_backend), int(m.crop), float(m.skip_forward_amount), int(m.cache_version))

    @classmethod
    def new(cls, cropdetect="letterbox", skip_forward_amount: float = DEFAULT_VID_HASH_SKIP_FORWARD) -> "CacheMetadata":
        """VdfCacheMetadata::new (cache_metadata.rs:54-78): Unix, FfmpegBackend, cache_version 1."""
        m = VdfCacheMetadata()
        rc = _capi.load().vdf_cache_metadata_new(_crop_code(cropdetect), float(skip_forward_amount), C.byref(m))
        if rc:
            raise VdfError(rc, "bad cropdetect")
        return cls._from_c(m)

    def to_disk_fmt(self) -> str:
        """cache_metadata.rs:80-89."""
        buf = C.create_string_buffer(512)
        n = C.c_size_t(0)
        m = self._c()
        rc = _capi.load().vdf_cache_metadata_format(C.byref(m), buf, len(buf), C.byref(n))
        if rc:
            raise VdfError(rc, "metadata fields out of range")
        return buf.raw[: n.value].decode("utf-8")

    @classmethod
    def try_parse(cls, text) -> "CacheMetadata":
        """cache_metadata.rs:91-125; CacheMetadataError carries the app's message."""
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        m = VdfCacheMetadata()
        err = C.create_string_buffer(1024)
        rc = _capi.load().vdf_cache_metadata_parse(raw, len(raw), C.byref(m), err, len(err))
        if rc:
            raise CacheMetadataError(rc, err.value.decode("utf-8", "replace"))
        return cls._from_c(m)

    def validate(self, exp_cropdetect="letterbox", exp_skip_forward_amount: float = DEFAULT_VID_HASH_SKIP_FORWARD) -> None:
        """cache_metadata.rs:127-168: raises CacheMetadataError naming the first field that differs."""
        err = C.create_string_buffer(1024)
        m = self._c()
        rc = _capi.load().vdf_cache_metadata_validate(C.byref(m), _crop_code(exp_cropdetect), float(exp_skip_forward_amount), err, len(err))
        if rc:
            raise CacheMetadataError(rc, err.value.decode("utf-8", "replace"))


def metadata_path(cache_path) -> str:
    """<dir>/<stem>.metadata.txt for a cache file path (video_hash_filesystem_cache.rs:93-104)."""
    raw = os.fsencode(cache_path)
    buf = C.create_string_buffer(len(raw) + 32)
    n = C.c_size_t(0)
    rc = _capi.load().vdf_cache_metadata_path(raw, len(raw), buf, len(buf), C.byref(n))
    if rc:
        raise VdfError(rc, f"{cache_path!r} has no file name")
    return os.fsdecode(buf.raw[: n.value])


def write_cache_files(cache_path, hashes, durations, paths: Sequence[str], mtime_secs=None, mtime_nanos=None, cropdetect="letterbox",
                      skip_forward_amount: float = DEFAULT_VID_HASH_SKIP_FORWARD) -> None:
    """A cache the app loads: the bincode file AND its sidecar (without the sidecar the app exits before it reads a byte,
    video_hash_filesystem_cache.rs:113-117).  cropdetect / skip_forward_amount must be what the hashes were made with."""
    data = encode_cache(hashes, durations, paths, mtime_secs, mtime_nanos)
    with open(cache_path, "wb") as f:
        f.write(data)
    with open(metadata_path(cache_path), "w", encoding="utf-8") as f:
        f.write(CacheMetadata.new(cropdetect, skip_forward_amount).to_disk_fmt())


def load_cache_files(cache_path, cropdetect="letterbox", skip_forward_amount: float = DEFAULT_VID_HASH_SKIP_FORWARD, n_threads: int = 0):
    """What VideoHashFilesystemCache::new does before it trusts a cache (video_hash_filesystem_cache.rs:76-139): the sidecar must be there
    (the app exits otherwise: FileNotFoundError here), parse, and match this run's crop detection / skip - only then decode.  A cache
    hashed with Cropdetect::None is refused when the engine is about to hash with letterbox detection, and the reverse."""
    mp = metadata_path(cache_path)
    if not os.path.exists(mp):
        raise FileNotFoundError(f"Cache exists but metadata is absent: {mp}")
    with open(mp, "r", encoding="utf-8") as f:
        CacheMetadata.try_parse(f.read()).validate(cropdetect, skip_forward_amount)
    return decode_cache(np.fromfile(cache_path, dtype=np.uint8), n_threads)


# ---- Search::sort's path order and the one-call search ------------------------------------------------------------------
def path_ranks(paths, n_threads: int = 0) -> np.ndarray:
    """u32 rank of every path in PathBuf (component-wise) order, equal paths sharing a rank (vdf_path_ranks).  paths: a PathTable
    (blob + offsets, no per-entry objects) or a sequence of str / bytes."""
    if isinstance(paths, PathTable):
        blob, offs = paths.blob, np.ascontiguousarray(paths.offsets, dtype=np.uint64)
        blob = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, bytes) else np.ascontiguousarray(blob, dtype=np.uint8)
    else:
        enc = [p if isinstance(p, bytes) else os.fsencode(p) for p in paths]
        offs = np.zeros(len(enc) + 1, np.uint64)
        offs[1:] = np.cumsum([len(e) for e in enc]) if enc else []
        blob = np.frombuffer(b"".join(enc), dtype=np.uint8)
    n = len(offs) - 1
    out = np.zeros(n, np.uint32)
    if n == 0:
        return out
    keep = blob if blob.size else np.zeros(1, np.uint8)
    rc = _capi.load().vdf_path_ranks(keep.ctypes.data, offs.ctypes.data, n, out.ctypes.data, int(n_threads))
    if rc:
        raise VdfError(rc, "vdf_path_ranks failed")
    return out


def sort_order_paths(engine, durations, paths):
    """Search::sort's order (search_algorithm.rs:55-61: stable by (duration, PathBuf order of the path)) of entries given as host arrays
    (vdf_sort_order_paths): the path half runs on the device when every path is plain, through the host's component comparator
    otherwise.  -> (order [n] u32: order[k] = the entry at position k, used_device: bool).  paths: PathTable or a sequence of str / bytes."""
    import ctypes as C

    if isinstance(paths, PathTable):
        blob, offs = paths.blob, np.ascontiguousarray(paths.offsets, dtype=np.uint64)
        blob = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, bytes) else np.ascontiguousarray(blob, dtype=np.uint8)
    else:
        enc = [p if isinstance(p, bytes) else os.fsencode(p) for p in paths]
        offs = np.zeros(len(enc) + 1, np.uint64)
        offs[1:] = np.cumsum([len(e) for e in enc]) if enc else []
        blob = np.frombuffer(b"".join(enc), dtype=np.uint8)
    n = len(offs) - 1
    dur = np.ascontiguousarray(durations, dtype=np.uint32)
    assert dur.shape == (n,)
    out = np.zeros(n, np.uint32)
    used = C.c_int(0)
    keep = blob if blob.size else np.zeros(1, np.uint8)
    engine._check(engine.lib.vdf_sort_order_paths(engine.ctx, dur.ctypes.data, offs.ctypes.data, keep.ctypes.data, n, out.ctypes.data, C.byref(used)))
    return out, bool(used.value)


def path_compare(a, b) -> int:
    a = a if isinstance(a, bytes) else os.fsencode(a)
    b = b if isinstance(b, bytes) else os.fsencode(b)
    return int(_capi.load().vdf_path_compare(a, len(a), b, len(b)))


def search_cache_arrays(cache: dict, tolerance: float, engine=None, cand_idx=None, ref_idx=None):
    """decode_cache's dict -> (offsets u64[g + 1], members u64[m], ref_index i64[g], timing dict): search() over the selected entries
    (cand_idx None = all), or search_with_references() when ref_idx is given; members / ref_index index the cache's arrays.
    One C call: PathBuf ranks -> upload -> Search::sort on the device -> search (vdf_search_cache_entries)."""
    from .api import default_engine
    from .engine import groups_to_arrays, tolerance_int

    eng = engine or default_engine()
    h = np.ascontiguousarray(cache["hashes"], dtype=np.uint64).reshape(-1, HASH_WORDS)
    d = np.ascontiguousarray(cache["durations"], dtype=np.uint32)
    pt: PathTable = cache["paths"]
    blob = pt.blob
    blob = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, bytes) else np.ascontiguousarray(blob, dtype=np.uint8)
    if blob.size == 0:
        blob = np.zeros(1, np.uint8)
    offs = np.ascontiguousarray(pt.offsets, dtype=np.uint64)
    ci = None if cand_idx is None else np.ascontiguousarray(cand_idx, dtype=np.uint64)
    ri = None if ref_idx is None else np.ascontiguousarray(ref_idx, dtype=np.uint64)
    g = VdfGroups()
    t = VdfCacheSearchTiming()
    eng._check(eng.lib.vdf_search_cache_entries(eng.ctx, h.ctypes.data, d.ctypes.data, offs.ctypes.data, blob.ctypes.data, len(d),
                                                ci.ctypes.data if ci is not None and len(ci) else None, len(ci) if ci is not None else 0,
                                                ri.ctypes.data if ri is not None and len(ri) else None, len(ri) if ri is not None else 0,
                                                tolerance_int(tolerance), C.byref(g), C.byref(t)))
    try:
        offsets, members, refs = groups_to_arrays(g)
    finally:
        eng.lib.vdf_groups_free(C.byref(g))
    return offsets, members, refs, {k: float(getattr(t, k)) for k, _ in VdfCacheSearchTiming._fields_}


def search_cache(data, tolerance: float, engine=None, cand_idx=None, ref_idx=None):
    """Cache bytes (or decode_cache's dict) -> List[MatchGroup] of paths, equal to vdf.search(video_hashes_from_cache(data), tolerance)
    (resp. search_with_references for ref_idx) without one Python object per ENTRY: only the grouped members' paths are materialised."""
    from .api import MatchGroup

    cache = data if isinstance(data, dict) else decode_cache(data)
    if cand_idx is not None and len(cand_idx) == 0:
        return []
    offsets, members, refs, _ = search_cache_arrays(cache, tolerance, engine, cand_idx, ref_idx)
    paths = cache["paths"]
    offs, mem = offsets.tolist(), members.tolist()
    out = []
    for k in range(len(offs) - 1):
        dups = [paths[m] for m in mem[offs[k]:offs[k + 1]]]
        out.append(MatchGroup(paths[int(refs[k])] if refs[k] >= 0 else None, dups))
    return out
