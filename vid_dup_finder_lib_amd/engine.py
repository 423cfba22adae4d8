"""Engine: one vdf_ctx (one GPU) behind numpy-friendly methods.

Thin host plumbing over the C ABI (include/vdf.h).  All arithmetic of the hot path runs in
libvdf_hip.so on the GPU; nothing here computes a hash bit or a Hamming distance on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from ._capi import HASH_WORDS, VdfError, VdfGroups, VdfHit, VdfSearchStats, VdfSearchTiming

UINT32_MAX = 0xFFFFFFFF


def _groups_to_lists(g: VdfGroups) -> List[Tuple[int, List[int]]]:
    offsets, members, refs = groups_to_arrays(g)
    offs = offsets.tolist()
    mem = members.tolist()
    return [(int(refs[i]), mem[offs[i]:offs[i + 1]]) for i in range(len(offs) - 1)]


def groups_to_arrays(g: VdfGroups):
    """(offsets u64[n+1], members u64[m], ref_index i64[n]) copies of a vdf_groups."""
    n = int(g.n_groups)
    offsets = np.ctypeslib.as_array(g.offsets, shape=(n + 1,)).copy() if g.offsets else np.zeros(1, np.uint64)
    m = int(offsets[-1])
    members = np.ctypeslib.as_array(g.members, shape=(m,)).copy() if m else np.zeros(0, np.uint64)
    refs = np.ctypeslib.as_array(g.ref_index, shape=(n,)).copy() if n else np.zeros(0, np.int64)
    return offsets, members, refs


def _ptr_array(ptrs: Sequence[int]):
    return (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p) or None) for p in ptrs])


def _size_array(sizes: Sequence[int]):
    return (C.c_size_t * len(sizes))(*[int(x) for x in sizes])


def _atoi(text: str) -> int:
    """C's atoi: optional blanks, optional sign, leading digits; anything else counts as 0 (how csrc/api.cpp reads its switches)."""
    import re

    m = re.match(r"\s*([+-]?\d+)", text or "")
    return int(m.group(1)) if m else 0


class Engine:
    """One context: one GPU (`device`, default LOCAL_RANK or 0), or - `devices=[...]` - ONE context over several GPUs
    of the node (vdf_ctx_create_multi: the host-array calls fan out inside the library, the *_shards methods take
    one device pointer per GPU).  A device may be listed twice to exercise the multi-GPU path on one GPU.

    *_device methods take raw device pointers and a hipStream_t handle (single-GPU contexts only).  stream=0 means the
    context's own non-blocking stream, which does NOT order against work queued elsewhere (e.g. torch's current
    stream): either pass the stream that produced the buffers, or synchronise before the call."""

    def __init__(self, device: Optional[int] = None, devices: Optional[Sequence[int]] = None):
        self.lib = _capi.load()
        ctx = C.c_void_p()
        if devices is not None:
            devs = [int(d) for d in devices]
            arr = (C.c_int * len(devs))(*devs)
            rc = self.lib.vdf_ctx_create_multi(arr, len(devs), C.byref(ctx))
            device = devs[0] if devs else 0
        else:
            if device is None:
                device = int(os.environ.get("LOCAL_RANK", "0"))
            rc = self.lib.vdf_ctx_create(int(device), C.byref(ctx))
        if rc != _capi.VDF_OK:
            msg = self.lib.vdf_last_error(None)
            raise VdfError(rc, (msg or b"").decode() or "vdf_ctx_create failed (is a GPU visible?)")
        self.ctx = ctx
        # the library read VDF_NO_HIT_FILTER when the context was made: without the filter a replay launch makes no exchange calls
        self.hit_filter_enabled = _atoi(os.environ.get("VDF_NO_HIT_FILTER", "0")) == 0
        self.device = int(device)
        self.n_devices = int(self.lib.vdf_ctx_device_count(ctx))
        self.devices = [int(self.lib.vdf_ctx_device_at(ctx, k)) for k in range(self.n_devices)]

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vdf_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _hit_buffer(self, capacity: int) -> np.ndarray:
        """Reusable host staging buffer for hit lists (a fresh 32 MB allocation per call costs ~10 ms)."""
        buf = getattr(self, "_hits", None)
        if buf is None or buf.shape[0] < max(capacity, 1):
            buf = np.empty((max(capacity, 1), 2), np.uint32)
            self._hits = buf
        return buf

    def _check(self, rc: int):
        if rc != _capi.VDF_OK:
            raise VdfError(rc, (self.lib.vdf_last_error(self.ctx) or b"").decode())

    def set_hit_capacity(self, capacity: int):
        self._check(self.lib.vdf_ctx_set_hit_capacity(self.ctx, int(capacity)))

    def last_stats(self) -> dict:
        s = VdfSearchStats()
        self._check(self.lib.vdf_ctx_last_search_stats(self.ctx, C.byref(s)))
        return {k: getattr(s, k) for k, _ in VdfSearchStats._fields_}

    def last_timing(self) -> dict:
        """Where the time of the last search call went (vdf_search_timing: ms per phase, suspect-queue fill)."""
        t = VdfSearchTiming()
        self._check(self.lib.vdf_ctx_last_search_timing(self.ctx, C.byref(t)))
        return {k: getattr(t, k) for k, _ in VdfSearchTiming._fields_}

    def device_stats(self, slot: int) -> dict:
        s = VdfSearchStats()
        self._check(self.lib.vdf_ctx_device_search_stats(self.ctx, int(slot), C.byref(s)))
        return {k: getattr(s, k) for k, _ in VdfSearchStats._fields_}

    def rccl_ranks(self) -> int:
        """RCCL communicators (one per GPU) this context has initialised (0: every replication so far was plain device copies)."""
        return int(self.lib.vdf_ctx_rccl_ranks(self.ctx))

    def device_timing(self, slot: int) -> dict:
        t = VdfSearchTiming()
        self._check(self.lib.vdf_ctx_device_search_timing(self.ctx, int(slot), C.byref(t)))
        return {k: getattr(t, k) for k, _ in VdfSearchTiming._fields_}

    # ------------------------------------------------------- multi-GPU contexts: device-resident shards
    def search_self_shards(self, d_hash_shards: Sequence[int], d_dur_shards: Sequence[int], shard_n: Sequence[int],
                           tol_int: int, as_arrays: bool = False):
        """search() over a sorted database cut into consecutive shards, shard k resident on the GPU of slot k (device
        pointers); the library replicates it with one all-gather (RCCL over xGMI) and searches on all GPUs.
        as_arrays: the CSR form (offsets, members) instead of Python lists (100 k members cost 2 ms to convert)."""
        g = VdfGroups()
        self._check(self.lib.vdf_search_self_shards(self.ctx, _ptr_array(d_hash_shards), _ptr_array(d_dur_shards),
                                                    _size_array(shard_n), int(tol_int), C.byref(g)))
        try:
            if as_arrays:
                return groups_to_arrays(g)[:2]
            return [m for _, m in _groups_to_lists(g)]
        finally:
            self.lib.vdf_groups_free(C.byref(g))

    def search_refs_shards(self, d_cand_hash_shards, d_cand_dur_shards, cand_shard_n, d_ref_hash_shards, d_ref_dur_shards,
                           ref_shard_n, tol_int: int):
        g = VdfGroups()
        self._check(self.lib.vdf_search_refs_shards(self.ctx, _ptr_array(d_cand_hash_shards), _ptr_array(d_cand_dur_shards),
                                                    _size_array(cand_shard_n), _ptr_array(d_ref_hash_shards),
                                                    _ptr_array(d_ref_dur_shards), _size_array(ref_shard_n), int(tol_int),
                                                    C.byref(g)))
        try:
            return _groups_to_lists(g)
        finally:
            self.lib.vdf_groups_free(C.byref(g))

    def hash_frames_shards(self, d_frames: Sequence[int], n_clips: Sequence[int], frames_per_clip: int, w: int, h: int,
                           d_out: Sequence[int], d_dontcare: Optional[Sequence[int]] = None):
        fs = w * h
        self._check(self.lib.vdf_hash_frames_u8_shards(self.ctx, _ptr_array(d_frames), _size_array(n_clips), frames_per_clip,
                                                       w, h, fs, fs * frames_per_clip, _ptr_array(d_out),
                                                       _ptr_array(d_dontcare) if d_dontcare is not None else None))

    # ------------------------------------------------------------------ hashing
    def hash_frames(self, frames: np.ndarray, want_dontcare: bool = False):
        """frames [n_clips, n_frames, H, W] u8 (host) -> hashes [n_clips, 16] u64.
        Raises VdfError(VDF_E_NOT_ENOUGH_FRAMES) when n_frames < 16 (video_hash.rs:53,61)."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim != 4:
            raise ValueError("frames must be [n_clips, n_frames, H, W]")
        nc, nf, h, w = frames.shape
        out = np.zeros((nc, HASH_WORDS), np.uint64)
        dc = np.zeros(nc, np.uint32) if want_dontcare else None
        rc = self.lib.vdf_hash_frames_u8(self.ctx, frames.ctypes.data, nc, nf, w, h, w * h, nf * w * h,
                                         out.ctypes.data, dc.ctypes.data if want_dontcare else None)
        self._check(rc)
        return (out, dc) if want_dontcare else out

    def hash_frames_device(self, d_frames: int, n_clips: int, frames_per_clip: int, w: int, h: int, d_out: int,
                           d_dontcare: int = 0, frame_stride: Optional[int] = None,
                           clip_stride: Optional[int] = None, stream: int = 0):
        fs = w * h if frame_stride is None else frame_stride
        cs = fs * frames_per_clip if clip_stride is None else clip_stride
        self._check(self.lib.vdf_hash_frames_u8_device(self.ctx, d_frames, n_clips, frames_per_clip, w, h, fs, cs,
                                                       d_out, d_dontcare or None, stream or None))

    def hash_frames_letterbox(self, frames: np.ndarray, want_dontcare: bool = False):
        """crop_video_frames(Cropdetect::Letterbox) + from_frames (video_hash_builder.rs:188-223) for a batch:
        frames [n_clips, n_frames >= 16, H, W] u8 -> (hashes [n_clips, 16] u64, crops [n_clips, 4] u32 = l, r, t, b
        [, dontcare])."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim != 4:
            raise ValueError("frames must be [n_clips, n_frames, H, W]")
        nc, nf, h, w = frames.shape
        out = np.zeros((nc, HASH_WORDS), np.uint64)
        crops = np.zeros((nc, 4), np.uint32)
        dc = np.zeros(nc, np.uint32) if want_dontcare else None
        self._check(self.lib.vdf_hash_frames_u8_letterbox(self.ctx, frames.ctypes.data, nc, nf, w, h, w * h, nf * w * h,
                                                          out.ctypes.data, crops.ctypes.data,
                                                          dc.ctypes.data if want_dontcare else None))
        return (out, crops, dc) if want_dontcare else (out, crops)

    def cropdetect_letterbox_device(self, d_frames: int, n_clips: int, frames_per_clip: int, w: int, h: int,
                                    d_crops: int, stream: int = 0, frame_stride: Optional[int] = None,
                                    clip_stride: Optional[int] = None):
        fs = w * h if frame_stride is None else frame_stride
        cs = fs * frames_per_clip if clip_stride is None else clip_stride
        self._check(self.lib.vdf_cropdetect_letterbox_device(self.ctx, d_frames, n_clips, frames_per_clip, w, h, fs, cs,
                                                             d_crops, stream or None))

    def hash_frames_cropped_device(self, d_frames: int, n_clips: int, frames_per_clip: int, w: int, h: int,
                                   crops: Optional[np.ndarray], d_out: int, d_dontcare: int = 0, stream: int = 0):
        c = None if crops is None else np.ascontiguousarray(crops, dtype=np.uint32).reshape(n_clips, 4)
        self._check(self.lib.vdf_hash_frames_u8_cropped_device(self.ctx, d_frames, n_clips, frames_per_clip, w, h, w * h,
                                                               w * h * frames_per_clip,
                                                               c.ctypes.data if c is not None else None, d_out,
                                                               d_dontcare or None, stream or None))

    def hash_frames_letterbox_device(self, d_frames: int, n_clips: int, frames_per_clip: int, w: int, h: int,
                                     d_out: int, d_dontcare: int = 0, stream: int = 0,
                                     frame_stride: Optional[int] = None, clip_stride: Optional[int] = None,
                                     d_crops: Optional[int] = None) -> Optional[np.ndarray]:
        """Cropdetect::Letterbox + from_frames on device frames.  Default: returns the boxes as a host array (the call then ends with a
        wait for its own work).  d_crops = device pointer to n_clips x 4 uint32 (0: the boxes are not wanted): the boxes stay on the
        device, ordered on `stream` like the hashes, and the call only queues work (vdf_hash_frames_u8_letterbox_device_async)."""
        fs = w * h if frame_stride is None else frame_stride
        cs = fs * frames_per_clip if clip_stride is None else clip_stride
        if d_crops is not None:
            self._check(self.lib.vdf_hash_frames_u8_letterbox_device_async(self.ctx, d_frames, n_clips, frames_per_clip, w, h, fs, cs,
                                                                           d_out, d_dontcare or None, d_crops or None, stream or None))
            return None
        crops = np.zeros((n_clips, 4), np.uint32)
        self._check(self.lib.vdf_hash_frames_u8_letterbox_device(self.ctx, d_frames, n_clips, frames_per_clip, w, h,
                                                                 fs, cs, d_out, d_dontcare or None, crops.ctypes.data,
                                                                 stream or None))
        return crops

    # ------------------------------------------------------------------- search
    def search_self_sorted(self, hashes, durations, tol_int: int) -> List[List[int]]:
        """search() on SoA input already in Search::sort order; groups of sorted indices."""
        h = np.ascontiguousarray(hashes, dtype=np.uint64).reshape(-1, HASH_WORDS)
        d = np.ascontiguousarray(durations, dtype=np.uint32)
        assert h.shape[0] == d.shape[0]
        g = VdfGroups()
        self._check(self.lib.vdf_search_self(self.ctx, h.ctypes.data, d.ctypes.data, len(d), int(tol_int), C.byref(g)))
        try:
            return [m for _, m in _groups_to_lists(g)]
        finally:
            self.lib.vdf_groups_free(C.byref(g))

    def search_refs_sorted(self, cand_hashes, cand_durations, ref_hashes, ref_durations, tol_int: int):
        """search_with_references() on sorted candidates; [(ref_input_index, [candidate indices])]."""
        ch = np.ascontiguousarray(cand_hashes, dtype=np.uint64).reshape(-1, HASH_WORDS)
        cd = np.ascontiguousarray(cand_durations, dtype=np.uint32)
        rh = np.ascontiguousarray(ref_hashes, dtype=np.uint64).reshape(-1, HASH_WORDS)
        rd = np.ascontiguousarray(ref_durations, dtype=np.uint32)
        g = VdfGroups()
        self._check(self.lib.vdf_search_refs(self.ctx, ch.ctypes.data, cd.ctypes.data, len(cd), rh.ctypes.data,
                                             rd.ctypes.data, len(rd), int(tol_int), C.byref(g)))
        try:
            return _groups_to_lists(g)
        finally:
            self.lib.vdf_groups_free(C.byref(g))

    def groups_max_distance(self, hashes, groups: Sequence[Sequence[int]], ref_hashes=None,
                            ref_index: Optional[Sequence[int]] = None) -> np.ndarray:
        """The app's Sorting::Distance key (search_output.rs:43-60): per group, the max Hamming distance over all
        pairs of its members (indices into `hashes`) plus, if given, its reference ref_hashes[ref_index[g]]."""
        h = np.ascontiguousarray(hashes, dtype=np.uint64).reshape(-1, HASH_WORDS)
        ng = len(groups)
        offs = np.zeros(ng + 1, np.uint64)
        offs[1:] = np.cumsum([len(g) for g in groups]) if ng else []
        mem = np.array([m for g in groups for m in g], dtype=np.uint64)
        if not len(mem):
            mem = np.zeros(1, np.uint64)
        g = VdfGroups()
        g.n_groups = ng
        g.offsets = offs.ctypes.data_as(C.POINTER(C.c_uint64))
        g.members = mem.ctypes.data_as(C.POINTER(C.c_uint64))
        rh = ri = None
        if ref_hashes is not None and ref_index is not None:
            rh = np.ascontiguousarray(ref_hashes, dtype=np.uint64).reshape(-1, HASH_WORDS)
            ri = np.ascontiguousarray(ref_index, dtype=np.int64)
            g.ref_index = ri.ctypes.data_as(C.POINTER(C.c_int64))
        out = np.zeros(ng, np.uint32)
        self._check(self.lib.vdf_groups_max_distance(self.ctx, h.ctypes.data, len(h),
                                                     rh.ctypes.data if rh is not None else None,
                                                     len(rh) if rh is not None else 0, C.byref(g), out.ctypes.data))
        return out

    def pin_database(self, d_hashes: int, n: int):
        """Promise that the n x 16 words at d_hashes stay unchanged (until pin_database(0, 0)): searches against exactly this
        database reuse its operand expansion."""
        self._check(self.lib.vdf_ctx_pin_database(self.ctx, d_hashes or None, int(n)))

    def sort_order_device(self, d_durations: int, n: int, d_perm_out: int, d_path_rank: int = 0, stream: int = 0):
        """Search::sort's permutation (stable by (duration, path rank)) of n device-resident entries into d_perm_out (u32)."""
        self._check(self.lib.vdf_sort_order_device(self.ctx, d_durations, d_path_rank or None, n, d_perm_out, stream or None))

    def apply_order_device(self, d_hashes: int, d_durations: int, d_perm: int, n: int, d_hashes_out: int,
                           d_durations_out: int = 0, stream: int = 0):
        self._check(self.lib.vdf_apply_order_device(self.ctx, d_hashes, d_durations or None, d_perm, n, d_hashes_out,
                                                    d_durations_out or None, stream or None))

    def search_self_device(self, d_hashes: int, d_durations: int, n: int, tol_int: int, shard_index: int = 0,
                           shard_count: int = 1, row_begin: int = 0, row_end: int = UINT32_MAX, d_matched: int = 0,
                           capacity: int = 1 << 22, stream: int = 0):
        """Thresholded adjacency of this shard's row tiles: (hits [k,2] u32 sorted, n_hits, overflow_row)."""
        hits = self._hit_buffer(capacity)  # filled by the library up to n_hits
        n_hits = C.c_uint64(0)
        overflow = C.c_uint32(0)
        self._check(self.lib.vdf_search_self_device(self.ctx, d_hashes, d_durations, n, int(tol_int), shard_index,
                                                    shard_count, row_begin, min(row_end, UINT32_MAX),
                                                    d_matched or None, hits.ctypes.data, capacity, C.byref(n_hits),
                                                    C.byref(overflow), stream or None))
        k = min(int(n_hits.value), capacity)
        return hits[:k].copy(), int(n_hits.value), int(overflow.value)

    def search_self_device_replay(self, d_hashes: int, d_durations: int, n: int, tol_int: int, shard_index: int = 0,
                                  shard_count: int = 1, row_begin: int = 0, row_end: int = UINT32_MAX, d_matched: int = 0,
                                  capacity: int = 1 << 22, stream: int = 0, exchange=None):
        """search_self_device for a caller that only replays the hits (vdf_search_self_device_replay): rows that can never
        become targets are dropped on the device.  `exchange` (sharded launches) is how the shards meet for that:
            exchange.agree(complete: bool, total_hits: int) -> (all_complete, total_hits_of_all_shards)
            exchange.or_bitmap(engine, d_bitmap: int, n_words: int, stream: int)   # OR over the shards, in place, on the device
        Returns (hits [k, 2] u32 sorted, hits kept, overflow_row)."""
        hits = self._hit_buffer(capacity)
        n_hits = C.c_uint64(0)
        overflow = C.c_uint32(0)
        x = None
        errors = []
        if exchange is not None:
            def _agree(_user, p_complete, p_total):
                try:
                    c, t = exchange.agree(bool(p_complete[0]), int(p_total[0]))
                    p_complete[0] = 1 if c else 0
                    p_total[0] = int(t)
                    return 0
                except Exception as e:  # noqa: BLE001 - an exception must not unwind through the C frames
                    errors.append(e)
                    return _capi.VDF_E_INVAL

            def _or(_user, d_bitmap, n_words, strm):
                try:
                    exchange.or_bitmap(self, int(d_bitmap or 0), int(n_words), int(strm or 0))
                    return 0
                except Exception as e:  # noqa: BLE001
                    errors.append(e)
                    return _capi.VDF_E_INVAL

            x = _capi.VdfShardExchange(None, _capi.AGREE_FN(_agree), _capi.OR_BITMAP_FN(_or))
        rc = self.lib.vdf_search_self_device_replay(self.ctx, d_hashes, d_durations, n, int(tol_int), shard_index, shard_count,
                                                    row_begin, min(row_end, UINT32_MAX), d_matched or None, hits.ctypes.data,
                                                    capacity, C.byref(n_hits), C.byref(overflow),
                                                    C.byref(x) if x is not None else None, stream or None)
        if errors:
            raise errors[0]
        self._check(rc)
        k = min(int(n_hits.value), capacity)
        return hits[:k].copy(), int(n_hits.value), int(overflow.value)

    def bitmap_or_device(self, d_dst: int, d_srcs: int, n_words: int, n_srcs: int, stream: int = 0):
        """d_dst[w] |= OR over n_srcs bitmaps of n_words u32 words laid out back to back at d_srcs (device pointers)."""
        self._check(self.lib.vdf_bitmap_or_device(self.ctx, d_dst, d_srcs, int(n_words), int(n_srcs), stream or None))

    def search_refs_device(self, d_cand_hashes: int, d_cand_durations: int, n_cand: int, d_ref_hashes: int,
                           d_ref_durations: int, n_ref: int, tol_int: int, ref_index_base: int = 0,
                           capacity: int = 1 << 22, stream: int = 0):
        """(hits [k,2] u32 sorted by (ref, cand), n_hits).  Grows the buffer once if it was too small."""
        for _ in range(6):
            hits = self._hit_buffer(capacity)
            n_hits = C.c_uint64(0)
            rc = self.lib.vdf_search_refs_device(self.ctx, d_cand_hashes, d_cand_durations, n_cand, d_ref_hashes,
                                                 d_ref_durations, n_ref, int(tol_int), ref_index_base,
                                                 hits.ctypes.data, capacity, C.byref(n_hits), stream or None)
            if rc == _capi.VDF_E_OVERFLOW and int(n_hits.value) > capacity:
                capacity = int(n_hits.value)
                continue
            self._check(rc)
            return hits[: int(n_hits.value)].copy(), int(n_hits.value)
        raise VdfError(_capi.VDF_E_OVERFLOW, "hit buffer overflow")


# ---------------------------------------------------------------- host-only helpers (no GPU needed)
def hamming_distance_words(a, b) -> int:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(HASH_WORDS)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(HASH_WORDS)
    return int(_capi.load().vdf_hamming_u1024(a.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              b.ctypes.data_as(C.POINTER(C.c_uint64))))


def tolerance_int(tolerance: float) -> int:
    return int(_capi.load().vdf_tolerance_int(float(tolerance)))


def count_pairs_self(sorted_durations) -> int:
    d = np.ascontiguousarray(sorted_durations, dtype=np.uint32)
    return int(_capi.load().vdf_count_pairs_self(d.ctypes.data_as(C.POINTER(C.c_uint32)), len(d)))


def count_pairs_refs(sorted_cand_durations, ref_durations) -> int:
    c = np.ascontiguousarray(sorted_cand_durations, dtype=np.uint32)
    r = np.ascontiguousarray(ref_durations, dtype=np.uint32)
    return int(_capi.load().vdf_count_pairs_refs(c.ctypes.data_as(C.POINTER(C.c_uint32)), len(c),
                                                 r.ctypes.data_as(C.POINTER(C.c_uint32)), len(r)))


def replay_self(n: int, hits: np.ndarray, matched: Optional[np.ndarray] = None, row_begin: int = 0,
                row_end: int = UINT32_MAX, groups: Optional[VdfGroups] = None) -> VdfGroups:
    """Host replay of search_self's consumption (search_algorithm.rs:131-170) over hits sorted by (row, col).
    Appends to `groups` (ascending target order); finish with finish_self()."""
    lib = _capi.load()
    hits = np.ascontiguousarray(hits, dtype=np.uint32).reshape(-1, 2)
    g = groups if groups is not None else VdfGroups()
    rc = lib.vdf_replay_self(n, hits.ctypes.data, len(hits), row_begin, min(row_end, UINT32_MAX),
                             matched.ctypes.data if matched is not None else None, C.byref(g))
    if rc:
        raise VdfError(rc, "vdf_replay_self failed")
    return g


def finish_self(groups: VdfGroups) -> List[List[int]]:
    lib = _capi.load()
    rc = lib.vdf_groups_finish_self(C.byref(groups))
    if rc:
        raise VdfError(rc, "vdf_groups_finish_self failed")
    try:
        return [m for _, m in _groups_to_lists(groups)]
    finally:
        lib.vdf_groups_free(C.byref(groups))


def sort_hits(hits: np.ndarray) -> np.ndarray:
    """[k, 2] uint32 hits into (row, col) order (C++ radix sort; np.lexsort needs seconds for 1e7 hits)."""
    hits = np.ascontiguousarray(hits, dtype=np.uint32).reshape(-1, 2)
    if not hits.flags.writeable:
        hits = hits.copy()
    rc = _capi.load().vdf_sort_hits(hits.ctypes.data, len(hits))
    if rc:
        raise VdfError(rc, "vdf_sort_hits failed")
    return hits


def ref_groups_csr(hits: np.ndarray):
    """search_with_references groups as arrays (offsets u64[g + 1], members u64[m], ref_index i64[g]) from sorted hits."""
    lib = _capi.load()
    hits = np.ascontiguousarray(hits, dtype=np.uint32).reshape(-1, 2)
    g = VdfGroups()
    rc = lib.vdf_groups_from_ref_hits(hits.ctypes.data, len(hits), C.byref(g))
    if rc:
        raise VdfError(rc, "vdf_groups_from_ref_hits failed")
    try:
        return groups_to_arrays(g)
    finally:
        lib.vdf_groups_free(C.byref(g))


def groups_from_ref_hits(hits: np.ndarray) -> List[Tuple[int, List[int]]]:
    lib = _capi.load()
    hits = np.ascontiguousarray(hits, dtype=np.uint32).reshape(-1, 2)
    g = VdfGroups()
    rc = lib.vdf_groups_from_ref_hits(hits.ctypes.data, len(hits), C.byref(g))
    if rc:
        raise VdfError(rc, "vdf_groups_from_ref_hits failed")
    try:
        return _groups_to_lists(g)
    finally:
        lib.vdf_groups_free(C.byref(g))


class HashQueue:
    """Thread-safe batching of per-clip hash requests (vdf_hash_queue_*): concurrent submit()
    calls from many threads share one batched GPU launch.  One queue per frame size."""

    def __init__(self, engine: Engine, w: int, h: int, max_batch: int = 256, max_wait_us: int = 2000,
                 letterbox: bool = False):
        self.engine = engine
        self.w, self.h = int(w), int(h)
        q = C.c_void_p()
        engine._check(engine.lib.vdf_hash_queue_create(engine.ctx, self.w, self.h, int(max_batch), int(max_wait_us),
                                                       1 if letterbox else 0, C.byref(q)))
        self.q = q

    def submit(self, frames: np.ndarray):
        """frames [>=16, H, W] u8 -> (hash [16] u64, crop (l, r, t, b)).  Blocks; releases the GIL while waiting."""
        f = np.ascontiguousarray(frames[:16], dtype=np.uint8)
        if f.shape != (16, self.h, self.w):
            raise ValueError("a queue takes 16 frames of its own frame size")
        out = np.zeros(HASH_WORDS, np.uint64)
        crop = np.zeros(4, np.uint32)
        self.engine._check(self.engine.lib.vdf_hash_queue_submit(self.q, f.ctypes.data, out.ctypes.data, crop.ctypes.data))
        return out, tuple(int(x) for x in crop)

    def stats(self):
        nb, nc = C.c_uint64(0), C.c_uint64(0)
        self.engine.lib.vdf_hash_queue_stats(self.q, C.byref(nb), C.byref(nc))
        return int(nb.value), int(nc.value)

    def in_flight_max(self) -> int:
        v = C.c_uint32(0)
        self.engine.lib.vdf_hash_queue_in_flight_max(self.q, C.byref(v))
        return int(v.value)

    def close(self):
        if getattr(self, "q", None):
            self.engine.lib.vdf_hash_queue_destroy(self.q)
            self.q = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
