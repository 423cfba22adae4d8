"""ctypes front end of the CPU oracle (oracle/vdf_oracle.c) + a numpy/scipy twin.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Parity status (details in vdf_oracle.c): search/Hamming pinned structurally by the
reference's tests; hash bits PARITY UNPINNED (no known-answer vector upstream).

Citations are relative to /root/reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvdf_oracle.so")

DCT_SIZE = 16
HASH_SIZE = 10
HASH_BITS = 1000
HASH_WORDS = 16
E_NOT_ENOUGH_FRAMES = -1
E_BAD_DIMS = -2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "vdf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u64p, u32p, u8p, i64p = (C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8),
                                 C.POINTER(C.c_int64))
        L.vdf_oracle_tolerance_int.restype = C.c_uint32
        L.vdf_oracle_tolerance_int.argtypes = [C.c_double]
        L.vdf_oracle_hamming.restype = C.c_uint32
        L.vdf_oracle_hamming.argtypes = [u64p, u64p]
        L.vdf_oracle_resize_frame_u8.restype = C.c_int
        L.vdf_oracle_resize_frame_u8.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.vdf_oracle_resize_coeffs.restype = C.c_int
        L.vdf_oracle_resize_coeffs.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int16), C.c_int32, C.POINTER(C.c_int32)]
        L.vdf_oracle_dct3d.restype = None
        L.vdf_oracle_dct3d.argtypes = [C.POINTER(C.c_double)]
        L.vdf_oracle_dct16.restype = None
        L.vdf_oracle_dct16.argtypes = [C.POINTER(C.c_double)]
        L.vdf_oracle_hash_frames16.restype = C.c_int
        L.vdf_oracle_hash_frames16.argtypes = [u8p, C.c_uint32, u64p, C.POINTER(C.c_double)]
        L.vdf_oracle_hash_clip.restype = C.c_int
        L.vdf_oracle_hash_clip.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, u64p,
                                           C.POINTER(C.c_double)]
        L.vdf_oracle_hash_clips.restype = C.c_int
        L.vdf_oracle_hash_clips.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, u64p]
        L.vdf_oracle_search_self.restype = C.c_int64
        L.vdf_oracle_search_self.argtypes = [u64p, u32p, C.c_size_t, C.c_uint32, u64p, u64p]
        L.vdf_oracle_search_refs.restype = C.c_int64
        L.vdf_oracle_search_refs.argtypes = [u64p, u32p, C.c_size_t, u64p, u32p, C.c_size_t, C.c_uint32, u64p, u64p,
                                             i64p, u64p]
        L.vdf_oracle_pairs_self.restype = C.c_uint64
        L.vdf_oracle_pairs_self.argtypes = [u32p, C.c_size_t]
        L.vdf_oracle_letterbox_crop.restype = None
        L.vdf_oracle_letterbox_crop.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32, u32p]
        L.vdf_oracle_cropdetect_letterbox.restype = C.c_int
        L.vdf_oracle_cropdetect_letterbox.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, u32p]
        L.vdf_oracle_hash_clip_letterbox.restype = C.c_int
        L.vdf_oracle_hash_clip_letterbox.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, u64p,
                                                     C.POINTER(C.c_double), u32p]
        _lib = L
    return _lib


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _hashes(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, HASH_WORDS)


# ---------------------------------------------------------------- arithmetic
def tolerance_int(tolerance: float) -> int:
    return int(lib().vdf_oracle_tolerance_int(float(tolerance)))


def hamming(x, y) -> int:
    x = np.ascontiguousarray(x, dtype=np.uint64).reshape(HASH_WORDS)
    y = np.ascontiguousarray(y, dtype=np.uint64).reshape(HASH_WORDS)
    return int(lib().vdf_oracle_hamming(_p(x, C.c_uint64), _p(y, C.c_uint64)))


def resize_frame(frame: np.ndarray) -> np.ndarray:
    """One HxW u8 frame -> 16x16 u8 (resize_gray.rs:11-54)."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w = frame.shape
    out = np.empty((DCT_SIZE, DCT_SIZE), dtype=np.uint8)
    rc = lib().vdf_oracle_resize_frame_u8(_p(frame, C.c_uint8), w, h, _p(out, C.c_uint8))
    if rc:
        raise ValueError(f"oracle resize failed: {rc}")
    return out


def resize_coeffs(in_size: int, out_size: int = DCT_SIZE):
    """(precision, window, start[out], size[out], w[out, window] i16)."""
    cap = (int(np.ceil(3.0 * max(in_size / out_size, 1.0))) * 2 + 1) * out_size
    start = np.zeros(out_size, np.int32)
    size = np.zeros(out_size, np.int32)
    w = np.zeros(cap, np.int16)
    window = C.c_int32(0)
    p = lib().vdf_oracle_resize_coeffs(in_size, out_size, _p(start, C.c_int32), _p(size, C.c_int32),
                                       _p(w, C.c_int16), cap, C.byref(window))
    if p < 0:
        raise ValueError("coefficient build failed")
    return p, window.value, start, size, w[: window.value * out_size].reshape(out_size, window.value)


def dct3d(cube: np.ndarray) -> np.ndarray:
    """Unnormalised 3-D DCT-II of a [t][x][y] f64 cube (raw_dct_ops.rs:107-142)."""
    out = np.array(cube, dtype=np.float64, order="C").reshape(DCT_SIZE, DCT_SIZE, DCT_SIZE).copy()
    lib().vdf_oracle_dct3d(_p(out, C.c_double))
    return out


def dct16(line) -> np.ndarray:
    """One unnormalised 16-point DCT-II by the oracle's split-radix butterfly."""
    out = np.ascontiguousarray(line, dtype=np.float64).copy()
    assert out.shape == (DCT_SIZE,)
    lib().vdf_oracle_dct16(_p(out, C.c_double))
    return out


def hash_clip(frames: np.ndarray, want_coefs: bool = False):
    """frames: [n_frames, H, W] u8 -> (rc, hash[16] u64, coefs[1000] f64 | None).
    rc = E_NOT_ENOUGH_FRAMES when n_frames < 16 (video_hash.rs:53,61)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if frames.ndim != 3:
        raise ValueError("frames must be [n_frames, H, W]")
    n, h, w = frames.shape
    out = np.zeros(HASH_WORDS, np.uint64)
    coefs = np.zeros(HASH_BITS, np.float64) if want_coefs else None
    rc = lib().vdf_oracle_hash_clip(_p(frames, C.c_uint8), n, w, h, w * h, _p(out, C.c_uint64),
                                    _p(coefs, C.c_double) if want_coefs else None)
    return rc, out, coefs


def hash_clips(frames: np.ndarray) -> np.ndarray:
    """frames: [n_clips, n_frames, H, W] u8 -> [n_clips, 16] u64 (raises on error)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    nc, nf, h, w = frames.shape
    out = np.zeros((nc, HASH_WORDS), np.uint64)
    rc = lib().vdf_oracle_hash_clips(_p(frames, C.c_uint8), nc, nf, w, h, _p(out, C.c_uint64))
    if rc:
        raise ValueError(f"oracle hash_clips failed: {rc}")
    return out


def hash_clips_with_coefs(frames: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    hs, cs = [], []
    for clip in frames:
        rc, h, c = hash_clip(clip, want_coefs=True)
        if rc:
            raise ValueError(f"oracle hash_clip failed: {rc}")
        hs.append(h)
        cs.append(c)
    return np.stack(hs), np.stack(cs)


# ------------------------------------------------------------ letterbox crop
def letterbox_crop(frame: np.ndarray, tol: int = 16):
    """(left, right, top, bottom) of one HxW u8 frame, LetterboxColour::AnyColour(tol)
    (vid_dup_finder_common/src/video_frames_gray.rs:38-128)."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w = frame.shape
    out = np.zeros(4, np.uint32)
    lib().vdf_oracle_letterbox_crop(_p(frame, C.c_uint8), w, h, w, int(tol), _p(out, C.c_uint32))
    return tuple(int(x) for x in out)


def cropdetect_letterbox(frames: np.ndarray):
    """Clip-level crop (video_frames_gray.rs:201-210): frames 0, 8, .. united.  frames [n, H, W] u8."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, h, w = frames.shape
    out = np.zeros(4, np.uint32)
    rc = lib().vdf_oracle_cropdetect_letterbox(_p(frames, C.c_uint8), n, w, h, w * h, _p(out, C.c_uint32))
    if rc:
        raise ValueError(f"cropdetect failed: {rc}")
    return tuple(int(x) for x in out)


def hash_clip_letterbox(frames: np.ndarray, want_coefs: bool = False):
    """crop_video_frames + from_frames (video_hash_builder.rs:188-204,222) -> (rc, hash, coefs|None, crop)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, h, w = frames.shape
    out = np.zeros(HASH_WORDS, np.uint64)
    crop = np.zeros(4, np.uint32)
    coefs = np.zeros(HASH_BITS, np.float64) if want_coefs else None
    rc = lib().vdf_oracle_hash_clip_letterbox(_p(frames, C.c_uint8), n, w, h, w * h, _p(out, C.c_uint64),
                                              _p(coefs, C.c_double) if want_coefs else None, _p(crop, C.c_uint32))
    return rc, out, coefs, tuple(int(x) for x in crop)


# -------------------------------------------------------------------- search
def search_self_sorted(hashes, dur, tol_int: int) -> List[List[int]]:
    """search_algorithm.rs:81-171 on already-sorted SoA input; returns groups of sorted indices
    (hits ascending, target last; groups in the reference's reversed order)."""
    hashes = _hashes(hashes)
    dur = np.ascontiguousarray(dur, dtype=np.uint32)
    n = len(dur)
    assert hashes.shape[0] == n
    offsets = np.zeros(n // 2 + 2, np.uint64)
    members = np.zeros(max(n, 1), np.uint64)
    ng = lib().vdf_oracle_search_self(_p(hashes, C.c_uint64), _p(dur, C.c_uint32), n, int(tol_int),
                                      _p(offsets, C.c_uint64), _p(members, C.c_uint64))
    return [[int(m) for m in members[int(offsets[g]): int(offsets[g + 1])]] for g in range(ng)]


def search_refs_sorted(cand_hashes, cand_dur, ref_hashes, ref_dur, tol_int: int) -> List[Tuple[int, List[int]]]:
    """video_dup_finder.rs:19-46 on sorted candidates; returns [(ref_input_index, [cand sorted indices])]."""
    ch = _hashes(cand_hashes)
    cd = np.ascontiguousarray(cand_dur, dtype=np.uint32)
    rh = _hashes(ref_hashes)
    rd = np.ascontiguousarray(ref_dur, dtype=np.uint32)
    total = C.c_uint64(0)
    args = (_p(ch, C.c_uint64), _p(cd, C.c_uint32), len(cd), _p(rh, C.c_uint64), _p(rd, C.c_uint32), len(rd),
            int(tol_int))
    ng = lib().vdf_oracle_search_refs(*args, None, None, None, C.byref(total))
    offsets = np.zeros(ng + 1, np.uint64)
    members = np.zeros(max(int(total.value), 1), np.uint64)
    ref_index = np.zeros(max(ng, 1), np.int64)
    lib().vdf_oracle_search_refs(*args, _p(offsets, C.c_uint64), _p(members, C.c_uint64), _p(ref_index, C.c_int64),
                                 C.byref(total))
    return [(int(ref_index[g]), [int(m) for m in members[int(offsets[g]): int(offsets[g + 1])]]) for g in range(ng)]


def pairs_self(dur) -> int:
    dur = np.ascontiguousarray(dur, dtype=np.uint32)
    return int(lib().vdf_oracle_pairs_self(_p(dur, C.c_uint32), len(dur)))


# --- Rust `PathBuf: Ord` (std::path::Path::cmp compares Components, not bytes) -------------------
def rust_path_key(path) -> tuple:
    """Sort key equal to Rust's component-wise Path ordering on Unix
    (search_algorithm.rs:55-61 sorts by (duration, src_path.to_owned()))."""
    b = os.fsencode(path) if not isinstance(path, bytes) else path
    comps = []
    has_root = b.startswith(b"/")
    if has_root:
        comps.append((1, b""))
    elif b == b"." or b.startswith(b"./"):
        comps.append((2, b""))
    for part in b.split(b"/"):
        if part == b"" or part == b".":
            continue  # repeated separators and inner "." are not components
        comps.append((3, b"") if part == b".." else (4, part))
    return tuple(comps)


def sort_order(durations: Sequence[int], paths: Sequence) -> List[int]:
    """Stable permutation produced by Search::sort (search_algorithm.rs:55-61)."""
    idx = list(range(len(durations)))
    idx.sort(key=lambda i: (int(durations[i]), rust_path_key(paths[i])))
    return idx


def search(hashes, durations, paths, tolerance: float) -> List[List]:
    """video_dup_finder.rs:7-13: returns MatchGroups as lists of paths (duplicates order)."""
    order = sort_order(durations, paths)
    h = _hashes(hashes)[order] if len(order) else np.zeros((0, HASH_WORDS), np.uint64)
    d = np.asarray(durations, dtype=np.uint32)[order] if len(order) else np.zeros(0, np.uint32)
    groups = search_self_sorted(h, d, tolerance_int(tolerance))
    return [[paths[order[m]] for m in g] for g in groups if len(g) >= 2]  # MatchGroup::new needs >= 2


def search_with_references(ref_hashes, ref_durations, ref_paths, hashes, durations, paths, tolerance: float):
    """video_dup_finder.rs:19-46: returns [(ref_path, [dup paths])] in reference input order."""
    order = sort_order(durations, paths)
    h = _hashes(hashes)[order] if len(order) else np.zeros((0, HASH_WORDS), np.uint64)
    d = np.asarray(durations, dtype=np.uint32)[order] if len(order) else np.zeros(0, np.uint32)
    res = search_refs_sorted(h, d, ref_hashes, ref_durations, tolerance_int(tolerance))
    return [(ref_paths[r], [paths[order[m]] for m in ms]) for r, ms in res]


# --- numpy / scipy twin (independent restatement used to cross-check the C oracle) ---------------
def np_lanczos3_weights(in_size: int, out_size: int = DCT_SIZE):
    """(window, start[out], size[out], w[out, window] f64): the normalised Lanczos3 weights before quantisation."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    radius = 3.0 * fscale
    window = int(np.ceil(radius)) * 2 + 1
    starts, sizes, rows = [], [], []
    for o in range(out_size):
        in_center = (o + 0.5) * scale
        x_min = int(max(np.floor(in_center - radius), 0.0))
        x_max = int(min(np.ceil(in_center + radius), float(in_size)))
        xs = np.arange(x_min, x_max, dtype=np.float64)
        arg = (xs - (in_center - 0.5)) / fscale
        w = np.where((arg >= -3.0) & (arg < 3.0), np.sinc(arg) * np.sinc(arg / 3.0), 0.0)
        lead = 0
        while lead < len(w) and w[lead] == 0.0:
            lead += 1
        w = w[lead:]
        x_min += lead
        trail = len(w)
        while trail > 0 and w[trail - 1] == 0.0:
            trail -= 1
        s = w.sum()
        if s != 0.0:
            w = w / s
        row = np.zeros(window)
        row[: len(w)] = w
        starts.append(x_min)
        sizes.append(trail)
        rows.append(row)
    return window, np.array(starts), np.array(sizes), np.stack(rows)


def np_lanczos3_coeffs(in_size: int, out_size: int = DCT_SIZE):
    window, starts, sizes, vals = np_lanczos3_weights(in_size, out_size)
    max_w = vals.max()
    precision = 0
    for p in range(16):
        precision = p
        if int(np.floor(max_w * (1 << (p + 1)) + 0.5)) >= (1 << 15):
            break
    q = np.floor(np.abs(vals) * (1 << precision) + 0.5) * np.sign(vals)  # round half away from zero
    return precision, window, starts, sizes, q.astype(np.int64)


def np_resize_frame(frame: np.ndarray) -> np.ndarray:
    frame = np.asarray(frame, dtype=np.int64)
    h, w = frame.shape
    if (h, w) == (DCT_SIZE, DCT_SIZE):
        return frame.astype(np.uint8)

    def conv(img, axis_len):
        p, _, st, sz, q = np_lanczos3_coeffs(axis_len)
        out = np.empty((img.shape[0], DCT_SIZE), np.int64)
        for o in range(DCT_SIZE):
            acc = (1 << (p - 1)) + (img[:, st[o]: st[o] + sz[o]] * q[o, : sz[o]]).sum(axis=1)
            out[:, o] = np.clip(acc >> p, 0, 255)
        return out

    tmp = conv(frame, w) if w != DCT_SIZE else frame
    if h != DCT_SIZE:
        tmp = conv(tmp.T.copy(), h).T
    return tmp.astype(np.uint8)


def _np_twiddle(i: int, fft_len: int):
    """rustdct::twiddles::single_twiddle(i, fft_len).conj() as (re, im)."""
    angle = (np.pi * -2.0 / fft_len) * i
    return float(np.cos(angle)), float(-np.sin(angle))


def np_dct2_splitradix(x: np.ndarray) -> np.ndarray:
    """Unnormalised DCT-II along the LAST axis (length 2, 4, 8 or 16) by rustdct's split-radix recursion
    (Type2And3SplitRadix / the Type2And3Butterfly{4,8,16} steps derived from it): an independent, vectorised twin of
    oracle/vdf_oracle.c dct2_len16.  numpy rounds every product and sum separately, as Rust does."""
    n = x.shape[-1]
    if n == 1:
        return x.copy()
    if n == 2:
        return np.stack([x[..., 0] + x[..., 1], (x[..., 0] - x[..., 1]) * np.sqrt(0.5)], axis=-1)
    half, q = n // 2, n // 4
    in2 = np.empty(x.shape[:-1] + (half,))
    ev = np.empty(x.shape[:-1] + (q,))
    od = np.empty(x.shape[:-1] + (q,))
    for i in range(q):
        bottom, top, hb, ht = x[..., i], x[..., n - i - 1], x[..., half - i - 1], x[..., half + i]
        in2[..., i] = top + bottom
        in2[..., half - i - 1] = hb + ht
        lower, upper = bottom - top, hb - ht
        re, im = _np_twiddle(2 * i + 1, 4 * n)
        ev[..., i] = lower * re + upper * im
        sin_in = upper * re - lower * im
        od[..., q - i - 1] = sin_in if i % 2 == 0 else -sin_in
    o2, oe, oo = np_dct2_splitradix(in2), np_dct2_splitradix(ev), np_dct2_splitradix(od)
    out = np.empty_like(x)
    out[..., 0], out[..., 1], out[..., 2] = o2[..., 0], oe[..., 0], o2[..., 1]
    for i in range(1, q):
        c = oe[..., i]
        sv = -oo[..., q - i] if (i + q) % 2 == 0 else oo[..., q - i]
        out[..., 4 * i - 1] = c + sv
        out[..., 4 * i] = o2[..., 2 * i]
        out[..., 4 * i + 1] = c - sv
        out[..., 4 * i + 2] = o2[..., 2 * i + 1]
    out[..., n - 1] = -oo[..., 0]
    return out


def np_dct3d(cube: np.ndarray) -> np.ndarray:
    """[t][x][y] f64 cube -> 3-D DCT-II, passes along y, x, t in the reference's order (raw_dct_ops.rs:118-132)."""
    d = np_dct2_splitradix(np.ascontiguousarray(cube, dtype=np.float64))                 # y (last axis)
    d = np.swapaxes(np_dct2_splitradix(np.ascontiguousarray(np.swapaxes(d, 1, 2))), 1, 2)  # x
    d = np.swapaxes(np_dct2_splitradix(np.ascontiguousarray(np.swapaxes(d, 0, 2))), 0, 2)  # t
    return d


def np_hash_frames16(frames16: np.ndarray, want_coefs: bool = False):
    """[16,16(y),16(x)] u8 -> hash words via the numpy split-radix twin (bit-identical coefficients to the C oracle)."""
    cube = np.transpose(np.asarray(frames16[:DCT_SIZE], dtype=np.float64), (0, 2, 1)) - 128.0  # [t][x][y]
    d = np_dct3d(cube)
    coefs = d[:HASH_SIZE, :HASH_SIZE, :HASH_SIZE].reshape(-1)
    bits = coefs > 0.0
    words = np.zeros(HASH_WORDS, np.uint64)
    for i in np.nonzero(bits)[0]:
        words[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return (words, coefs) if want_coefs else words
