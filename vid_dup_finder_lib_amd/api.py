"""Python mirror of the reference crate's public surface for the hot path
(vid_dup_finder_lib/src/lib.rs:132-140): VideoHash, MatchGroup, Error, search,
search_with_references.  Same names, argument meaning and error behaviour; the compute goes
through the C ABI to the GPU.  The Rust-side shim a maintainer would add is in INTEGRATION.md.

Out of scope here (callers of the path, decode-bound): VideoHashBuilder / CreationOptions
(video_hash_builder.rs) -- their output contract (16 equal-size gray frames + duration in
seconds) is exactly the input of VideoHash.from_frames below.
"""
from __future__ import annotations

import enum
import itertools
import os
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np

from . import _capi
from ._capi import DEFAULT_SEARCH_TOLERANCE, HASH_BITS, HASH_WORDS, TOLERANCE_SCALING_FACTOR, VdfError
from .engine import Engine, hamming_distance_words, tolerance_int

__all__ = ["Crop", "Cropdetect", "cropdetect_letterbox", "gen_hashes", "VideoHash", "MatchGroup", "Error", "NotEnoughFrames", "NotVideo", "VidProc", "TooFewEntries", "search",
           "search_with_references", "default_engine", "hash_frame_stacks", "rust_path_key", "sort_order",
           "DEFAULT_SEARCH_TOLERANCE", "TOLERANCE_SCALING_FACTOR"]


# ---- Error (vid_dup_finder_lib/src/video_hashing/mod.rs:17-28) --------------------------------------
class Error(Exception):
    """An error that prevented a video hash from being created."""


class NotVideo(Error):
    def __init__(self):
        super().__init__("File is not a video")


class VidProc(Error):
    def __init__(self, msg: str):
        super().__init__(f"Video processing error: {msg}")


class NotEnoughFrames(Error):
    def __init__(self):
        super().__init__("Could not extract enough frames")


class Cropdetect(enum.Enum):
    """vid_dup_finder_lib/src/definitions.rs:47-54.  Motion-based detection (vid_dup_finder_common/src/motioncrop/)
    is not on the accelerated path."""
    NONE = "none"
    LETTERBOX = "letterbox"
    MOTION = "motion"


class Crop:
    """The crop box of a frame as edge offsets (vid_dup_finder_common/src/crop.rs:3-10): what the letterbox detection yields and
    what crop_resize_buf takes.  On the C ABI a box is the four u32 {left, right, top, bottom} (`out_crops`); this is its host-side
    type with the reference's constructors, checks and accessors.  Ordered like the derive: orig_res, left, right, top, bottom."""

    __slots__ = ("orig_res", "left", "right", "top", "bottom")
    _U32 = 0xFFFFFFFF

    def __init__(self, orig_res, left: int, right: int, top: int, bottom: int):
        self.orig_res = (int(orig_res[0]), int(orig_res[1]))
        self.left, self.right, self.top, self.bottom = int(left), int(right), int(top), int(bottom)

    @classmethod
    def from_edge_offsets(cls, orig_res, left: int, right: int, top: int, bottom: int) -> "Crop":
        """crop.rs:13-30: a box that leaves no pixel is a panic there, an AssertionError here."""
        assert left + right < orig_res[0], "crop box leaves no columns"
        assert top + bottom < orig_res[1], "crop box leaves no rows"
        return cls(orig_res, left, right, top, bottom)

    @classmethod
    def from_topleft_and_dims(cls, orig_res, x: int, y: int, width: int, height: int) -> "Crop":
        """crop.rs:32-50 (u32 arithmetic: a box that sticks out of the frame underflows there; refused here)."""
        right, bottom = orig_res[0] - width - x, orig_res[1] - height - y
        if right < 0 or bottom < 0:
            raise OverflowError("box outside the frame")
        return cls(orig_res, x, right, y, bottom)

    @classmethod
    def from_abi(cls, orig_res, box) -> "Crop":
        """One row {left, right, top, bottom} of the C ABI's `out_crops`."""
        return cls.from_edge_offsets(orig_res, int(box[0]), int(box[1]), int(box[2]), int(box[3]))

    @classmethod
    def default(cls) -> "Crop":
        """crop.rs:183-194: an 'enormous' crop to start a fold of unions with."""
        return cls((cls._U32, cls._U32), cls._U32 // 8, cls._U32 // 8, cls._U32 // 8, cls._U32 // 8)

    def union(self, other: "Crop") -> "Crop":
        """crop.rs:53-68: per-edge minimum (what unites the crops of the probed frames; the resolutions are not compared)."""
        return Crop.from_edge_offsets(self.orig_res, min(self.left, other.left), min(self.right, other.right),
                                      min(self.top, other.top), min(self.bottom, other.bottom))

    def as_view_args(self):
        """crop.rs:92-103: (x, y, width, height) of the box - the arguments of crop_resize_buf's .crop()."""
        w, h = self.orig_res[0] - (self.left + self.right), self.orig_res[1] - (self.top + self.bottom)
        if w < 0 or h < 0:
            raise OverflowError("crop offsets exceed the frame")  # checked_sub(..).unwrap()
        return (self.left, self.top, w, h)

    def as_abi(self) -> np.ndarray:
        return np.array([self.left, self.right, self.top, self.bottom], np.uint32)

    def width(self) -> int:
        return self.orig_res[0] - (self.left + self.right)

    def height(self) -> int:
        return self.orig_res[1] - (self.top + self.bottom)

    def area(self) -> int:
        return self.width() * self.height()

    def aspect_ratio(self) -> float:
        return float(self.width()) / float(self.height())

    def enumerate_coords(self) -> Iterator:
        """crop.rs:121-135: the box's (x, y), x outermost."""
        for x in range(self.left, self.orig_res[0] - self.right):
            for y in range(self.top, self.orig_res[1] - self.bottom):
                yield (x, y)

    def enumerate_coords_excluded(self) -> Iterator:
        """crop.rs:137-161: the (x, y) outside the box, the eight surrounding regions clockwise from the top left."""
        xs = (0, self.left, self.orig_res[0] - self.right, self.orig_res[0])
        ys = (0, self.top, self.orig_res[1] - self.bottom, self.orig_res[1])
        for xi, yi in ((0, 0), (1, 0), (2, 0), (2, 1), (0, 2), (1, 2), (2, 2), (0, 1)):
            for x in range(xs[xi], xs[xi + 1]):
                for y in range(ys[yi], ys[yi + 1]):
                    yield (x, y)

    def eroded(self) -> Optional["Crop"]:
        """crop.rs:163-176: one pixel more off every edge, None when nothing would be left."""
        if self.left + self.right + 2 >= self.orig_res[0] or self.top + self.bottom + 2 >= self.orig_res[1]:
            return None
        return Crop(self.orig_res, self.left + 1, self.right + 1, self.top + 1, self.bottom + 1)

    def is_uncropped(self) -> bool:
        return self.left == 0 and self.right == 0 and self.top == 0 and self.bottom == 0

    def _key(self):
        return (self.orig_res, self.left, self.right, self.top, self.bottom)

    def __eq__(self, other):
        return isinstance(other, Crop) and self._key() == other._key()

    def __lt__(self, other):
        return self._key() < other._key()

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return f"Crop(orig_res={self.orig_res}, left={self.left}, right={self.right}, top={self.top}, bottom={self.bottom})"


class TooFewEntries(Exception):
    """match_group.rs:15-16"""


_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    """Process-wide engine on GPU LOCAL_RANK (or 0).  Fails loudly without a GPU or the built library."""
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine


# ---- Rust `PathBuf: Ord` --------------------------------------------------------------------------
def rust_path_key(path) -> tuple:
    """Key reproducing std::path::Path's component-wise ordering on Unix (Search::sort uses
    (duration, src_path.to_owned()) as its key, search_algorithm.rs:55-61).  Components:
    RootDir < CurDir < ParentDir < Normal(bytes); repeated '/' and inner '.' are not components."""
    b = path if isinstance(path, bytes) else os.fsencode(path)
    comps = []
    if b.startswith(b"/"):
        comps.append((1, b""))
    elif b == b"." or b.startswith(b"./"):
        comps.append((2, b""))
    for part in b.split(b"/"):
        if part in (b"", b"."):
            continue
        comps.append((3, b"") if part == b".." else (4, part))
    return tuple(comps)


def sort_order(hashes: Sequence["VideoHash"], engine: Optional[Engine] = None) -> List[int]:
    """Stable permutation that Search::sort applies (search_algorithm.rs:55-61).  With an engine and more than a handful of hashes the
    order comes from the library (vdf_sort_order_paths: the path half on the device for plain paths, the native component comparator
    otherwise - the same order): a million Python keys cost seconds beside a 0.1 s search."""
    if engine is not None and len(hashes) >= 2048:
        from .cache import sort_order_paths

        order, _used_device = sort_order_paths(engine, [h.duration() for h in hashes], [h.src_path() for h in hashes])
        return order.tolist()
    idx = list(range(len(hashes)))
    idx.sort(key=lambda i: (hashes[i].duration(), rust_path_key(hashes[i].src_path())))
    return idx


# ---- VideoHash (video_hash.rs:26-32) ----------------------------------------------------------------
class VideoHash:
    """hash: 16 x u64 (1024 bits, Lsb0; bits 1000..1023 are zero when built from frames),
    src_path, duration in seconds."""

    __slots__ = ("hash", "_src_path", "_duration")

    def __init__(self, hash_words=None, src_path="", duration: int = 0):
        if hash_words is None:
            hash_words = np.zeros(HASH_WORDS, np.uint64)  # Default, video_hash.rs:34-42
        self.hash = np.ascontiguousarray(hash_words, dtype=np.uint64).reshape(HASH_WORDS).copy()
        self._src_path = src_path
        self._duration = int(duration)

    @classmethod
    def from_frames(cls, frames: Iterable[np.ndarray], src_path, duration: int,
                    engine: Optional[Engine] = None) -> "VideoHash":
        """video_hash.rs:45-73.  frames: iterable of equal-size 2-D u8 arrays (gray, row-major).
        Empty or fewer than 16 -> NotEnoughFrames; only the first 16 are used (dct_3d.rs:25)."""
        taken = list(itertools.islice(iter(frames), _capi.DCT_SIZE))
        if len(taken) < _capi.DCT_SIZE:
            raise NotEnoughFrames()
        first = np.asarray(taken[0])
        for f in taken:
            if np.asarray(f).shape != first.shape or np.asarray(f).ndim != 2:
                raise VidProc("frames must be equal-size 2-D gray images")
        stack = np.stack([np.ascontiguousarray(f, dtype=np.uint8) for f in taken])[None]
        try:
            words = (engine or default_engine()).hash_frames(stack)[0]
        except VdfError as e:
            if e.code == _capi.VDF_E_NOT_ENOUGH_FRAMES:
                raise NotEnoughFrames() from e
            if e.code == _capi.VDF_E_BAD_DIMS:
                raise VidProc(str(e)) from e
            raise
        return cls(words, src_path, duration)

    def src_path(self):
        return self._src_path

    def duration(self) -> int:
        return self._duration

    def hamming_distance(self, other: "VideoHash") -> int:
        """video_hash.rs:190-192: raw distance over all 16 words."""
        return hamming_distance_words(self.hash, other.hash)

    def normalized_hamming_distance(self, other: "VideoHash") -> float:
        """video_hash.rs:200-204 (feature app_only_fns)."""
        return float(self.hamming_distance(other)) / TOLERANCE_SCALING_FACTOR

    def hash_bits(self) -> np.ndarray:
        """The 1000 hash bits as a bool array (video_hash.rs:226-228)."""
        bits = np.unpackbits(self.hash.view(np.uint8), bitorder="little")
        return bits[:HASH_BITS].astype(bool)

    # test_util-style constructors (video_hash.rs:240-308) that need no RNG
    def with_duration(self, duration: int) -> "VideoHash":
        return VideoHash(self.hash, self._src_path, duration)

    def with_src_path(self, src_path) -> "VideoHash":
        return VideoHash(self.hash, src_path, self._duration)

    @classmethod
    def full_hash(cls, name) -> "VideoHash":
        return cls(np.full(HASH_WORDS, np.uint64(0xFFFFFFFFFFFFFFFF)), name, 0)

    @classmethod
    def empty_hash(cls, name) -> "VideoHash":
        return cls(np.zeros(HASH_WORDS, np.uint64), name, 0)

    # test_util constructors that draw random bits (video_hash.rs:272-306); rng = numpy Generator in place of StdRng
    @classmethod
    def random_hash(cls, rng: np.random.Generator) -> "VideoHash":
        """1000 fair bits, padding bits 1000..1023 zero, empty path, duration 0 (video_hash.rs:293-306)."""
        bits = rng.integers(0, 2, size=HASH_WORDS * 64, dtype=np.uint8)
        bits[HASH_BITS:] = 0
        return cls(np.packbits(bits, bitorder="little").view(np.uint64).copy(), "", 0)

    def hash_with_spatial_distance(self, target_distance: int, rng: np.random.Generator) -> "VideoHash":
        """A hash at exactly `target_distance` bits from this one; any of the 1024 positions may flip, padding included
        (video_hash.rs:272-291).  Upstream random-walks single flips until the distance is first reached, which never
        terminates in practice beyond the 512-bit equilibrium; by symmetry the first-hit point is uniform on the sphere of
        that radius, which `target_distance` distinct random positions sample directly."""
        if not 0 <= target_distance <= HASH_WORDS * 64:
            raise ValueError("target_distance must be within 0..1024")
        bits = np.unpackbits(self.hash.view(np.uint8), bitorder="little")
        bits[rng.choice(HASH_WORDS * 64, size=target_distance, replace=False)] ^= 1
        out = VideoHash(np.packbits(bits, bitorder="little").view(np.uint64).copy(), self._src_path, self._duration)
        assert self.hamming_distance(out) == target_distance
        return out

    def _key(self):
        return (tuple(int(x) for x in self.hash), rust_path_key(self._src_path), self._duration)

    def __eq__(self, other):
        return isinstance(other, VideoHash) and self._key() == other._key()

    def __lt__(self, other):  # derive(Ord): field order hash, src_path, duration
        return self._key() < other._key()

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return f"VideoHash(src_path={self._src_path!r}, duration={self._duration}, hash=0x{int(self.hash[0]):016x}..)"


# ---- MatchGroup (matches/match_group.rs) --------------------------------------------------------------
class MatchGroup:
    __slots__ = ("_reference", "_duplicates")

    def __init__(self, reference, duplicates: List):
        self._reference = reference
        self._duplicates = list(duplicates)

    @classmethod
    def new(cls, entries: Iterable) -> "MatchGroup":
        dups = list(entries)
        if len(dups) < 2:  # match_group.rs:25
            raise TooFewEntries()
        return cls(None, dups)

    @classmethod
    def new_with_reference(cls, reference, entries: Iterable) -> "MatchGroup":
        dups = list(entries)
        if not dups:  # match_group.rs:41
            raise TooFewEntries()
        return cls(reference, dups)

    def len(self) -> int:
        return len(self._duplicates)

    __len__ = len

    def reference(self):
        return self._reference

    def duplicates(self) -> Iterator:
        return iter(self._duplicates)

    def contained_paths(self) -> Iterator:
        """duplicates, then the reference if there is one (match_group.rs:68-81)."""
        yield from self._duplicates
        if self._reference is not None:
            yield self._reference

    def dup_combinations(self) -> List["MatchGroup"]:
        """match_group.rs:87-105."""
        if self._reference is not None:
            return [MatchGroup.new_with_reference(self._reference, [d]) for d in self._duplicates]
        return [MatchGroup.new([a, b]) for a, b in itertools.combinations(self._duplicates, 2)]

    def __eq__(self, other):
        return (isinstance(other, MatchGroup) and self._reference == other._reference
                and self._duplicates == other._duplicates)

    def __repr__(self):
        return f"MatchGroup(reference={self._reference!r}, duplicates={self._duplicates!r})"


# ---- search / search_with_references (video_dup_finder.rs) -------------------------------------------
def _soa(hashes: Sequence[VideoHash], order: Sequence[int]):
    if not order:
        return np.zeros((0, HASH_WORDS), np.uint64), np.zeros(0, np.uint32)
    # (one bytes object of all words: half the time of np.stack over a million 16-word arrays)
    words = np.frombuffer(b"".join([hashes[i].hash.tobytes() for i in order]), np.uint64).reshape(-1, HASH_WORDS)
    dur = np.array([hashes[i].duration() for i in order], dtype=np.uint32)
    return words, dur


def search(hashes: Iterable[VideoHash], tolerance: float, engine: Optional[Engine] = None) -> List[MatchGroup]:
    """video_dup_finder.rs:7-13.  Sort (host, needs paths), GPU all-pairs search inside the one-sided
    x1.1 duration window, host replay of the greedy consumption; groups come back in the reference's order."""
    hashes = list(hashes)
    if not hashes:
        return []  # search_algorithm.rs:89-91
    engine = engine or default_engine()
    order = sort_order(hashes, engine)
    words, dur = _soa(hashes, order)
    groups = engine.search_self_sorted(words, dur, tolerance_int(tolerance))
    out = []
    for g in groups:
        try:
            out.append(MatchGroup.new(hashes[order[m]].src_path() for m in g))
        except TooFewEntries:  # filter_map(.. .ok()), video_dup_finder.rs:11
            pass
    return out


def search_with_references(ref_hashes: Iterable[VideoHash], new_hashes: Iterable[VideoHash], tolerance: float,
                           engine: Optional[Engine] = None) -> List[MatchGroup]:
    """video_dup_finder.rs:19-46: one group per reference with >= 1 match, in reference input order."""
    refs = list(ref_hashes)
    news = list(new_hashes)
    if not refs or not news:
        return []
    engine = engine or default_engine()
    order = sort_order(news, engine)
    words, dur = _soa(news, order)
    rwords, rdur = _soa(refs, list(range(len(refs))))
    res = engine.search_refs_sorted(words, dur, rwords, rdur, tolerance_int(tolerance))
    return [MatchGroup.new_with_reference(refs[r].src_path(), [news[order[m]].src_path() for m in ms])
            for r, ms in res]


def hash_frame_stacks(frames: np.ndarray, src_paths: Sequence, durations: Sequence[int],
                      engine: Optional[Engine] = None) -> List[VideoHash]:
    """Batched VideoHash.from_frames: frames [n_clips, n_frames >= 16, H, W] u8 -> one VideoHash per clip
    (the batching a caller of VideoHashBuilder::hash would do to feed the GPU)."""
    try:
        words = (engine or default_engine()).hash_frames(frames)
    except VdfError as e:
        if e.code == _capi.VDF_E_NOT_ENOUGH_FRAMES:
            raise NotEnoughFrames() from e
        raise
    return [VideoHash(words[i], src_paths[i], durations[i]) for i in range(len(words))]


def gen_hashes(frames: np.ndarray, src_paths: Sequence, durations: Sequence[int],
               cropdetect: Cropdetect = Cropdetect.LETTERBOX, engine: Optional[Engine] = None) -> List[VideoHash]:
    """The part of `gen_hash` after decode (video_hash_builder.rs:214-223) for a batch of clips:
    crop_video_frames(cropdetect) -- default Letterbox, like CreationOptions::default (:55-63) -- then
    VideoHash::from_frames.  frames [n_clips, n_frames >= 16, H, W] u8.  Detection (frames 0 and 8,
    video_frames_gray.rs:201-210) and the cropped resize both run on the GPU; no cropped copies are made."""
    if cropdetect == Cropdetect.NONE:
        return hash_frame_stacks(frames, src_paths, durations, engine)
    if cropdetect != Cropdetect.LETTERBOX:
        raise VidProc("Cropdetect::Motion is not supported by the accelerated path")
    try:
        words, _crops = (engine or default_engine()).hash_frames_letterbox(frames)
    except VdfError as e:
        if e.code == _capi.VDF_E_NOT_ENOUGH_FRAMES:
            raise NotEnoughFrames() from e
        if e.code == _capi.VDF_E_BAD_DIMS:
            raise VidProc(str(e)) from e
        raise
    return [VideoHash(words[i], src_paths[i], durations[i]) for i in range(len(words))]


def cropdetect_letterbox(frames: np.ndarray, engine: Optional[Engine] = None) -> List[Crop]:
    """cropdetect_letterbox (vid_dup_finder_common/src/video_frames_gray.rs:201-210) for a batch of clips on the GPU: frames
    [n_clips, n_frames >= 16, H, W] u8 -> the union (crop.rs:53-68) of the letterbox crops of frames 0 and 8 of each clip."""
    try:
        _words, crops = (engine or default_engine()).hash_frames_letterbox(frames)
    except VdfError as e:
        if e.code == _capi.VDF_E_NOT_ENOUGH_FRAMES:
            raise NotEnoughFrames() from e
        if e.code == _capi.VDF_E_BAD_DIMS:
            raise VidProc(str(e)) from e
        raise
    h, w = int(frames.shape[2]), int(frames.shape[3])
    return [Crop.from_abi((w, h), c) for c in np.asarray(crops).reshape(-1, 4)]
