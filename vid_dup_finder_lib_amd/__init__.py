"""vid_dup_finder_lib_amd: MI355X-native VideoHash construction + Hamming search behind the
vid_dup_finder_lib API (VideoHash / search / search_with_references / MatchGroup).

The compute path is hand-written HIP for gfx950 in csrc/, reached through the C ABI of
include/vdf.h (libvdf_hip.so).  There is no CPU fallback.
"""
from ._capi import (DEFAULT_SEARCH_TOLERANCE, HASH_BITS, HASH_WORDS, TOLERANCE_SCALING_FACTOR, VdfError)
from .api import (Crop, Cropdetect, cropdetect_letterbox, Error, MatchGroup, gen_hashes, NotEnoughFrames, NotVideo, TooFewEntries, VideoHash, VidProc, default_engine,
                  hash_frame_stacks, rust_path_key, search, search_with_references, sort_order)
from .engine import Engine

__all__ = ["Crop", "Cropdetect", "cropdetect_letterbox", "gen_hashes", "VideoHash", "MatchGroup", "Error", "NotEnoughFrames", "NotVideo", "VidProc", "TooFewEntries", "search",
           "search_with_references", "hash_frame_stacks", "Engine", "default_engine", "VdfError", "rust_path_key",
           "sort_order", "DEFAULT_SEARCH_TOLERANCE", "TOLERANCE_SCALING_FACTOR", "HASH_BITS", "HASH_WORDS"]
