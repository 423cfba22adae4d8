"""ctypes binding of include/vdf.h (libvdf_hip.so).

This is the Python twin of the `vdf-sys` binding shown in INTEGRATION.md: every symbol
declared in include/vdf.h is bound here with its exact C signature.  There is no fallback:
if the shared library is missing the import of the compute API fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdf_hip.so")

VDF_OK = 0
VDF_E_NOT_ENOUGH_FRAMES = -1
VDF_E_BAD_DIMS = -2
VDF_E_HIP = -3
VDF_E_OOM = -4
VDF_E_INVAL = -5
VDF_E_OVERFLOW = -6
VDF_E_RCCL = -7

DCT_SIZE = 16
HASH_SIZE = 10
HASH_BITS = 1000
HASH_WORDS = 16
DEFAULT_SEARCH_TOLERANCE = 0.35  # vid_dup_finder_lib/src/definitions.rs:5
TOLERANCE_SCALING_FACTOR = 1000.0  # definitions.rs:40


class VdfHit(C.Structure):
    _fields_ = [("row", C.c_uint32), ("col", C.c_uint32)]


class VdfGroups(C.Structure):
    _fields_ = [
        ("n_groups", C.c_uint64),
        ("offsets", C.POINTER(C.c_uint64)),
        ("members", C.POINTER(C.c_uint64)),
        ("ref_index", C.POINTER(C.c_int64)),
    ]


class VdfSearchStats(C.Structure):
    _fields_ = [
        ("pairs", C.c_uint64),
        ("pairs_computed", C.c_uint64),
        ("n_hits", C.c_uint64),
        ("n_tiles", C.c_uint64),
        ("n_launches", C.c_uint32),
        ("kernel_ms", C.c_float),
        ("pairs_early_exit", C.c_uint64),
        ("early_exit_bits", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class VdfSearchTiming(C.Structure):
    _fields_ = [
        ("prep_ms", C.c_float), ("stream_ms", C.c_float), ("resolve_ms", C.c_float), ("download_ms", C.c_float),
        ("replay_ms", C.c_float), ("total_ms", C.c_float), ("suspects", C.c_uint64), ("suspect_capacity", C.c_uint64), ("hits_filtered", C.c_uint64),
    ]


class VdfCacheSoa(C.Structure):
    _fields_ = [
        ("n_entries", C.c_uint64), ("n_ok", C.c_uint64), ("n_err", C.c_uint64), ("n_key_differs", C.c_uint64),
        ("hashes", C.POINTER(C.c_uint64)), ("durations", C.POINTER(C.c_uint32)),
        ("path_offsets", C.POINTER(C.c_uint64)), ("paths", C.POINTER(C.c_char)),
        ("mtime_secs", C.POINTER(C.c_uint64)), ("mtime_nanos", C.POINTER(C.c_uint32)),
    ]


class VdfCacheMetadata(C.Structure):
    _fields_ = [("operating_system", C.c_int32), ("decode_backend", C.c_int32), ("crop", C.c_int32), ("reserved", C.c_int32),
                ("skip_forward_amount", C.c_double), ("cache_version", C.c_uint64)]


class VdfCacheSearchTiming(C.Structure):
    _fields_ = [("rank_ms", C.c_float), ("upload_ms", C.c_float), ("sort_ms", C.c_float), ("search_ms", C.c_float),
                ("map_ms", C.c_float), ("total_ms", C.c_float)]


AGREE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64))
OR_BITMAP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class VdfShardExchange(C.Structure):
    _fields_ = [("user", C.c_void_p), ("agree", AGREE_FN), ("or_bitmap", OR_BITMAP_FN)]


VDF_CACHE_OS_WINDOWS, VDF_CACHE_OS_UNIX = 0, 1
VDF_CACHE_BACKEND_FFMPEG, VDF_CACHE_BACKEND_GSTREAMER = 0, 1
VDF_CROPDETECT_NONE, VDF_CROPDETECT_LETTERBOX, VDF_CROPDETECT_MOTION = 0, 1, 2

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_ctx = C.c_void_p

# name -> (restype, argtypes).  Keep in step with include/vdf.h; tests/test_capi_symbols.py checks both ways.
SIGNATURES = {
    "vdf_ctx_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "vdf_ctx_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_ctx)]),
    "vdf_ctx_device_count": (C.c_int, [_ctx]),
    "vdf_ctx_device_at": (C.c_int, [_ctx, C.c_int]),
    "vdf_ctx_device_search_stats": (C.c_int, [_ctx, C.c_int, C.POINTER(VdfSearchStats)]),
    "vdf_ctx_device_search_timing": (C.c_int, [_ctx, C.c_int, C.POINTER(VdfSearchTiming)]),
    "vdf_ctx_rccl_ranks": (C.c_int, [_ctx]),
    "vdf_ctx_destroy": (None, [_ctx]),
    "vdf_last_error": (C.c_char_p, [_ctx]),
    "vdf_version": (C.c_char_p, []),
    "vdf_ctx_device": (C.c_int, [_ctx]),
    "vdf_ctx_set_hit_capacity": (C.c_int, [_ctx, C.c_uint64]),
    "vdf_ctx_last_search_stats": (C.c_int, [_ctx, C.POINTER(VdfSearchStats)]),
    "vdf_ctx_last_search_timing": (C.c_int, [_ctx, C.POINTER(VdfSearchTiming)]),
    "vdf_live_device_bytes": (C.c_longlong, []),
    "vdf_live_pinned_bytes": (C.c_longlong, []),
    "vdf_hamming_u1024": (C.c_uint32, [_u64p, _u64p]),
    "vdf_tolerance_int": (C.c_uint32, [C.c_double]),
    "vdf_count_pairs_self": (C.c_uint64, [_u32p, C.c_size_t]),
    "vdf_count_pairs_refs": (C.c_uint64, [_u32p, C.c_size_t, _u32p, C.c_size_t]),
    "vdf_groups_free": (None, [C.POINTER(VdfGroups)]),
    "vdf_hash_frames_u8": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t,
                                     C.c_size_t, C.c_void_p, C.c_void_p]),
    "vdf_hash_frames_u8_device": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdf_cropdetect_letterbox_device": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                  C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "vdf_hash_frames_u8_cropped_device": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                    C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]),
    "vdf_hash_frames_u8_letterbox_device": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                      C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p]),
    "vdf_hash_frames_u8_letterbox_device_async": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                            C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                            C.c_void_p]),
    "vdf_hash_frames_u8_letterbox": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdf_sort_order_paths": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]),
    "vdf_search_self": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(VdfGroups)]),
    "vdf_search_refs": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_uint32, C.POINTER(VdfGroups)]),
    "vdf_search_self_device": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p]),
    "vdf_search_self_device_replay": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64,
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(VdfShardExchange), C.c_void_p]),
    "vdf_bitmap_or_device": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "vdf_search_refs_device": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                         C.c_void_p]),
    "vdf_search_self_shards": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.c_uint32, C.POINTER(VdfGroups)]),
    "vdf_search_refs_shards": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32,
                                         C.POINTER(VdfGroups)]),
    "vdf_hash_frames_u8_shards": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]),
    "vdf_sort_order_device": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "vdf_apply_order_device": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdf_ctx_pin_database": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "vdf_row_tile_size": (C.c_uint32, []),
    "vdf_replay_self": (C.c_int, [C.c_size_t, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                  C.POINTER(VdfGroups)]),
    "vdf_groups_finish_self": (C.c_int, [C.POINTER(VdfGroups)]),
    "vdf_sort_hits": (C.c_int, [C.c_void_p, C.c_uint64]),
    "vdf_groups_from_ref_hits": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(VdfGroups)]),
    "vdf_hash_queue_create": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                        C.POINTER(C.c_void_p)]),
    "vdf_hash_queue_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vdf_hash_queue_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vdf_hash_queue_in_flight_max": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "vdf_hash_queue_destroy": (None, [C.c_void_p]),
    "vdf_groups_max_distance": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(VdfGroups),
                                          C.c_void_p]),
    "vdf_cache_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(VdfCacheSoa)]),
    "vdf_cache_free": (None, [C.POINTER(VdfCacheSoa)]),
    "vdf_cache_encode": (C.c_int, [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "vdf_buffer_free": (None, [C.c_void_p]),
    "vdf_cache_decode_fallbacks": (C.c_ulonglong, []),
    "vdf_cache_decode_mt": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(VdfCacheSoa)]),
    "vdf_cache_metadata_new": (C.c_int, [C.c_int32, C.c_double, C.POINTER(VdfCacheMetadata)]),
    "vdf_cache_metadata_format": (C.c_int, [C.POINTER(VdfCacheMetadata), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vdf_cache_metadata_parse": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(VdfCacheMetadata), C.c_void_p, C.c_size_t]),
    "vdf_cache_metadata_validate": (C.c_int, [C.POINTER(VdfCacheMetadata), C.c_int32, C.c_double, C.c_void_p, C.c_size_t]),
    "vdf_cache_metadata_path": (C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "vdf_path_compare": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "vdf_path_ranks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "vdf_search_cache_entries": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(VdfGroups), C.POINTER(VdfCacheSearchTiming)]),
}

_lib = None


def load() -> C.CDLL:
    """Load libvdf_hip.so and bind every entry point; raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C vid_dup_finder_lib_amd/csrc`). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  If libvdf_hip.so pulled in
        # /opt/rocm's copy first, a later `import torch` would load a SECOND runtime, which then finds no GPU.
        # Importing torch first makes the loader satisfy our NEEDED libamdhip64.so.7 with the copy already mapped.
        # (C/C++/Rust hosts have no torch in the process and link /opt/rocm's runtime as usual.)
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class VdfError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"vdf error {code}: {message}")
        self.code = code
