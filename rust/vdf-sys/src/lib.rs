//! Raw bindings: one declaration per entry point of include/vdf.h that the crate's seams use.  Keep in step with the
//! header (tests/test_capi_symbols.py of the engine checks header <-> exports <-> the Python twin of this file).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct vdf_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct vdf_hash_queue {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct vdf_hit {
    pub row: u32,
    pub col: u32,
}

/// CSR form of Vec<MatchGroup> (matches/match_group.rs:10-13); payload owned by the library, free with vdf_groups_free.
#[repr(C)]
pub struct vdf_groups {
    pub n_groups: u64,
    pub offsets: *mut u64,
    pub members: *mut u64,
    pub ref_index: *mut i64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct vdf_search_stats {
    pub pairs: u64,
    pub pairs_computed: u64,
    pub n_hits: u64,
    pub n_tiles: u64,
    pub n_launches: u32,
    pub kernel_ms: f32,
    pub pairs_early_exit: u64,
    pub early_exit_bits: u32,
    pub reserved: u32,
}

pub const VDF_OK: c_int = 0;
pub const VDF_E_NOT_ENOUGH_FRAMES: c_int = -1;
pub const VDF_E_BAD_DIMS: c_int = -2;
pub const VDF_E_HIP: c_int = -3;
pub const VDF_E_OOM: c_int = -4;
pub const VDF_E_INVAL: c_int = -5;
pub const VDF_E_OVERFLOW: c_int = -6;
pub const VDF_E_RCCL: c_int = -7;

extern "C" {
    pub fn vdf_ctx_create(device_id: c_int, out: *mut *mut vdf_ctx) -> c_int;
    pub fn vdf_ctx_create_multi(device_ids: *const c_int, n_devices: c_int, out: *mut *mut vdf_ctx) -> c_int;
    pub fn vdf_ctx_device_count(ctx: *const vdf_ctx) -> c_int;
    pub fn vdf_ctx_destroy(ctx: *mut vdf_ctx);
    pub fn vdf_last_error(ctx: *const vdf_ctx) -> *const c_char;
    pub fn vdf_ctx_last_search_stats(ctx: *const vdf_ctx, out: *mut vdf_search_stats) -> c_int;
    /// bytes of device memory held by the library's growable buffers over all contexts (diagnostics)
    pub fn vdf_live_device_bytes() -> std::os::raw::c_longlong;

    pub fn vdf_tolerance_int(tolerance: f64) -> u32;
    pub fn vdf_hamming_u1024(a: *const u64, b: *const u64) -> u32;
    pub fn vdf_groups_free(g: *mut vdf_groups);

    pub fn vdf_hash_frames_u8(
        ctx: *mut vdf_ctx, frames: *const u8, n_clips: usize, frames_per_clip: u32, w: u32, h: u32,
        frame_stride: usize, clip_stride: usize, out_hashes: *mut u64, out_dontcare: *mut u32,
    ) -> c_int;
    pub fn vdf_hash_frames_u8_letterbox(
        ctx: *mut vdf_ctx, frames: *const u8, n_clips: usize, frames_per_clip: u32, w: u32, h: u32,
        frame_stride: usize, clip_stride: usize, out_hashes: *mut u64, out_crops: *mut u32, out_dontcare: *mut u32,
    ) -> c_int;

    pub fn vdf_search_self(
        ctx: *mut vdf_ctx, hashes: *const u64, durations: *const u32, n: usize, tol_int: u32, out: *mut vdf_groups,
    ) -> c_int;
    pub fn vdf_search_refs(
        ctx: *mut vdf_ctx, cand_hashes: *const u64, cand_durations: *const u32, n_cand: usize,
        ref_hashes: *const u64, ref_durations: *const u32, n_ref: usize, tol_int: u32, out: *mut vdf_groups,
    ) -> c_int;

    pub fn vdf_hash_queue_create(
        ctx: *mut vdf_ctx, w: u32, h: u32, max_batch: u32, max_wait_us: u32, letterbox: c_int, out: *mut *mut vdf_hash_queue,
    ) -> c_int;
    pub fn vdf_hash_queue_submit(q: *mut vdf_hash_queue, frames: *const u8, out_hash: *mut u64, out_crop: *mut u32) -> c_int;
    pub fn vdf_hash_queue_destroy(q: *mut vdf_hash_queue);

    // Device-resident / sharded building blocks (see include/vdf.h); hipStream_t travels as *mut c_void.
    pub fn vdf_search_self_device(
        ctx: *mut vdf_ctx, d_hashes: *const u64, d_durations: *const u32, n: usize, tol_int: u32, shard_index: u32,
        shard_count: u32, row_begin: u32, row_end: u32, d_matched: *const u32, hits: *mut vdf_hit, capacity: u64,
        n_hits: *mut u64, overflow_row: *mut u32, stream: *mut c_void,
    ) -> c_int;
    pub fn vdf_replay_self(
        n: usize, hits: *const vdf_hit, n_hits: u64, row_begin: u32, row_end: u32, matched: *mut u8, out: *mut vdf_groups,
    ) -> c_int;
    pub fn vdf_groups_finish_self(g: *mut vdf_groups) -> c_int;
    // Search::sort on the device for a database that stays in HBM between hashing and searching
    // (search_algorithm.rs:55-61; d_path_rank = each entry's rank among the caller's paths in PathBuf order, or null).
    pub fn vdf_sort_order_device(
        ctx: *mut vdf_ctx, d_durations: *const u32, d_path_rank: *const u32, n: usize, d_perm_out: *mut u32, stream: *mut c_void,
    ) -> c_int;
    pub fn vdf_apply_order_device(
        ctx: *mut vdf_ctx, d_hashes: *const u64, d_durations: *const u32, d_perm: *const u32, n: usize, d_hashes_out: *mut u64,
        d_durations_out: *mut u32, stream: *mut c_void,
    ) -> c_int;
    pub fn vdf_sort_hits(hits: *mut vdf_hit, n_hits: u64) -> c_int;
    // A promise that the resident database (d_hashes, n) does not change until the next call: reference searches against it reuse
    // its operand expansion.  Null withdraws the promise.
    pub fn vdf_ctx_pin_database(ctx: *mut vdf_ctx, d_hashes: *const u64, n: usize) -> c_int;
}
