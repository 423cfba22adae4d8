/*
 * vdf.h -- C ABI of the MI355X-native VideoHash construction + Hamming search engine.
 *
 * This is the drop-in boundary for the hot path of Farmadupe/vid_dup_finder_lib.  The
 * reference has no FFI layer of its own (it is one Rust crate); the entry points below are
 * what a `vdf-sys` binding would declare to replace the loops named next to each function.
 * The Rust-side stub is shown in INTEGRATION.md.  Citations are relative to the reference
 * repository root.
 *
 * Conventions
 *   - Plain pointers and sizes only.  No Rust, C++ or torch type crosses this boundary.
 *   - Return value: VDF_OK (0) or a negative vdf_status.  A human-readable message for the
 *     last failure on a context is available from vdf_last_error().
 *   - A hash is 16 x uint64_t = 1024 bits: bit i of the hash is bit (i & 63) of word (i >> 6)
 *     (bitvec Lsb0 over [usize;16], vid_dup_finder_lib/src/video_hashing/video_hash.rs:29,64-68).
 *   - "sorted order" means the order Search::sort leaves the entries in
 *     (src/video_hashing/search_algorithm.rs:55-61: stable by (duration, src_path)).  Sorting
 *     needs paths and stays on the caller's side; every index this library returns is an index
 *     into the arrays the caller passed.
 *   - Functions with the suffix _device take DEVICE pointers (HIP allocations on the
 *     context's GPU) and a hipStream_t passed as void* (NULL = the context's own non-blocking
 *     stream, NOT the legacy default stream) and need a single-device context; everything else
 *     takes host pointers (or, *_shards, one device pointer per GPU of a multi-GPU context).
 *     Stream order is all the ordering there is: work the caller queued on ANOTHER stream - a fill of the output buffer, the
 *     decoder's writes of the frames - is not waited for.  Pass the stream that work is on, or finish it first.
 *   - The search calls need their (candidate) arrays in sorted order and return VDF_E_INVAL when the
 *     durations are not ascending.
 *   - vdf_ctx_create() binds a context to one GPU.  vdf_ctx_create_multi() makes ONE context over several GPUs of
 *     the node (one host thread and one stream per device inside the library): the host-array calls
 *     (vdf_search_self, vdf_search_refs, vdf_hash_frames_u8[_letterbox]) then fan out by themselves, and the
 *     *_shards calls take data that is already resident on the devices.  See DESIGN.md "Multi-GPU".
 *   - Thread safety: a context serialises its own calls with an internal mutex, so
 *     vdf_hash_frames_u8 may be called from many threads (the app hashes from rayon workers,
 *     vid_dup_finder_app/src/video_hash_filesystem_cache/video_hash_filesystem_cache.rs:246).
 */
#ifndef VDF_H
#define VDF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDF_DCT_SIZE 16    /* src/definitions.rs:34 */
#define VDF_HASH_SIZE 10   /* src/definitions.rs:36 */
#define VDF_HASH_BITS 1000 /* src/definitions.rs:42 */
#define VDF_HASH_WORDS 16  /* src/definitions.rs:43 */
#define VDF_DEFAULT_SEARCH_TOLERANCE 0.35 /* src/definitions.rs:5 */

typedef enum vdf_status {
    VDF_OK = 0,
    VDF_E_NOT_ENOUGH_FRAMES = -1, /* Error::NotEnoughFrames, src/video_hashing/mod.rs:27 */
    VDF_E_BAD_DIMS = -2,          /* the size asserts of dct_3d.rs:31-38 / Error::VidProc */
    VDF_E_HIP = -3,               /* HIP runtime failure or no usable GPU */
    VDF_E_OOM = -4,
    VDF_E_INVAL = -5,
    VDF_E_OVERFLOW = -6,          /* a caller-provided hit buffer was too small */
    VDF_E_RCCL = -7               /* librccl could not be loaded, or a collective failed (multi-GPU contexts only) */
} vdf_status;

typedef struct vdf_ctx vdf_ctx;

/* One thresholded pair.  Self-search: row = target index i, col = candidate index j > i.
 * Reference search: row = reference index (caller's order), col = candidate index. */
typedef struct vdf_hit {
    uint32_t row;
    uint32_t col;
} vdf_hit;

/* CSR result, the C form of Vec<MatchGroup> (src/video_hashing/matches/match_group.rs:10-13).
 * members[offsets[g] .. offsets[g+1]) are the duplicates of group g, already in the
 * reference's order.  ref_index[g] is the reference's position for search_with_references
 * and -1 for search().  Payload is library-allocated; release with vdf_groups_free(). */
typedef struct vdf_groups {
    uint64_t n_groups;
    uint64_t *offsets;  /* n_groups + 1 */
    uint64_t *members;  /* offsets[n_groups] */
    int64_t *ref_index; /* n_groups */
} vdf_groups;

/* Statistics of the last search on a context (for benchmarks and profiles). */
typedef struct vdf_search_stats {
    uint64_t pairs;         /* comparisons the reference's duration windows admit */
    uint64_t pairs_computed;/* pairs the tiles actually evaluated (>= pairs) */
    uint64_t n_hits;        /* thresholded pairs produced by the device */
    uint64_t n_tiles;       /* workgroups launched for the distance kernel */
    uint32_t n_launches;    /* distance-kernel launches (1 unless the hit buffer overflowed) */
    float kernel_ms;        /* HIP-event time of the distance kernel(s), on their stream */
    uint64_t pairs_early_exit; /* of pairs_computed: comparisons that stopped before the last bit because the partial distance
                                  of their whole block already exceeded the tolerance (exact; both backends) */
    uint32_t early_exit_bits;  /* bit positions counted before that test (0 = test disabled) */
    uint32_t reserved;
} vdf_search_stats;

/* Where the time of the last search call on a context went (benchmarks; accumulated over the launches of the call).
 * Device figures are HIP-event times on the kernels' own stream, host figures wall time. */
typedef struct vdf_search_timing {
    float prep_ms;      /* host wall: operand expansion + window / tile kernels, up to the launch of the distance kernel */
    float stream_ms;    /* device: the distance kernel proper */
    float resolve_ms;   /* device: exact evaluation of the queued suspect pairs (matrix-core backend; else 0) */
    float download_ms;  /* host wall: hit list into (row, col) order and into the caller's buffer */
    float replay_ms;    /* host wall: greedy replay / group assembly (host-level calls only) */
    float total_ms;     /* host wall of the whole call (host-level calls only) */
    uint64_t suspects;          /* suspect-queue entries written (matrix-core backend) */
    uint64_t suspect_capacity;  /* size of that queue in the last launch */
    uint64_t hits_filtered;     /* thresholded pairs dropped on the device because their row can never become a target of the
                                   greedy replay (host-level search() and the *_replay device call, sharded or not; they are counted in
                                   vdf_search_stats.n_hits) */
} vdf_search_timing;

/* ---- context ------------------------------------------------------------------------------ */
int vdf_ctx_create(int device_id, vdf_ctx **out);
/* One context over n_devices GPUs of this node: the single search() / search_with_references() call of the crate
 * (src/video_hashing/video_dup_finder.rs:7-13,19-46, called once per run from vid_dup_finder_app/src/app/app_fns.rs:
 * 478-482) then uses all of them.  Inside: one host thread + stream per device; the sorted database is replicated on every
 * device (from the caller's host arrays directly, or - *_shards calls - by an RCCL all-gather over xGMI), row tiles of the
 * triangle are dealt round-robin, the host merges the hits and replays the greedy grouping once, so the MatchGroups are
 * identical for every device count.  A device may be listed more than once (testing on one GPU): its slots then share
 * the GPU and the collective is replaced by device-to-device copies. */
int vdf_ctx_create_multi(const int *device_ids, int n_devices, vdf_ctx **out);
int vdf_ctx_device_count(const vdf_ctx *ctx);        /* 1 for vdf_ctx_create */
int vdf_ctx_device_at(const vdf_ctx *ctx, int slot); /* HIP device id of a slot, -1 if out of range */
/* Per-device statistics of the last search on a multi-GPU context (vdf_ctx_last_search_stats gives the sums, with
 * kernel_ms = the slowest device's). */
int vdf_ctx_device_search_stats(const vdf_ctx *ctx, int slot, vdf_search_stats *out);
int vdf_ctx_device_search_timing(const vdf_ctx *ctx, int slot, vdf_search_timing *out); /* that slot's phases and hits_filtered */
/* RCCL communicators (= ranks, one per GPU) the context has initialised so far: 0 until a *_shards call replicated through librccl
 * (a device list that repeats a GPU, or a single-device context, never does: plain device copies). */
int vdf_ctx_rccl_ranks(const vdf_ctx *ctx);
void vdf_ctx_destroy(vdf_ctx *ctx);
const char *vdf_last_error(const vdf_ctx *ctx); /* ctx may be NULL: last ctx_create failure */
const char *vdf_version(void);
int vdf_ctx_device(const vdf_ctx *ctx);
/* Hit-buffer capacity (entries) used by the host-level search calls; default 1<<24: 128 MB of DEVICE memory per GPU of the context
 * (8 bytes per entry).  The page-locked host staging the lists come down through is sized by what the searches actually produce
 * (512 KB to begin with), not by this capacity. */
int vdf_ctx_set_hit_capacity(vdf_ctx *ctx, uint64_t capacity);
int vdf_ctx_last_search_stats(const vdf_ctx *ctx, vdf_search_stats *out);
int vdf_ctx_last_search_timing(const vdf_ctx *ctx, vdf_search_timing *out); /* multi-GPU context: host figures + the slowest device's */
/* Diagnostics: bytes of device memory held by the library's growable buffers over all contexts of the process (tables, staging, hit
 * lists, crop descriptors ...).  Back to its earlier value after vdf_ctx_destroy, or a buffer was forgotten. */
long long vdf_live_device_bytes(void);
long long vdf_live_pinned_bytes(void); /* the same for page-locked host staging */

/* ---- host helpers (no GPU needed) ------------------------------------------------------------ */
/* VideoHash::hamming_distance, video_hash.rs:190-192,311-317: all 16 words, padding included. */
uint32_t vdf_hamming_u1024(const uint64_t *a, const uint64_t *b);
/* `(tolerance * TOLERANCE_SCALING_FACTOR) as u32`, search_algorithm.rs:64,82. */
uint32_t vdf_tolerance_int(double tolerance);
/* Sum over targets of the candidates inside the one-sided x1.1 window (search_algorithm.rs:99). */
uint64_t vdf_count_pairs_self(const uint32_t *sorted_durations, size_t n);
/* Sum over references of the +-5% window sizes (search_algorithm.rs:173-185). */
uint64_t vdf_count_pairs_refs(const uint32_t *sorted_cand_durations, size_t n_cand, const uint32_t *ref_durations,
                              size_t n_ref);
void vdf_groups_free(vdf_groups *g);

/* ---- hash construction: replaces VideoHash::from_frames, video_hash.rs:45-73 ----------------
 * (crop_resize_buf per frame, vid_dup_finder_common/src/resize_gray.rs:11-54; Dct3d::from_images
 * + dct_3d, dct_3d.rs:15-53 and raw_dct_ops.rs:107-142; hash_bits, dct_3d.rs:55-66; Lsb0 pack).
 * frames: n_clips clips of frames_per_clip gray u8 frames of w x h (row-major, tightly packed rows);
 * frame f of clip c starts at frames + c*clip_stride + f*frame_stride (bytes).  Only the first 16
 * frames of a clip are read.  frames_per_clip < 16 (or 0) -> VDF_E_NOT_ENOUGH_FRAMES.
 * out_hashes: n_clips x 16 words.  out_dontcare (nullable): per clip, the number of the 1000
 * coefficients with |coef| < 1e-6 (their sign is rounding noise; DESIGN.md "Parity rule"). */
int vdf_hash_frames_u8(vdf_ctx *ctx, const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w,
                       uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *out_hashes,
                       uint32_t *out_dontcare);
int vdf_hash_frames_u8_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                              uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *d_out_hashes,
                              uint32_t *d_out_dontcare, void *stream);

/* ---- letterbox crop detection + cropped hashing (the step right before from_frames) -------------
 * Replaces crop_video_frames with Cropdetect::Letterbox, the builder's default
 * (src/video_hashing/video_hash_builder.rs:188-212,59): cropdetect_letterbox
 * (vid_dup_finder_common/src/video_frames_gray.rs:201-210: frames 0 and 8, letterbox_crop with
 * AnyColour(16), :38-128, crops united by per-edge minimum, crop.rs:53-68), then every frame is
 * cropped and handed to from_frames.  Here the crop box is read in place: no cropped copies.
 * Crops are 4 x uint32 per clip: left, right, top, bottom edge offsets (crop.rs:3-10). */
int vdf_cropdetect_letterbox_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                    uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                    uint32_t *d_crops /* DEVICE [n_clips][4] */, void *stream);
/* crops: HOST [n_clips][4] (NULL or all zero = no crop).  l + r >= w or t + b >= h -> VDF_E_INVAL. */
int vdf_hash_frames_u8_cropped_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                      uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                      const uint32_t *crops, uint64_t *d_out_hashes, uint32_t *d_out_dontcare,
                                      void *stream);
/* detect + crop + hash in one call; out_crops (HOST, nullable) receives the detected boxes: they are there when the call returns (the
 * call then ends with a wait for its own work; pass NULL, or use the _async form below, to only queue). */
int vdf_hash_frames_u8_letterbox_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                        uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                        uint64_t *d_out_hashes, uint32_t *d_out_dontcare, uint32_t *out_crops,
                                        void *stream);
/* The same with the boxes left on the DEVICE (d_out_crops: [n_clips][4], nullable), ordered on `stream` like the hashes.  For frames of
 * at most 256 columns and 128 rows the call only queues work: no copy to the host and no wait anywhere between detect and hash (frames of at
 * most 64 x 64: one kernel does both) - the reference's builder likewise detects, crops and hashes a clip in one pass
 * (video_hash_builder.rs:188-204).  Larger frames: the kernel per box shape is chosen on the host, so the call waits for the detect. */
int vdf_hash_frames_u8_letterbox_device_async(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                              uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                              uint64_t *d_out_hashes, uint32_t *d_out_dontcare, uint32_t *d_out_crops,
                                              void *stream);
int vdf_hash_frames_u8_letterbox(vdf_ctx *ctx, const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip,
                                 uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *out_hashes,
                                 uint32_t *out_crops, uint32_t *out_dontcare);

/* ---- search(): replaces Search::search_self, search_algorithm.rs:81-171 (hot loop :150-156) --
 * hashes: n x 16 words, durations: n, both in sorted order.  tol_int from vdf_tolerance_int().
 * Groups come back exactly as search() builds them (video_dup_finder.rs:7-13): members = hits in
 * sorted order then the target; groups in descending target order; every group has >= 2 members. */
int vdf_search_self(vdf_ctx *ctx, const uint64_t *hashes, const uint32_t *durations, size_t n, uint32_t tol_int,
                    vdf_groups *out);

/* ---- search_with_references(): replaces Search::search_one + duration_slice,
 * search_algorithm.rs:63-77,173-185, driven as video_dup_finder.rs:19-46 does (one reference at a
 * time, consume = false).  Candidates in sorted order; references in the caller's order.
 * Groups: one per reference with >= 1 hit, in reference input order; members ascending. */
int vdf_search_refs(vdf_ctx *ctx, const uint64_t *cand_hashes, const uint32_t *cand_durations, size_t n_cand,
                    const uint64_t *ref_hashes, const uint32_t *ref_durations, size_t n_ref, uint32_t tol_int,
                    vdf_groups *out);

/* ---- device-resident pieces (what the host-level calls above are built from) -----------------
 * These let a caller keep the hash database in HBM, shard the work over several processes
 * (one GPU each) and merge on the host.  All produce the thresholded adjacency; the greedy,
 * order-dependent part of search_self is replayed on the host by vdf_replay_self().
 *
 * vdf_search_self_device: emits every pair (i, j), i < j < rhs(i), hamming <= tol_int, where
 *   rhs(i) = first index with duration > (f64(duration[i]) * 1.1) as u32   (search_algorithm.rs:99).
 *   Row tiles (vdf_row_tile_size() consecutive targets) are dealt round-robin: this call handles
 *   tiles t with t % shard_count == shard_index.  Only rows in [row_begin, row_end) are searched.
 *   d_matched (nullable): bitmap, 1 bit per entry, bit set = entry already consumed; such
 *   entries are skipped both as targets and as candidates (exactly the `matched` flag).
 *   hits (HOST buffer, capacity entries) receives the pairs sorted by (row, col); *n_hits is the
 *   number produced by the device (may exceed capacity); *overflow_row is the smallest row that
 *   lost a hit (or, on the matrix-core backend, a suspect pair that could not be queued for the exact
 *   second pass: only possible with very dense near-duplicates), or UINT32_MAX.  Rows below
 *   *overflow_row are complete.
 *   Returns VDF_OK also when a buffer overflowed: check *overflow_row. */
int vdf_search_self_device(vdf_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_durations, size_t n,
                           uint32_t tol_int, uint32_t shard_index, uint32_t shard_count, uint32_t row_begin,
                           uint32_t row_end, const uint32_t *d_matched, vdf_hit *hits, uint64_t capacity,
                           uint64_t *n_hits, uint32_t *overflow_row, void *stream);

/* The replay-only form of vdf_search_self_device for a caller that shards search() over PROCESSES (one GPU each) and feeds the
 * merged lists to vdf_replay_self and nothing else: hits of rows that can never become targets of the greedy loop
 * (search_algorithm.rs:131-170: a row with an incoming hit from a root - a row nobody hits - is consumed before its turn) are
 * dropped on the device before the sort and the download; a cluster of s mutual duplicates sends down s - 1 pairs, not
 * s (s - 1) / 2.  Whether a row is such a row is a property of the COMPLETE hit set, which a sharded launch has spread over
 * its shards, so the shards meet three times through the caller's callbacks (every shard of the launch must make the call,
 * also one whose tiles produce no hits; xchg = NULL with shard_count = 1 is the unsharded case, and with shard_count > 1
 * disables the filter):
 *   agree:     in/out *all_complete (this shard saw no buffer overflow -> AND over the shards), *total_hits (-> sum);
 *              the filter runs only if all are complete (and the total is worth it) - the same decision on every shard;
 *   or_bitmap: d_bitmap[0 .. n_words) (DEVICE, 1 bit per entry) |= every other shard's, in place; the library's kernels that
 *              wrote it were queued on `stream` and the ones that read it will be - order the exchange on it or finish it
 *              before returning (all-gather + vdf_bitmap_or_device, or an all-reduce(MAX) of a byte map: RCCL has no OR).
 *              Called twice per filtered launch (has-incoming, covered).
 * Callbacks return 0 or a negative vdf_status, which the call hands back.  *n_hits = the pairs kept; everything else as
 * vdf_search_self_device (after an overflow nothing is dropped and the overflow protocol applies unchanged). */
typedef struct vdf_shard_exchange {
    void *user;
    int (*agree)(void *user, int *all_complete, uint64_t *total_hits);
    int (*or_bitmap)(void *user, uint32_t *d_bitmap, size_t n_words, void *stream);
} vdf_shard_exchange;
int vdf_search_self_device_replay(vdf_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_durations, size_t n,
                                  uint32_t tol_int, uint32_t shard_index, uint32_t shard_count, uint32_t row_begin,
                                  uint32_t row_end, const uint32_t *d_matched, vdf_hit *hits, uint64_t capacity,
                                  uint64_t *n_hits, uint32_t *overflow_row, const vdf_shard_exchange *xchg, void *stream);
/* d_dst[w] |= d_srcs[0 * n_words + w] | ... | d_srcs[(n_srcs - 1) * n_words + w] on the device (the local half of an OR over
 * shards after an all-gather; with n_srcs = 1 and a zeroed destination: a copy).  Single-device contexts.  Takes no lock:
 * callable from inside an or_bitmap callback. */
int vdf_bitmap_or_device(vdf_ctx *ctx, uint32_t *d_dst, const uint32_t *d_srcs, size_t n_words, uint32_t n_srcs, void *stream);

/* vdf_search_refs_device: emits every pair (r, j) with j inside reference r's +-5% window
 * (search_algorithm.rs:173-185) and hamming <= tol_int; row = r + ref_index_base, hits (HOST
 * buffer) sorted by (row, col).  Every hit is part of the output here (consume = false), so an
 * undersized buffer is simply an error: VDF_E_OVERFLOW with *n_hits = the size required. */
int vdf_search_refs_device(vdf_ctx *ctx, const uint64_t *d_cand_hashes, const uint32_t *d_cand_durations,
                           size_t n_cand, const uint64_t *d_ref_hashes, const uint32_t *d_ref_durations,
                           size_t n_ref, uint32_t tol_int, uint32_t ref_index_base, vdf_hit *hits,
                           uint64_t capacity, uint64_t *n_hits, void *stream);

/* Search::sort on the device (search_algorithm.rs:55-61: stable sort_by_key on (duration, src_path)) for a database that is
 * already resident in HBM - hashes just produced on the GPU need not visit the host between hashing and searching.
 * d_perm_out[k] = input index of the entry at sorted position k.  Paths stay with the caller: d_path_rank (DEVICE, nullable)
 * gives each entry the rank of its src_path among the caller's paths in PathBuf's (component-wise) order, equal paths sharing
 * a rank; NULL = all paths equal.  Entries with equal (duration, rank) keep their input order, as the stable sort does. */
int vdf_sort_order_device(vdf_ctx *ctx, const uint32_t *d_durations, const uint32_t *d_path_rank, size_t n,
                          uint32_t *d_perm_out, void *stream);
/* d_hashes_out[k] = d_hashes[d_perm[k]] (and the durations likewise; both duration pointers may be NULL).  Not in place. */
int vdf_apply_order_device(vdf_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_durations, const uint32_t *d_perm, size_t n,
                           uint64_t *d_hashes_out, uint32_t *d_durations_out, void *stream);

/* A promise that lets searches against ONE resident database skip work: until the next call of this function the n x 16 words at
 * d_hashes will not change.  Searches whose candidate database is exactly (d_hashes, n) then reuse the operand expansion the
 * matrix-core backend makes of it (0.15 ms per million hashes - a sixth of a reference search at the BASELINE configs[4] shape:
 * the app searches its one cache database with reference set after reference set, app_fns.rs:428-482).  d_hashes = NULL withdraws the
 * promise.  Single-device contexts. */
int vdf_ctx_pin_database(vdf_ctx *ctx, const uint64_t *d_hashes, size_t n);

uint32_t vdf_row_tile_size(void); /* rows per tile of the default backend (informational: any shard_count works) */

/* Host replay of search_self's consumption order (search_algorithm.rs:131-170) over hits sorted
 * by (row, col).  matched (nullable, n bytes, in/out) carries consumption state between partial
 * replays; rows in [row_begin, row_end) are replayed.  Appends groups in ASCENDING target order
 * to *out (which must be zero-initialised before the first call); call vdf_groups_finish_self()
 * once at the end to apply ret.reverse() (search_algorithm.rs:167). */
int vdf_replay_self(size_t n, const vdf_hit *hits, uint64_t n_hits, uint32_t row_begin, uint32_t row_end,
                    uint8_t *matched, vdf_groups *out);
int vdf_groups_finish_self(vdf_groups *g);
/* Hits into (row, col) order, in place (host; what the replay and the grouping expect: merging the lists of several
 * processes, one GPU each, is concatenate + this). */
int vdf_sort_hits(vdf_hit *hits, uint64_t n_hits);
/* Groups for search_with_references from hits sorted by (row, col). */
int vdf_groups_from_ref_hits(const vdf_hit *hits, uint64_t n_hits, vdf_groups *out);

/* ---- multi-GPU contexts: data already resident on the devices -------------------------------------------------
 * Array arguments have one entry per slot of the context (vdf_ctx_device_count()); pointer k is a DEVICE pointer on
 * the GPU of slot k.  The caller's work that produced the buffers must be complete (the library runs on its own
 * streams).  Results are identical to the single-device calls on the concatenated arrays.
 *
 * vdf_search_self_shards: the sorted database (Search::sort order, search_algorithm.rs:55-61) cut into consecutive
 *   shards, shard k on device k (sizes may differ, may be 0).  One all-gather of the hashes ((n / G) x 16 x u64 per
 *   device) and one of the durations replicate it - ncclAllGather over xGMI when the shards are equal, the same
 *   exchange as grouped ncclBroadcasts otherwise - then as vdf_search_self.  RCCL failures -> VDF_E_RCCL.
 * vdf_search_refs_shards: candidates sharded and gathered the same way; the references are the concatenation of the
 *   per-device reference shards (group ref_index = position in that concatenation); device k searches its own.
 * vdf_hash_frames_u8_shards: every device hashes its own clips (independent: no communication);
 *   d_out_dontcare may be NULL (or hold NULLs). */
int vdf_search_self_shards(vdf_ctx *ctx, const uint64_t *const *d_hash_shards, const uint32_t *const *d_dur_shards,
                           const size_t *shard_n, uint32_t tol_int, vdf_groups *out);
int vdf_search_refs_shards(vdf_ctx *ctx, const uint64_t *const *d_cand_hash_shards, const uint32_t *const *d_cand_dur_shards,
                           const size_t *cand_shard_n, const uint64_t *const *d_ref_hash_shards,
                           const uint32_t *const *d_ref_dur_shards, const size_t *ref_shard_n, uint32_t tol_int,
                           vdf_groups *out);
int vdf_hash_frames_u8_shards(vdf_ctx *ctx, const uint8_t *const *d_frames, const size_t *n_clips, uint32_t frames_per_clip,
                              uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *const *d_out_hashes,
                              uint32_t *const *d_out_dontcare);

/* ---- batching queue for concurrent per-file callers ------------------------------------------------
 * The app hashes one file per rayon worker (vid_dup_finder_app/src/video_hash_filesystem_cache/
 * video_hash_filesystem_cache.rs:237-257 -> VideoHashBuilder::hash, video_hash_builder.rs:80-82,214-223).
 * vdf_hash_queue_submit may be called from any number of threads: the clips of concurrent callers are
 * hashed by batched launches (the first caller of a batch waits up to max_wait_us for others or until
 * max_batch clips joined).  The queue keeps two batch slots per GPU of the context, each with pinned staging and a
 * private context: while one batch is on the GPU the next one is already collecting (and may launch).
 * frames = 16 gray frames of w x h, tightly packed; blocks until the hash is
 * there.  letterbox != 0 applies Cropdetect::Letterbox first (out_crop, nullable, gets the box). */
typedef struct vdf_hash_queue vdf_hash_queue;
int vdf_hash_queue_create(vdf_ctx *ctx, uint32_t w, uint32_t h, uint32_t max_batch, uint32_t max_wait_us, int letterbox,
                          vdf_hash_queue **out);
int vdf_hash_queue_submit(vdf_hash_queue *q, const uint8_t *frames, uint64_t *out_hash, uint32_t *out_crop);
int vdf_hash_queue_stats(vdf_hash_queue *q, uint64_t *n_batches, uint64_t *n_clips);
/* The largest number of batches that were on the GPU(s) at the same time since the queue was created (>= 2 shows that a
 * second batch was collected and launched while the first was still running). */
int vdf_hash_queue_in_flight_max(vdf_hash_queue *q, uint32_t *out);
void vdf_hash_queue_destroy(vdf_hash_queue *q);

/* ---- the app's Sorting::Distance key (vid_dup_finder_app/src/app/search_output.rs:43-60) ----------
 * out_max[g] = max hamming distance over all pairs of group g's contained paths: its members (indices
 * into hashes, n x 16, host) and, when ref_hashes and groups->ref_index are given and ref_index[g] >= 0,
 * the group's reference ref_hashes[ref_index[g]].  Every group has >= 2 contained paths in the reference. */
int vdf_groups_max_distance(vdf_ctx *ctx, const uint64_t *hashes, size_t n, const uint64_t *ref_hashes, size_t n_ref,
                            const vdf_groups *groups, uint32_t *out_max);

/* ---- the app's on-disk hash cache <-> SoA (host only) ---------------------------------------------
 * Format: bincode 2 `config::standard()` (little endian, varint) of
 * HashMap<PathBuf, MtimeCacheEntry { cache_mtime: SystemTime, value: Result<VideoHash, Error> }>
 * (vid_dup_finder_app/src/video_hash_filesystem_cache/generic_filesystem_cache/base_fs_cache.rs:26,
 * 106-118,192-204; processing_fs_cache.rs:23-27; generic_cache_if.rs:23; video_hash.rs:26-32).
 * Decoding yields the arrays the search ABI takes; entries whose value is Err(..) are counted, not
 * returned.  Entry order is the file's (a HashMap's: arbitrary); sort before searching. */
typedef struct vdf_cache_soa {
    uint64_t n_entries;      /* map length */
    uint64_t n_ok;           /* entries with Ok(VideoHash): length of the arrays below */
    uint64_t n_err;          /* entries with Err(..) */
    uint64_t n_key_differs;  /* Ok entries whose map key != VideoHash.src_path (0 for caches the app wrote) */
    uint64_t *hashes;        /* n_ok x 16 */
    uint32_t *durations;     /* n_ok */
    uint64_t *path_offsets;  /* n_ok + 1 byte offsets into paths (VideoHash.src_path, UTF-8, not NUL terminated) */
    char *paths;
    uint64_t *mtime_secs;    /* n_ok: cache_mtime.secs_since_epoch */
    uint32_t *mtime_nanos;   /* n_ok */
} vdf_cache_soa;
int vdf_cache_decode(const uint8_t *data, size_t len, vdf_cache_soa *out); /* malformed input -> VDF_E_INVAL */
/* The same on n_threads host threads (0 = one per 8 MB of file, at most 32 and at most the machine's; what vdf_cache_decode does).
 * bincode has no entry index, so the file is cut where the byte pattern of an Ok entry's hash words resynchronises, every range
 * is parsed by its own thread, and the ranges must meet exactly: if they do not, the file is decoded front to back instead - the
 * result never depends on the speculation.  Measured (profiles/r05_cache_ingest.txt, the GPU box: 256 hardware threads under a 16-CPU cgroup quota): 10 M entries
 * (2.69 GB) in 35 ms with the automatic thread count, 0.42 s on one thread.  The output arrays ask for transparent huge pages. */
int vdf_cache_decode_mt(const uint8_t *data, size_t len, int n_threads, vdf_cache_soa *out);
unsigned long long vdf_cache_decode_fallbacks(void); /* diagnostics: how often this process fell back to the front-to-back decode */
void vdf_cache_free(vdf_cache_soa *c);
/* Writes a cache the app can load: every entry Ok(VideoHash), key = src_path.  mtime arrays may be NULL (0).
 * *out_data is library-allocated; release with vdf_buffer_free(). */
int vdf_cache_encode(uint64_t n, const uint64_t *hashes, const uint32_t *durations, const uint64_t *path_offsets,
                     const char *paths, const uint64_t *mtime_secs, const uint32_t *mtime_nanos, uint8_t **out_data,
                     size_t *out_len);
void vdf_buffer_free(void *p);

/* ---- the cache's metadata sidecar (host only) ---------------------------------------------------------
 * Next to its cache file <dir>/<stem>.<ext> the app keeps <dir>/<stem>.metadata.txt holding
 * "{operating_system:?},{decode_backend:?},{crop:?},{skip_forward_amount},{cache_version}"
 * (vid_dup_finder_app/src/video_hash_filesystem_cache/cache_metadata.rs:45-51,80-89; video_hash_filesystem_cache.rs:76-139).
 * A cache file WITHOUT its sidecar makes the app exit (video_hash_filesystem_cache.rs:113-117), and a sidecar whose fields differ
 * from the run's options refuses the cache (:127-137) - the crop field is what keeps Cropdetect::None hashes and Letterbox hashes
 * of the same files apart.  Writers of a cache for the app: vdf_cache_encode + vdf_cache_metadata_new / _format / _path.
 * Readers of an app-written cache: vdf_cache_metadata_parse + _validate before the entries are mixed with this engine's hashes. */
enum { VDF_CACHE_OS_WINDOWS = 0, VDF_CACHE_OS_UNIX = 1 };               /* cache_metadata.rs:6-10 */
enum { VDF_CACHE_BACKEND_FFMPEG = 0, VDF_CACHE_BACKEND_GSTREAMER = 1 }; /* cache_metadata.rs:25-29 */
enum { VDF_CROPDETECT_NONE = 0, VDF_CROPDETECT_LETTERBOX = 1, VDF_CROPDETECT_MOTION = 2 }; /* vid_dup_finder_lib/src/definitions.rs:46-54 */
typedef struct vdf_cache_metadata {
    int32_t operating_system;
    int32_t decode_backend;
    int32_t crop;
    int32_t reserved;
    double skip_forward_amount;
    uint64_t cache_version;
} vdf_cache_metadata;
/* VdfCacheMetadata::new (cache_metadata.rs:54-78) as this library's target sees it: Unix, FfmpegBackend (the app's default
 * features), cache_version 1. */
int vdf_cache_metadata_new(int32_t crop, double skip_forward_amount, vdf_cache_metadata *out);
/* to_disk_fmt (:80-89).  *out_len = the text's length (no NUL counted; one is appended when it fits); cap too small -> VDF_E_OVERFLOW. */
int vdf_cache_metadata_format(const vdf_cache_metadata *m, char *buf, size_t cap, size_t *out_len);
/* try_parse (:91-125): exactly five comma-separated fields; operating system and backend are trimmed and lower-cased, crop is the
 * exact variant name, the numbers are Rust's str::parse (no white space).  VDF_E_INVAL with the app's message in err (nullable). */
int vdf_cache_metadata_parse(const char *text, size_t len, vdf_cache_metadata *out, char *err, size_t err_cap);
/* validate (:127-168) against new(exp_crop, exp_skip_forward_amount): VDF_OK, or VDF_E_INVAL with the first mismatch in err. */
int vdf_cache_metadata_validate(const vdf_cache_metadata *act, int32_t exp_crop, double exp_skip_forward_amount, char *err, size_t err_cap);
/* The sidecar's path for a cache path: file_stem() + ".metadata.txt" in the same directory (video_hash_filesystem_cache.rs:93-104).
 * A path without a file name ("..", "/") -> VDF_E_INVAL (the app reports EINVAL there). */
int vdf_cache_metadata_path(const char *cache_path, size_t len, char *buf, size_t cap, size_t *out_len);

/* ---- Search::sort's path order for a whole cache (host only) ------------------------------------------
 * Search::sort keys on (duration, src_path) and PathBuf orders by COMPONENTS (search_algorithm.rs:55-61; std::path::Path::cmp):
 * [RootDir | CurDir], then one Normal component per non-empty piece between '/', inner "." skipped, ".." = ParentDir;
 * RootDir < CurDir < ParentDir < Normal(bytes), a component-wise prefix sorts first.  "a/b" < "a.b", "a//b" == "a/b".
 * vdf_path_compare: -1 / 0 / +1.  vdf_path_ranks: out_rank[i] = number of distinct paths that sort before path i (equal paths
 * share a rank) for the n paths paths[path_offsets[i] .. path_offsets[i + 1]) - the layout of vdf_cache_soa - on n_threads host
 * threads (0 = all); what vdf_sort_order_device takes as d_path_rank.  No per-entry allocation. */
int vdf_path_compare(const char *a, size_t len_a, const char *b, size_t len_b);
int vdf_path_ranks(const char *paths, const uint64_t *path_offsets, size_t n, uint32_t *out_rank, int n_threads);

/* Search::sort's order itself (search_algorithm.rs:55-61) for n entries given as HOST arrays: out_order[k] = index of the entry at
 * position k of the stable order by (duration, PathBuf order of the path).  For plain paths (no empty, "." or ".." component, no NUL:
 * what a directory walk produces) of at most 1024 bytes the path half runs on the device - an LSD radix sort over 8-byte words of the
 * paths - otherwise through vdf_path_ranks on the host; *used_device (nullable) says which.  Same order either way. */
int vdf_sort_order_paths(vdf_ctx *ctx, const uint32_t *durations, const uint64_t *path_offsets, const char *paths, size_t n,
                         uint32_t *out_order, int *used_device);

/* ---- from the entries of a decoded cache to MatchGroups in one call -----------------------------------
 * What the app does between loading its cache and printing groups (vid_dup_finder_app/src/app/app_fns.rs:428-482: fetch the
 * hashes of the --files paths and of the --with-refs paths, then search() or search_with_references()), on the SoA arrays of
 * vdf_cache_decode, with no per-entry host object: upload -> Search::sort on the device (paths included when they are plain:
 * vdf_sort_order_paths; else PathBuf ranks from vdf_path_ranks + vdf_sort_order_device) -> gather -> search.
 * cand_idx (nullable = all n entries, n_cand ignored) and ref_idx select entries of the arrays; n_ref == 0 -> search(), else
 * search_with_references() with the references in ref_idx order.  Group members (and ref_index) are indices into the caller's
 * arrays (NOT into cand_idx / ref_idx), in the reference's order.  timing (nullable) receives the phases. */
typedef struct vdf_cache_search_timing {
    float rank_ms;    /* durations, path blob and offsets to the device + the plain-path check (host route: + vdf_path_ranks) */
    float upload_ms;  /* host wall: hashes, durations, ranks to the device(s) */
    float sort_ms;    /* device + host wall: Search::sort order, gather, order download */
    float search_ms;  /* the search call proper (vdf_ctx_last_search_timing has its phases) */
    float map_ms;     /* host: group members back to the caller's indices */
    float total_ms;
} vdf_cache_search_timing;
int vdf_search_cache_entries(vdf_ctx *ctx, const uint64_t *hashes, const uint32_t *durations, const uint64_t *path_offsets,
                             const char *paths, size_t n, const uint64_t *cand_idx, size_t n_cand, const uint64_t *ref_idx,
                             size_t n_ref, uint32_t tol_int, vdf_groups *out, vdf_cache_search_timing *timing);

#ifdef __cplusplus
}
#endif
#endif /* VDF_H */
