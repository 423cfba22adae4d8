// Internal declarations shared by the HIP translation units and the C-ABI layer.
// Nothing here is part of the public boundary (include/vdf.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vdf.h"
#include "resize_dispatch.h"

namespace vdf {

// ---- Hamming search tiling -------------------------------------------------------------------
// One workgroup = 4 waves; every lane keeps ROWS_PER_LANE target hashes in VGPRs and streams the
// candidate hashes through SGPRs (scalar loads), so a workgroup covers TILE_ROWS targets.
constexpr int kDefaultRowsPerLane = 2;                       // tile = 256 * rows-per-lane targets
constexpr uint32_t kDefaultChunkCols = 2048;                // candidates per workgroup

struct SearchLaunch {
    // rows (targets / references)
    const uint32_t *row_hashes;  // [n_rows][32] dwords
    const uint32_t *row_perm;    // nullable: row position -> index into row_hashes / reported row
    uint32_t n_rows;
    uint32_t row_index_base;     // added to the reported row
    // columns (candidates)
    const uint32_t *col_hashes;  // [n_cols][32]
    uint32_t n_cols;
    // fp4-expanded copies for the MFMA backend ([n_pad][512 B], zero padded); null for the VALU backend
    const void *row_exp, *col_exp;
    // popcounts of rows / columns as floats: [pop | popk | popkT] x n_pad
    const float *row_pop3 = nullptr, *col_pop3 = nullptr;
    uint32_t row_pad = 0, col_pad = 0;
    void *cand = nullptr;                 // suspect queue of the MFMA kernel (16-byte entries, pre-filled with 0xFF)
    uint32_t cand_capacity = 0;
    unsigned long long *cand_head = nullptr;  // next free slot (advanced in per-wave chunks; keeps counting past a full queue)
    // windows + tiles (device scratch, filled by launch_windows_tiles)
    uint32_t *row_lo, *row_hi;   // [n_row_tiles * tile_rows]
    uint32_t *tile_lo, *tile_hi, *tile_first, *tile_count, *tile_offset;  // [n_row_tiles (+1)]
    uint32_t *group_cmin, *group_offset, *group_blocks;  // [n_groups (+1)]: chunk-major grouped order (MFMA backend), else null
    int prune_step = 16;                  // MFMA backend: k-step after which a block that cannot contain a hit stops (16 = never)
    uint32_t shard_index = 0, shard_count = 1;  // row tiles t with t % shard_count == shard_index are this launch's
    uint32_t n_groups, group_size;        // n_groups = 0 for the VALU backend (compact tile list)
    uint32_t n_row_tiles;
    uint32_t tile_rows;          // 256 * rows-per-lane (256, 512 or 1024)
    uint32_t chunk_cols;
    uint32_t tol;
    const uint32_t *matched;     // nullable bitmap over column indices (self mode: rows too)
    int self_mode;
    // output
    vdf_hit *hits;
    unsigned long long capacity;
    unsigned long long *counters;  // [0] hits produced, [1] pairs computed, [2] pairs admitted
    uint32_t *overflow_row;
};

// mode 0: self-search windows [i+1, rhs(i));  mode 1: reference windows [lhs(r), rhs(r)).
hipError_t launch_windows_tiles(int mode, const uint32_t *col_dur, uint32_t n_cols, const uint32_t *row_dur,
                                const uint32_t *row_perm, uint32_t n_rows, uint32_t row_begin, uint32_t row_end,
                                uint32_t shard_index, uint32_t shard_count, const SearchLaunch &L,
                                hipStream_t stream);
hipError_t launch_hamming_tiles(const SearchLaunch &L, uint32_t total_tiles, hipStream_t stream);
// MFMA backend: {0, 1} fp4 encoding, exact.  Rows are padded to a multiple of the tile, columns by a further 128.
constexpr uint32_t kMfmaRowPad = 512, kMfmaColPad = 128;  // kMfmaRowPad = rows of the largest MFMA workgroup tile (8 waves x 64 rows)
// {0, 1} nibbles + popcount arrays pop3 = [pop | popk | popkT] of n_pad floats each
hipError_t launch_expand_fp4(const uint32_t *packed, uint32_t n, uint32_t n_pad, void *expanded, uint32_t k_steps,
                             float *pop3, hipStream_t stream);
hipError_t launch_hamming_tiles_mfma2(const SearchLaunch &L, uint32_t total_tiles, hipStream_t stream);  // branch-free stream; tile_rows 512 or 256
hipError_t launch_resolve_candidates(const SearchLaunch &L, hipStream_t stream);  // its second pass: suspects evaluated exactly
// drops the hits of rows that can never become targets of the greedy replay (hamming.hip); needs the COMPLETE hit set: a sharded
// launch ORs has_in (after step 1) and covered (after step 2) over its shards.  bitmaps = has_in | covered, ceil(n_entries / 32) words each
hipError_t launch_filter_mark_incoming(const vdf_hit *hits, unsigned long long n_hits, uint32_t n_entries, uint32_t *bitmaps, hipStream_t stream);
hipError_t launch_filter_mark_covered(const vdf_hit *hits, unsigned long long n_hits, uint32_t n_entries, uint32_t *bitmaps, hipStream_t stream);
hipError_t launch_filter_compact(const vdf_hit *hits, unsigned long long n_hits, uint32_t n_entries, const uint32_t *bitmaps, vdf_hit *out,
                                 unsigned long long *counter, hipStream_t stream);
hipError_t launch_bitmap_or(uint32_t *dst, const uint32_t *srcs, size_t n_words, uint32_t n_srcs, hipStream_t stream);  // dst |= OR of n_srcs bitmaps
hipError_t launch_group_max_distance(const uint32_t *hashes, const unsigned long long *offsets,
                                     const unsigned long long *members, const uint32_t *ref_hashes,
                                     const long long *ref_index, uint32_t n_groups, uint32_t *out, hipStream_t stream);

// ---- hash construction -----------------------------------------------------------------------
struct ResizeAxisTable {  // device pointers, one axis
    const int32_t *start;  // [16]
    const int32_t *size;   // [16]
    const int16_t *w;      // [16][window]
    int32_t window;
    int32_t precision;
};

hipError_t launch_resize_generic(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                 size_t clip_stride, ResizeAxisTable th, ResizeAxisTable tv, int need_h, int need_v,
                                 int32_t y_first, int32_t tmp_rows, uint8_t *small, hipStream_t stream);
struct MfmaResizeArgs {  // device pointers to the MFMA-layout tables (resize_tables.h: MfmaAxisTable)
    const void *bh, *av;
    const int32_t *bias_h, *bias_v;
    int32_t prec_h, prec_v, n_kt, n_rg;
    const int32_t *band_meta = nullptr;  // bh in band form (resize_tables.h): kt_lo[16], nt[16] on the device
    int32_t band_stride = 0;
    int32_t no_persistent = 0;         // debugging: force the one-clip-per-workgroup fused kernel
    int32_t persistent_wgs_per_cu = 3; // resident workgroups per CU for the persistent kernel
    int32_t wavestream_knob = 0;       // vdf_ctx::wavestream_knob (resize_dispatch.h): the launcher must decide as the caller did
};
hipError_t launch_resize_dct_fused(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                   size_t clip_stride, const uint8_t *buf_end, const MfmaResizeArgs &a,
                                   const double *cos_table, uint64_t *out_hashes, uint32_t *out_dontcare,
                                   hipStream_t stream);
hipError_t launch_resize_mfma_frames(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h,
                                     size_t frame_stride, size_t clip_stride, const uint8_t *buf_end,
                                     const MfmaResizeArgs &a, uint8_t *small, bool wide, hipStream_t stream);
// linear-stream form for tightly packed frames whose width is a multiple of 16 but not of the 128-byte line
// (a.av in kMfmaLayoutVertical order); resize_stream_eligible says whether a call qualifies
// clips / tables (both or neither): per-clip row ranges - clip c contributes rows y0 .. y0 + h of its frames (full-width crop
// boxes: top / bottom letterbox bars), resized with vertical table entry v_table; a.av / a.bias_v / a.prec_v / a.n_rg are then unused
struct CropStreamClip;
struct CropStreamTable;
hipError_t launch_resize_mfma_frames_stream(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h,
                                            size_t frame_stride, size_t clip_stride, const MfmaResizeArgs &a,
                                            uint8_t *small, hipStream_t stream, const CropStreamClip *clips = nullptr,
                                            const CropStreamTable *tables = nullptr);
// K-split form for wide frames (1024..4096 columns, a multiple of 16): horizontal table in registers, a.bh in plain
// kMfmaLayoutHorizontal form, a.av in kMfmaLayoutVertical order
hipError_t launch_resize_mfma_frames_ksplit(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h,
                                            size_t frame_stride, size_t clip_stride, const MfmaResizeArgs &a,
                                            uint8_t *small, hipStream_t stream, const CropStreamClip *clips = nullptr,
                                            const CropStreamTable *tables = nullptr);
// ---- letterbox crop detection + cropped resize -------------------------------------------
struct CropClipDesc {  // per clip: crop box inside the W x H frame and the table entries for its size
    uint32_t x0, y0, w, h;
    uint32_t h_table, v_table;
    uint32_t src_clip, pad;  // which clip of the caller's batch (a launch may cover a subset: see CropStreamClip::src_clip)
};
struct CropTableEntry {  // one MFMA-layout axis table (device pointers)
    const void *operand;
    const int32_t *bias;
    int32_t n_tiles, precision;
};
// The same for the linear-stream form (resize_mfma_cropped_stream_kernel): LDS row pitch, the chunk geometry and the
// reciprocal the DMA lanes divide by are fixed per clip on the host; the horizontal table is in band form
struct CropStreamClip {
    uint32_t x0, y0, w, h;
    uint32_t wp, nb, n_chunks, src_clip;  // LDS pitch (odd multiple of 16 >= w + 3, or the frame pitch for a full-width box), 16-row blocks per chunk, chunks per frame;
                                          // src_clip: which clip of the caller's batch this entry is (a launch may cover a subset of the batch: frames are
                                          // read from, and the 16 x 16 results written to, clip src_clip's place)
    uint32_t h_table, v_table, step_rows, step_x;  // 4096 = step_rows * wp + step_x: what one DMA instruction advances a lane by
};
struct CropStreamTable {
    const void *operand;
    const int32_t *bias;
    const int32_t *meta;  // band form: kt_lo[16], nt[16] (horizontal tables only)
    int32_t n_tiles, precision, band_stride, pad;
};
hipError_t launch_resize_mfma_cropped_stream(const uint8_t *frames, size_t n_clips, uint32_t pitch, uint32_t frame_rows,
                                             size_t frame_stride, size_t clip_stride, const CropStreamClip *clips,
                                             const CropStreamTable *tables, int cls, bool shift, uint8_t *small,
                                             hipStream_t stream);  // shift: some row of some box starts off a dword boundary
// crop boxes that share their column range (x0, box_w; e.g. the 4 : 3 picture of every pillarboxed clip in a 16 : 9 batch): the per-wave
// stream kernel gathers the box's bytes of each row; a = the band table of box_w; per-clip rows as in the ROWCROP launches
hipError_t launch_resize_mfma_box_wavestream(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                             size_t clip_stride, const MfmaResizeArgs &a, uint32_t x0, uint32_t box_w, const CropStreamClip *clips,
                                             const CropStreamTable *tables, uint8_t *small, hipStream_t stream);
// work: scratch of letterbox_work_bytes(n_clips, frames_per_clip) bytes (the list of frames whose side bars a second pass walks)
size_t letterbox_work_bytes(size_t n_clips, uint32_t frames_per_clip);
// side_strips: 0 = by frame height (32 column strips per pass from 512 rows, 16 from 256, else 8), 16 = at most 16 (VDF_LB_NC16: A/B runs)
hipError_t launch_letterbox(const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w, uint32_t h,
                            size_t frame_stride, size_t clip_stride, uint32_t *crops, uint32_t *work, hipStream_t stream, int side_strips = 0);
// small frames (at most 128 rows, 256 columns): one workgroup per clip, resize + DCT + hash; vertical tables in kMfmaLayoutVertical order
hipError_t launch_resize_dct_cropped_small(const uint8_t *frames, size_t n_clips, uint32_t pitch, size_t frame_stride,
                                           size_t clip_stride, const uint8_t *buf_end, const CropClipDesc *desc,
                                           const CropTableEntry *tables, const double *cos_table, uint64_t *out_hashes,
                                           uint32_t *out_dontcare, hipStream_t stream);
// the same kernel fed from the device (round 6): boxes = the detect's output [n_clips][4] {left, right, top, bottom} in device memory, tables = the
// (w, h) set of ALL box sizes - horizontal table of box width bw at [bw], vertical table of box height bh at [w + 1 + bh] (api.cpp: box_table_set)
hipError_t launch_resize_dct_cropped_small_boxes(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                                 size_t clip_stride, const uint8_t *buf_end, const uint32_t *boxes,
                                                 const CropTableEntry *tables, const double *cos_table, uint64_t *out_hashes,
                                                 uint32_t *out_dontcare, hipStream_t stream);
// frames of at most 64 x 64: detect + crop + resize + DCT + hash in one persistent kernel (dct_hash.hip: letterbox_resize_dct_hash_small_kernel).
// Every clip's 16-byte loads must stay inside the caller's buffer: the caller keeps the clips within 64 bytes of its end out of this launch.
// box_tables: the set's tables at kSmallBoxTableStride bytes each, in the order above: operand hi | lo (2 x 1024 B), bias[16], precision.
// out_crops: DEVICE [n_clips][4], always written.  wgs_per_cu: 0 = what fits.
constexpr uint32_t kSmallBoxTableStride = 2176;
hipError_t launch_letterbox_hash_small(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                       const void *box_tables, const double *cos_table, uint64_t *out_hashes, uint32_t *out_dontcare,
                                       uint32_t *out_crops, int wgs_per_cu, hipStream_t stream);
hipError_t launch_resize_mfma_cropped(const uint8_t *frames, size_t n_clips, uint32_t pitch, size_t frame_stride,
                                      size_t clip_stride, const uint8_t *buf_end, const CropClipDesc *desc,
                                      const CropTableEntry *tables, uint8_t *small, bool wide, hipStream_t stream);
// ---- Search::sort on the device (sort_order.hip) ----------------------------------------------------------------
// perm_out = the stable order by (duration, path rank); rank nullable (all paths equal).  scratch from sort_order_scratch_bytes.
size_t sort_order_scratch_bytes(uint32_t n, bool with_rank);
hipError_t launch_sort_order(const uint32_t *dur, const uint32_t *rank, uint32_t n, uint32_t *perm_out, void *scratch,
                             size_t scratch_bytes, hipStream_t stream);
// Search::sort's order of cache entries computed from their PATHS on the device (sort_order.hip): facts first (are the paths plain, how
// long, what do they share: 16 bytes {not_plain, max_len, shared, pad} at d_facts16), then the stable order by (duration, path).
hipError_t launch_path_facts(const char *d_blob, const unsigned long long *d_off, const uint32_t *d_sel, uint32_t n, void *d_facts16, hipStream_t stream);
size_t path_order_scratch_bytes(uint32_t n);
hipError_t launch_path_duration_order(const char *d_blob, const unsigned long long *d_off, const uint32_t *d_sel, const uint32_t *d_dur, uint32_t n,
                                      uint32_t first_word, uint32_t n_words, uint32_t *order_out, void *scratch, size_t scratch_bytes,
                                      hipStream_t stream);
// hashes_out[k] = hashes[perm[k]], dur_out[k] = dur[perm[k]] (dur / dur_out nullable)
// hit list into (row, col) order, in place on the device; rows < 2^row_bits
size_t sort_hits_scratch_bytes(size_t n);
hipError_t launch_sort_hits(vdf_hit *hits, size_t n, unsigned row_bits, void *scratch, size_t scratch_bytes, hipStream_t stream,
                            unsigned col_bits = 32);  // columns < 2^col_bits
hipError_t launch_gather_hashes(const uint64_t *hashes, const uint32_t *dur, const uint32_t *perm, uint32_t n, uint64_t *hashes_out,
                                uint32_t *dur_out, hipStream_t stream);
hipError_t launch_dct_hash(const uint8_t *small, size_t small_clip_stride, size_t small_frame_stride, size_t n_clips,
                           const double *cos_table, uint64_t *out_hashes, uint32_t *out_dontcare, hipStream_t stream);

}  // namespace vdf
