// Which linear-stream resize kernel a frame size gets and with what geometry (LDS row pitch, 16-row blocks per chunk).
// Host-only arithmetic (no HIP), shared by the launchers in dct_hash.hip and by api.cpp; tests/cpp/resize_dispatch_main.cpp
// checks the LDS budgets and the pitch rules for every width on the CPU.
#pragma once
#include <cstddef>
#include <cstdint>

#include "resize_tables.h"

namespace vdf {

// LDS budgets of the linear-stream kernels.  Chunk form (resize_mfma_frame_stream_kernel, frames up to 512 wide): S = two workgroups
// per CU, 64-row chunks, whole table.  The M sizes (one workgroup per CU, 32 KB of table, two 62 KB chunk buffers; the band form
// of the table for crops wider than 1024: 16 outputs x at most 15 tiles x 128 B + padding) serve the general cropped kernel only -
// uncropped M-class frames take the per-wave form.  The K-split kernel keeps its table in registers: two 75 KB buffers.
// + 128: the operand reads' overrun past the last row.
constexpr int kStreamBufS = 30 * 1024 + 128, kStreamTabS = 8;
constexpr int kStreamBufM = 62 * 1024 + 128, kStreamTabM = 16;
constexpr int kKsplitBuf = 75 * 1024 + 128;
constexpr int kWaveStreamBuf = 16 * 1920 + 128;  // one 16-row block of a frame up to 1920 wide per wave (round 3: resize_mfma_frame_wavestream_kernel)
constexpr int kWaveStreamTabBytes = 16 * (kMfmaBandMaxTiles * 128 + 32) + 128;  // band table + zero slot
// narrower frames: more waves, each with its own (smaller) block buffer, so that the blocks in flight per CU stay near 120 KB.
// A block's DMA instructions fill whole KBs: the buffers are the block rounded up to 1 KB (+ the operand reads' overrun).
constexpr int kWaveStreamBuf5 = 25 * 1024 + 128, kWaveStreamBuf6 = 21 * 1024 + 128, kWaveStreamBuf8 = 15 * 1024 + 128;  // pitches up to 1600, 1344, 960
constexpr int kWaveStreamBuf3 = 37 * 1024 + 128;  // ... and wider frames (pitches up to 2368: 2048 x 1152, 2160 x 3840 portrait) with three waves
constexpr int kWaveStreamTabMid = 16 * (12 * 128 + 32) + 128;    // band tables of at most 12 tiles per output (five waves: frames up to 1600 columns)
constexpr int kWaveStreamTabSmall = 16 * (10 * 128 + 32) + 128;  // at most 10 (six and eight waves: up to 1344 columns)
constexpr int kStreamPartBytes = 3 * 64 * 4 * 4;  // s_part: the vertical partial sums of waves 1..3
constexpr int kLdsPerCu = 160 * 1024;
static_assert(16 * (kMfmaBandMaxTiles * 128 + 32) + 128 <= kStreamTabM * 2048, "band table + zero slot fit the M class");
static_assert(2 * (2 * kStreamBufS + kStreamTabS * 2048 + kStreamPartBytes) <= kLdsPerCu, "two S workgroups per CU");
static_assert(2 * kStreamBufM + kStreamTabM * 2048 + kStreamPartBytes <= kLdsPerCu, "one M workgroup per CU");
static_assert(2 * kKsplitBuf + 2 * 3 * 64 * 16 + kStreamPartBytes <= kLdsPerCu, "one K-split workgroup per CU");
static_assert(4 * kWaveStreamBuf + kWaveStreamTabBytes + 2 * kStreamPartBytes <= kLdsPerCu, "one per-wave-stream workgroup per CU");
static_assert(3 * kWaveStreamBuf3 + kWaveStreamTabBytes + 2 * 2 * 1024 <= kLdsPerCu, "three waves");
static_assert(5 * kWaveStreamBuf5 + kWaveStreamTabMid + 2 * 4 * 1024 <= kLdsPerCu, "five waves");
static_assert(6 * kWaveStreamBuf6 + kWaveStreamTabSmall + 2 * 5 * 1024 <= kLdsPerCu, "six waves");
static_assert(8 * kWaveStreamBuf8 + kWaveStreamTabSmall + 2 * 7 * 1024 <= kLdsPerCu, "eight waves");

// LDS row pitch of the stream kernel: the frame's own for multiples of 16 - unless it is a multiple of 256, where the 16
// rows of a block would share one bank group (16-way conflict on every operand read: re-pitched, 768 / 1024 / 1280 wide
// gain 12 / 10 / 6 %; pitches with 8-way conflicts or fewer - 1920, 640, 480 - are faster left alone: the linear DMA is
// worth more than the conflicts cost); otherwise the next odd multiple of 16 that holds the row and the up to 3 bytes a
// dword-aligned row start puts in front of it.
uint32_t stream_pitch(uint32_t w);
// 16-row blocks per chunk (at most 4) that fit a buffer at that pitch; 0 = not even one
uint32_t stream_blocks_per_chunk(uint32_t wp, int buf_bytes);
// buffer class of a width: 0 none, 1 = S (the chunk kernel), 2 = M, 3 = M with the band table; *nb = 16-row blocks per chunk
int stream_class(uint32_t w, uint32_t *nb);
bool resize_stream_wants_band(uint32_t w, int knob = 0);  // the width's kernel takes a.bh in kMfmaLayoutHorizontalBand form (= the per-wave form)
// One block stream per wave (resize_mfma_frame_wavestream_kernel): every M-class width whose (re-pitched, whole-KB) block fits a wave's
// buffer - 512 .. 2368 columns - with as many waves per workgroup as buffers fit (8 / 6 / 5 / 4 / 3; 0 = the width does not take this form).
// knob (vdf_ctx::wavestream_knob, read once per context): 0 = the measured rule, n > 0 = VDF_WAVESTREAM_NW=n forces n waves where the block
// fits the n-wave buffer, -1 = VDF_NO_WAVESTREAM=1 switches the form off (measurements / tests).
int resize_wavestream_waves(uint32_t w, int knob = 0);
bool resize_wavestream_applies(uint32_t w, int knob = 0);
// a launch over crop boxes that share their column range (x0, box_w) of frames frame_w wide: LDS row pitch and addressing mode (0 linear
// copy of whole rows, 1 gather, 2 gather + the operand shift for rows that start off a dword), and the wave count (0: not this kernel)
uint32_t box_stream_pitch(uint32_t frame_w, uint32_t x0, uint32_t box_w, int *mode);
int resize_wavestream_waves_box(uint32_t frame_w, uint32_t x0, uint32_t box_w, int knob = 0);
bool resize_wavestream_table_fits(int nw, int band_stride);  // does a band table of that stride (bytes per output) fit the nw-wave kernel's table array
// Clips whose crop boxes are full-width (top / bottom bars only): do the ROWCROP instantiations of the stream kernels beat the general
// cropped kernels at this frame width?  (measured; the frame must also pass resize_stream_eligible / resize_ksplit_eligible)
bool resize_rowcrop_streams(uint32_t w);
// Does a call take a linear-stream kernel (chunk or per-wave form)?  Frames starting on 16-byte boundaries and ending on one, rows packed
// inside a frame (frames and clips may be padded), 64 .. 1920 columns.
bool resize_stream_eligible(const uint8_t *frames, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, int knob = 0);
// frames of at most 128 rows (the fused kernel's range) that measured faster on the stream kernels: wide, short ones
bool resize_short_prefers_stream(uint32_t w, uint32_t h);
// narrow frames of 129 ... 256 rows that measured faster on the tiled persistent kernel (the fused family) than on the stream kernels
bool resize_tall_prefers_tiled(uint32_t w, uint32_t h);

// K-split form (1024..4096 columns, a multiple of 16): LDS pitch (an odd multiple of 16) and blocks per chunk (0: does not fit)
uint32_t ksplit_geometry(uint32_t w, uint32_t *wp);
bool resize_ksplit_eligible(const uint8_t *frames, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride);

// cropped clips: can frames of this pitch take the stream form and with which buffers (1 = S, 2 = M); per crop box the LDS
// pitch and the blocks per chunk (a full-width box at a pitch without 16-way conflicts keeps the frame's pitch: linear DMA)
bool resize_cropped_stream_class(uint32_t pitch, int *cls);
uint32_t resize_cropped_stream_blocks(uint32_t crop_w, uint32_t x0, uint32_t pitch, int cls, uint32_t *wp);

}  // namespace vdf
