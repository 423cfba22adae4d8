#include "resize_dispatch.h"

namespace vdf {

uint32_t stream_pitch(uint32_t w)
{
#ifdef VDF_STREAM_REPITCH_ALL  // experiment: every line-aligned pitch from 640 columns up is re-pitched to an odd multiple of 16 bytes
    if (w % 16 == 0) return ((w % 256 == 0 && w >= 768) || (w % 128 == 0 && w >= 640 && (w / 16) % 2 == 0)) ? w + 16 : w;
#else
    if (w % 16 == 0) return (w % 256 == 0 && w >= 768) ? w + 16 : w;
#endif
    uint32_t wp = (w + (w % 4 ? 3u : 0u) + 15u) & ~15u;
    if ((wp / 16) % 2 == 0) wp += 16;
    return wp;
}

uint32_t stream_blocks_per_chunk(uint32_t wp, int buf_bytes)
{
    for (uint32_t nb = 4; nb >= 1; nb--)
        if ((size_t)((16u * nb * wp + 1023u) & ~1023u) + 128u <= (size_t)buf_bytes) return nb;  // a DMA instruction fills whole KBs
    return 0;
}

int stream_class(uint32_t w, uint32_t *nb)
{
    const int n_kt = (int)((w + 63) / 64);
    const uint32_t wp = stream_pitch(w);
    if (n_kt <= kStreamTabS && stream_blocks_per_chunk(wp, kStreamBufS) == 4) { *nb = 4; return 1; }
    *nb = stream_blocks_per_chunk(wp, kStreamBufM);
    if (n_kt <= kStreamTabM) return *nb >= 2 ? 2 : 0;
    return *nb >= 2 ? 3 : 0;
}

static bool wavestream_fits(uint32_t w, int nw)
{
    const uint32_t need = ((16u * stream_pitch(w) + 1023u) & ~1023u) + 128u;  // a DMA instruction fills whole KBs
    const int buf = nw == 3 ? kWaveStreamBuf3 : nw == 4 ? kWaveStreamBuf : nw == 5 ? kWaveStreamBuf5 : nw == 6 ? kWaveStreamBuf6 : nw == 8 ? kWaveStreamBuf8 : 0;
    return need <= (uint32_t)buf && w >= 256;
}

int resize_wavestream_waves(uint32_t w, int knob)
{
    if (knob < 0) return 0;  // VDF_NO_WAVESTREAM (measurements / tests): the chunk kernel everywhere
    if (knob > 0) return wavestream_fits(w, knob) ? knob : 0;  // VDF_WAVESTREAM_NW
    uint32_t nb = 0;
    const int cls = stream_class(w, &nb);
    // Frames up to 512 wide keep the chunk kernel (two workgroups per CU, whole table in LDS: 480 x 270 6.5 TB/s against 6.0 with eight
    // waves).  From there on one block stream per wave, with as many waves as block buffers fit: measured against the chunk kernel
    // (gpurun_out/r03_nw_sweep2.txt) 576 wide 5.9 -> 6.8 TB/s, 640 6.4 -> 6.8, 1152 5.3 -> 6.8, 1200 6.1 -> 6.7, 1366 5.4 -> 6.3,
    // 1440 5.6 -> 6.3, 1520 6.0 -> 6.8; level at 896 / 960.  (With FOUR waves the three-block widths had measured slower: 1280 x 720
    // 6.6 -> 5.8 - the point is the bytes in flight per CU, not the absence of the barrier.)
    if (cls == 1) return 0;
    for (int nw : {8, 6, 5, 4})
        if (cls >= 2 && wavestream_fits(w, nw)) return nw;
    // Wider than the four-wave buffers: three waves with 37 KB blocks (pitches up to 2368) - for the widths the K-split kernel cannot take
    // (not a multiple of 16: 1950 x 1096 3.2 -> 5.2 TB/s against the whole-line kernel).  Multiples of 16 stay with the K-split form,
    // which measured 2-4 % ahead of three waves (2048 wide 5.79 against 5.53, 2304 5.88 against 5.63; gpurun_out/r03nw3).
    if (w % 16 != 0 && wavestream_fits(w, 3)) return 3;
    return 0;
}

uint32_t box_stream_pitch(uint32_t frame_w, uint32_t x0, uint32_t box_w, int *mode)
{
    if (x0 == 0 && box_w == frame_w) {  // whole rows
        const uint32_t wp = stream_pitch(frame_w);
        *mode = wp == frame_w ? 0 : frame_w % 4 == 0 ? 1 : 2;
        return wp;
    }
    // a box: gathered row by row.  Rows of the box start at (row * frame_w + x0): off a dword unless both are multiples of 4, and then
    // the LDS row holds the up to 3 bytes in front of the box's first pixel too (MODE 2 shifts them out of the operands)
    const bool shifted = frame_w % 4 != 0 || x0 % 4 != 0;
    *mode = shifted ? 2 : 1;
    uint32_t wp = (box_w + (shifted ? 3u : 0u) + 15u) & ~15u;
    if ((wp / 16) % 2 == 0) wp += 16;  // an odd multiple of 16 bytes: the 16 rows of a block in 16 different bank groups
    return wp;
}

int resize_wavestream_waves_box(uint32_t frame_w, uint32_t x0, uint32_t box_w, int knob)
{
    if (x0 == 0 && box_w == frame_w) return resize_wavestream_waves(frame_w, knob);
    if (knob < 0 || box_w < 513) return 0;  // narrower boxes: the gather kernel with two workgroups per CU
    int mode = 0;
    const uint32_t need = ((16u * box_stream_pitch(frame_w, x0, box_w, &mode) + 1023u) & ~1023u) + 128u;
    for (int nw : {8, 6, 5, 4, 3}) {
        const int buf = nw == 3 ? kWaveStreamBuf3 : nw == 4 ? kWaveStreamBuf : nw == 5 ? kWaveStreamBuf5 : nw == 6 ? kWaveStreamBuf6 : kWaveStreamBuf8;
        if (need <= (uint32_t)buf) return nw;
    }
    return 0;
}

bool resize_wavestream_table_fits(int nw, int band_stride)
{
    const int tab_bytes = 16 * band_stride + 128;  // + the zero slot
    return tab_bytes <= (nw <= 4 ? kWaveStreamTabBytes : nw == 5 ? kWaveStreamTabMid : kWaveStreamTabSmall);
}

bool resize_stream_wants_band(uint32_t w, int knob) { return resize_wavestream_waves(w, knob) != 0; }

bool resize_wavestream_applies(uint32_t w, int knob) { return resize_wavestream_waves(w, knob) != 0; }

bool resize_rowcrop_streams(uint32_t w)
{
    // Full-width crop boxes (top / bottom bars) through the ROWCROP stream kernels against the general cropped kernels, detect + crop +
    // hash of clips with 12 % bars (gpurun_out/r03v, r03w, r03y): 1600 wide 3.83 -> 2.87 ms, 1152 3.51 -> 3.12, 1280 4.39 -> 3.91,
    // 1366 1.73 -> 1.51, 3840 5.29 -> 4.65, 1920 4.76 -> 4.33, 1536 3.13 -> 2.90, 1024 2.80 -> 2.64, 640 / 768 / 854 -3 %, 2560 level;
    // 2048 wide (the K-split form with two-block chunks) lost 4 % to the whole-line cropped kernel and stays there.
    return w != 2048;
}

bool resize_stream_eligible(const uint8_t *frames, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, int knob)
{
    // narrow tall frames (portrait video) gain the most: 240 x 426 4.6 -> 6.1 TB/s, 160 x 200 3.5 -> 4.2 against the whole-line kernels
    if (w < 64 || (uint64_t)w * h >= (1ull << 31)) return false;
    // frames must also END on a 16-byte boundary: the buffer resource is sized to the frame and the range check drops a dword that
    // straddles its end (203 x 301 frames lost their last pixels: found by the strided-buffer test)
    if (((uint64_t)w * h) % 16 != 0) return false;
    if (((uintptr_t)frames | frame_stride | clip_stride) % 16 != 0) return false;
    uint32_t nb = 0;
    return stream_class(w, &nb) == 1 || resize_wavestream_applies(w, knob);  // the chunk form (frames up to 512 wide) or one block stream per wave
}

bool resize_short_prefers_stream(uint32_t w, uint32_t h)
{
    // Frames of at most 128 rows fuse resize and DCT in one kernel, one workgroup per clip (a frame at a time: 4 - 16 KB in flight per
    // workgroup).  That is the faster form for small frames only.  Measured against the linear-stream kernels + dct_hash_kernel, round 5
    // (TB/s of frame bytes, fused -> stream; gpurun_out/r05u, r05w = profiles/r05_short_frames.txt): 176 x 99 4.1 -> 4.1, 192 x 108 4.7 -> 4.8,
    // 160 x 120 4.4 -> 4.7, 200 x 112 3.7 -> 4.8, 208 x 117 3.6 -> 5.4, 224 x 126 3.9 -> 5.8, 256 x 128 4.8 -> 6.1, 320 x 96 4.5 -> 6.1,
    // 480 x 128 4.1 -> 6.4, 640 x 120 3.9 -> 5.8, 854 x 128 2.5 -> 5.7, 1920 x 128 3.5 -> 6.0, 1920 x 64 4.7 -> 6.0; the other way:
    // 128 x 128 4.8 -> 4.3, 160 x 90 4.2 -> 3.8, 192 x 80 4.3 -> 4.0, 128 x 96 4.6 -> 3.5, 512 x 64 4.8 -> 4.2 (one chunk per frame).
    // Later in round 5 frames of up to 256 x 128 with W % 16 == 0 got a persistent kernel of their own (resize_dct_hash_tiled_kernel), which moved
    // the line for those widths (tiled -> stream, gpurun_out/r06h against r05x): 160 x 120 5.3 -> 4.9, 192 x 108 5.6 -> 5.2, 192 x 128 5.7 -> 5.5,
    // 208 x 117 5.4 -> 5.4, 224 x 126 5.8 -> 6.3, 256 x 128 5.7 -> 6.5.
    if (h > 128) return false;  // (not asked: such frames never fuse)
    if (w > 512) return h >= 64;  // the per-wave form
    // (r06i, the same box for both: 192 x 128 5.7 -> 6.0, 208 x 117 5.5 -> 5.9, 240 x 100 5.4 -> 5.6, 256 x 96 6.0 -> 5.8: the line is at 24 000 pixels)
    if (w <= 256 && w % 16 == 0) return h > 64 && (uint64_t)w * h >= 24000;  // the tiled kernel's widths
    return h > 64 && (uint64_t)w * h >= 19000;
}

bool resize_tall_prefers_tiled(uint32_t w, uint32_t h)
{
    // Narrow frames of 129 ... 256 rows: the chunk-stream kernel moves 64 rows at a time whatever the width - 11 KB of a 176-wide frame, 4 KB of
    // a 64-wide one, two workgroups per CU: latency-bound.  The tiled persistent kernel (four row groups) keeps 16 KB per wave in flight.
    // Measured, stream -> tiled (TB/s, gpurun_out/r06l = profiles/r05_short_frames.txt): 64 x 160 2.4 -> 5.5, 64 x 256 2.7 -> 4.9, 80 x 240 3.1 -> 5.2,
    // 96 x 160 3.2 -> 5.1, 112 x 200 3.5 -> 5.4, 128 x 160 4.0 -> 5.4, 128 x 256 4.8 -> 5.4, 160 x 144 4.3 -> 4.9, 160 x 200 4.6 -> 5.0,
    // 176 x 144 4.6 -> 5.1; level at 144 x 256, 176 x 208, 192 x 144; the other way from there: 208 x 160 5.4 -> 5.0, 240 x 160 6.1 -> 5.0.
    // (any width since the persistent kernels take the last clip apart: 100 x 200 3.1 -> 4.7, gpurun_out/r06z)
    return h > 128 && h <= 256 && w >= 16 && w <= 176 && (uint64_t)w * h <= 36000;
}

uint32_t ksplit_geometry(uint32_t w, uint32_t *wp)
{
    uint32_t p = w;  // w % 16 == 0
    if ((p / 16) % 2 == 0) p += 16;
    *wp = p;
    return stream_blocks_per_chunk(p, kKsplitBuf);
}

bool resize_ksplit_eligible(const uint8_t *frames, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride)
{
    if (w % 16 != 0 || w < 1024 || w > 4096 || (uint64_t)w * h >= (1ull << 31)) return false;
    if (((uintptr_t)frames | frame_stride | clip_stride) % 16 != 0) return false;
    uint32_t wp = 0;
    return ksplit_geometry(w, &wp) >= 1;
}

bool resize_cropped_stream_class(uint32_t pitch, int *cls)
{
    if (pitch < 64 || pitch > 1984) return false;
    uint32_t nb = 0;
    const int c = stream_class(pitch | 1u, &nb);  // | 1: size the buffers for the re-pitched form of a full-width box
    if (c == 0) return false;
    *cls = c == 1 ? 1 : 2;
    return true;
}

uint32_t resize_cropped_stream_blocks(uint32_t crop_w, uint32_t x0, uint32_t pitch, int cls, uint32_t *wp)
{
    uint32_t p = (crop_w + 3u + 15u) & ~15u;
    if ((p / 16) % 2 == 0) p += 16;
    // full-width box (top / bottom bars): the DMA is a linear copy - unless that pitch puts a block's 16 rows on one bank group
    if (x0 == 0 && crop_w == pitch && pitch % 16 == 0 && pitch % 256 != 0) p = pitch;
    *wp = p;
    return stream_blocks_per_chunk(p, cls == 1 ? kStreamBufS : kStreamBufM);
}

}  // namespace vdf
