#pragma once
#include <cstdint>
#include <vector>

namespace vdf {

struct HostAxisTable {
    std::vector<int32_t> start;  // [out] first source index
    std::vector<int32_t> size;   // [out] taps
    std::vector<int16_t> w;      // [out][window]
    int32_t window = 0;
    int32_t precision = 0;
};

bool build_axis_table(uint32_t in_size, uint32_t out_size, HostAxisTable &t);

// The same coefficients laid out as v_mfma_i32_16x16x64_i8 operands.  Each i16 coefficient c is split
// c = 256 * hi + lo with lo in [-128, 127]; pixels are centred (p - 128) so both factors are signed i8 and
// bias = 2^(precision-1) + 128 * sum(c) restores the unsigned sum exactly.
struct MfmaAxisTable {
    // n_tiles x {hi, lo} x 64 lanes x 16 bytes.
    //   horizontal (B operand): byte j of lane l in tile kt  = C[o = l & 15][x = 64 kt + 16 (l >> 4) + j]
    //   vertical   (A operand): byte j = 4 m + r of lane l in group rg = C[oy = l & 15][y = 64 rg + 16 m + 4 (l >> 4) + r]
    //   vertical, wide kernel:  byte j of lane l in group rg           = C[oy = l & 15][y = 64 rg + 8 (j >> 1) + 2 (l >> 4) + (j & 1)]
    //   horizontal, band form:  only the K tiles an output's taps reach are stored, output-major: the 16 bytes of lane
    //                           (o, g) for tile kt and half hl sit at  o * band_stride + (kt - kt_lo[o]) * 128 + hl * 64 + g * 16
    //                           for kt_lo[o] <= kt < kt_lo[o] + nt[o]; every other fragment of that lane is zero
    std::vector<int8_t> operand;
    std::vector<int32_t> bias;  // [16]
    std::vector<int32_t> band_meta;  // band form only: kt_lo[16] then nt[16]
    int32_t band_stride = 0;         // band form only: bytes per output
    int32_t n_tiles = 0;
    int32_t precision = 0;
    bool ok = false;            // false if some hi part does not fit i8 (caller falls back to the generic kernel)
};

// in_size == 16 yields the identity (the reference copies when no resize is needed).
// layout: 0 = horizontal, 1 = vertical, 2 = vertical for resize_mfma_frame_wide_kernel (the k order of a 64-row group is free
// as long as both operands of the product agree; the wide kernel's lanes hold rows 8 oct + 2 g + {0, 1} of every octet).
// 3 = horizontal in band form (a Lanczos3 row of a 16-output downscale reaches 6/16 of the width, so 60 % of a wide frame's
// fragments are zero): 1920 wide shrinks from 60 KB to 27 KB, small enough to sit in LDS beside the pixel chunks of the
// linear-stream kernel.  ok = false if an output reaches more than kMfmaBandMaxTiles tiles.
enum { kMfmaLayoutHorizontal = 0, kMfmaLayoutVertical = 1, kMfmaLayoutVerticalWide = 2, kMfmaLayoutHorizontalBand = 3 };
constexpr int kMfmaBandMaxTiles = 15;
bool build_mfma_axis_table(uint32_t in_size, int layout, MfmaAxisTable &t);

}  // namespace vdf
