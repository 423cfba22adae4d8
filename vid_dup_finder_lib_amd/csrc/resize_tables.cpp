// Host-side Lanczos3 coefficient tables for the resize stage (product code; the parity oracle has
// its own independent restatement in oracle/).
//
// Follows what vid_dup_finder_common/src/resize_gray.rs:34-47 asks fast_image_resize 5.1 for:
// ResizeAlg::Convolution(FilterType::Lanczos3) on PixelType::U8 with the whole image as crop box.
// The crate's published algorithm: per output pixel a window of f64 weights normalised to 1,
// quantised to i16 with the largest precision p such that round(max_w * 2^(p+1)) < 2^15.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "resize_tables.h"

namespace vdf {

static double sinc_pi(double x)
{
    if (x == 0.0) return 1.0;
    x *= M_PI;
    return std::sin(x) / x;
}

static double lanczos3(double x)
{
    if (x >= -3.0 && x < 3.0) return sinc_pi(x) * sinc_pi(x / 3.0);
    return 0.0;
}

bool build_axis_table(uint32_t in_size, uint32_t out_size, HostAxisTable &t)
{
    t = HostAxisTable();
    if (in_size == 0 || out_size == 0) return false;
    const double scale = (double)in_size / (double)out_size;
    const double fscale = scale > 1.0 ? scale : 1.0;
    const double radius = 3.0 * fscale;
    const int window = (int)std::ceil(radius) * 2 + 1;
    const double recip = 1.0 / fscale;
    std::vector<double> vals((size_t)window * out_size, 0.0);
    t.start.assign(out_size, 0);
    t.size.assign(out_size, 0);
    t.window = window;
    for (uint32_t o = 0; o < out_size; o++) {
        const double in_center = ((double)o + 0.5) * scale;
        double lo = std::floor(in_center - radius), hi = std::ceil(in_center + radius);
        if (lo < 0.0) lo = 0.0;
        if (hi > (double)in_size) hi = (double)in_size;
        const uint32_t x_min = (uint32_t)lo, x_max = (uint32_t)hi;
        const double center = in_center - 0.5;
        double *k = vals.data() + (size_t)o * window;
        int n = 0;
        double ww = 0.0;
        uint32_t b0 = x_min, b1 = x_max;
        for (uint32_t x = x_min; x < x_max; x++) {
            const double w = lanczos3(((double)x - center) * recip);
            if (x == b0 && w == 0.0) b0++;  // leading zero taps are not stored
            else { k[n++] = w; ww += w; }
        }
        for (int i = n - 1; i >= 0 && b1 > b0 && k[i] == 0.0; i--) b1--;  // trailing zero taps
        if (ww != 0.0)
            for (int i = 0; i < n; i++) k[i] /= ww;
        t.start[o] = (int32_t)b0;
        t.size[o] = (int32_t)(b1 - b0);
    }
    double max_w = vals[0];
    for (double v : vals) max_w = v > max_w ? v : max_w;
    int precision = 0;
    for (int p = 0; p < 16; p++) {
        precision = p;
        if ((int32_t)std::round(max_w * (double)(1 << (p + 1))) >= (1 << 15)) break;
    }
    t.precision = precision;
    t.w.resize(vals.size());
    const double q = (double)(1 << precision);
    for (size_t i = 0; i < vals.size(); i++) t.w[i] = (int16_t)std::round(vals[i] * q);
    return true;
}

bool build_mfma_axis_table(uint32_t in_size, int layout, MfmaAxisTable &t)
{
    t = MfmaAxisTable();
    const uint32_t D = 16;
    const int32_t n_tiles = (int32_t)((in_size + 63) / 64);
    std::vector<int32_t> full((size_t)D * n_tiles * 64, 0);  // C[o][x], zero padded
    int precision;
    if (in_size == D) {
        precision = 8;  // identity: (256 (p - 128) + 128 * 256 + 128) >> 8 == p
        for (uint32_t o = 0; o < D; o++) full[(size_t)o * n_tiles * 64 + o] = 256;
    } else {
        HostAxisTable h;
        if (!build_axis_table(in_size, D, h)) return false;
        precision = h.precision;
        for (uint32_t o = 0; o < D; o++)
            for (int32_t k = 0; k < h.size[o]; k++)
                full[(size_t)o * n_tiles * 64 + h.start[o] + k] = h.w[(size_t)o * h.window + k];
    }
    if (precision < 1) return false;
    t.n_tiles = n_tiles;
    t.precision = precision;
    t.bias.assign(D, 0);
    t.operand.assign((size_t)n_tiles * 2 * 64 * 16, 0);
    t.ok = true;
    for (uint32_t o = 0; o < D; o++) {
        int64_t sum = 0;
        for (int32_t x = 0; x < n_tiles * 64; x++) sum += full[(size_t)o * n_tiles * 64 + x];
        t.bias[o] = (int32_t)((1 << (precision - 1)) + 128 * sum);
    }
    if (layout == kMfmaLayoutHorizontalBand) {
        t.band_meta.assign(32, 0);
        int32_t max_nt = 1;
        for (uint32_t o = 0; o < D; o++) {
            int32_t lo = n_tiles, hi = -1;
            for (int32_t x = 0; x < n_tiles * 64; x++)
                if (full[(size_t)o * n_tiles * 64 + x] != 0) { lo = std::min(lo, x / 64); hi = std::max(hi, x / 64); }
            if (hi < lo) { lo = 0; hi = 0; }
            t.band_meta[o] = lo;
            t.band_meta[16 + o] = hi - lo + 1;
            max_nt = std::max(max_nt, hi - lo + 1);
        }
        if (max_nt > kMfmaBandMaxTiles) t.ok = false;
        t.band_stride = max_nt * 128 + 32;  // + 32: consecutive outputs start 8 banks apart
        t.operand.assign((size_t)D * t.band_stride, 0);
        for (uint32_t o = 0; o < D; o++)
            for (int32_t j = 0; j < t.band_meta[16 + o]; j++)
                for (int g = 0; g < 4; g++)
                    for (int b = 0; b < 16; b++) {
                        const int32_t c = full[(size_t)o * n_tiles * 64 + 64 * (t.band_meta[o] + j) + 16 * g + b];
                        const int32_t lo = ((c + 128) & 255) - 128, hi = (c - lo) / 256;
                        if (hi < -128 || hi > 127) t.ok = false;
                        int8_t *q = t.operand.data() + (size_t)o * t.band_stride + (size_t)j * 128 + g * 16 + b;
                        q[0] = (int8_t)hi;
                        q[64] = (int8_t)lo;
                    }
        return true;
    }
    for (int32_t tile = 0; tile < n_tiles; tile++)
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 16; j++) {
                const int g = l >> 4, o = l & 15;
                const int pos = layout == kMfmaLayoutVertical       ? 64 * tile + 16 * (j >> 2) + 4 * g + (j & 3)
                                : layout == kMfmaLayoutVerticalWide ? 64 * tile + 8 * (j >> 1) + 2 * g + (j & 1)
                                                                    : 64 * tile + 16 * g + j;
                const int32_t c = full[(size_t)o * n_tiles * 64 + pos];
                int32_t lo = ((c + 128) & 255) - 128;  // c = 256 hi + lo, lo in [-128, 127]
                int32_t hi = (c - lo) / 256;
                if (hi < -128 || hi > 127) t.ok = false;
                t.operand[(((size_t)tile * 2 + 0) * 64 + l) * 16 + j] = (int8_t)hi;
                t.operand[(((size_t)tile * 2 + 1) * 64 + l) * 16 + j] = (int8_t)lo;
            }
    return true;
}

}  // namespace vdf
