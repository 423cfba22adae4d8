// Host-only check of the MFMA coefficient-table builders (csrc/resize_tables.cpp): the band form of the horizontal table
// must expand to exactly the plain form, every layout must hold the same coefficients, and hi/lo must recombine to the
// scalar table's weights.  Built with g++ (no HIP).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../vid_dup_finder_lib_amd/csrc/resize_tables.h"

using namespace vdf;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); fails++; } } while (0)

// coefficient C[o][k] of a table in the given layout, recombined from hi / lo
static int coef(const MfmaAxisTable &t, int layout, int o, int k)
{
    const int tile = k / 64, r = k % 64;
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) {
            if ((l & 15) != o) continue;
            const int g = l >> 4;
            const int pos = layout == kMfmaLayoutVertical ? 16 * (j >> 2) + 4 * g + (j & 3) : layout == kMfmaLayoutVerticalWide ? 8 * (j >> 1) + 2 * g + (j & 1) : 16 * g + j;
            if (pos != r) continue;
            const int hi = t.operand[(((size_t)tile * 2 + 0) * 64 + l) * 16 + j], lo = t.operand[(((size_t)tile * 2 + 1) * 64 + l) * 16 + j];
            return 256 * hi + lo;
        }
    return 0x7fffffff;
}

int main()
{
    const unsigned sizes[] = {16, 17, 33, 64, 100, 270, 333, 480, 854, 1024, 1080, 1366, 1920, 1984, 2160, 3840};
    for (unsigned n : sizes) {
        HostAxisTable h;
        CHECK(build_axis_table(n, 16, h), "axis table %u", n);
        MfmaAxisTable plain, vert, wide, band;
        CHECK(build_mfma_axis_table(n, kMfmaLayoutHorizontal, plain) && build_mfma_axis_table(n, kMfmaLayoutVertical, vert) &&
              build_mfma_axis_table(n, kMfmaLayoutVerticalWide, wide) && build_mfma_axis_table(n, kMfmaLayoutHorizontalBand, band), "mfma tables %u", n);
        if (!plain.ok) continue;  // coefficients that do not fit the i8 split: the caller takes the scalar kernel
        const int n_kt = plain.n_tiles;
        CHECK(n_kt == (int)((n + 63) / 64) && band.n_tiles == n_kt && band.precision == plain.precision, "tile counts %u", n);
        // plain form == scalar table (identity for n == 16: weight 256 at precision 8)
        for (int o = 0; o < 16; o++) {
            long sum = 0;
            for (int k = 0; k < n_kt * 64; k++) {
                int want = 0;
                if (n == 16) want = k == o ? 256 : 0;
                else if (k >= h.start[o] && k < h.start[o] + h.size[o]) want = h.w[(size_t)o * h.window + (k - h.start[o])];
                const int got = coef(plain, kMfmaLayoutHorizontal, o, k);
                CHECK(got == want, "plain n=%u o=%d k=%d: %d != %d", n, o, k, got, want);
                CHECK(coef(vert, kMfmaLayoutVertical, o, k) == want && coef(wide, kMfmaLayoutVerticalWide, o, k) == want, "vertical layouts n=%u o=%d k=%d", n, o, k);
                sum += want;
            }
            CHECK(plain.bias[o] == (1 << (plain.precision - 1)) + 128 * sum && band.bias[o] == plain.bias[o], "bias n=%u o=%d", n, o);
        }
        // band form: every fragment inside an output's band equals the plain fragment, everything outside the band is zero in the plain form
        CHECK((int)band.band_meta.size() == 32 && band.band_stride >= 128 + 32 && band.operand.size() == (size_t)16 * band.band_stride, "band shape %u", n);
        if (band.ok) CHECK(band.band_stride <= kMfmaBandMaxTiles * 128 + 32, "band stride %u", n);
        for (int o = 0; o < 16; o++) {
            const int lo_t = band.band_meta[o], nt = band.band_meta[16 + o];
            CHECK(lo_t >= 0 && nt >= 1 && lo_t + nt <= n_kt, "band range n=%u o=%d", n, o);
            for (int kt = 0; kt < n_kt; kt++)
                for (int g = 0; g < 4; g++)
                    for (int hl = 0; hl < 2; hl++)
                        for (int b = 0; b < 16; b++) {
                            const int l = 16 * g + o;
                            const int p = plain.operand[(((size_t)kt * 2 + hl) * 64 + l) * 16 + b];
                            const int j = kt - lo_t;
                            const int q = (j >= 0 && j < nt) ? band.operand[(size_t)o * band.band_stride + (size_t)j * 128 + hl * 64 + g * 16 + b] : 0;
                            CHECK(p == q, "band n=%u o=%d kt=%d g=%d hl=%d b=%d: %d != %d", n, o, kt, g, hl, b, q, p);
                        }
        }
    }
    // Round 6: the letterbox path of small frames keeps the tables of EVERY crop-box size resident (csrc/api.cpp: box_table_set) so that the
    // boxes never visit the host.  That route needs every size from 1 to 256 to fit the i8 split in both layouts, every size up to 64 to be
    // one tile (the fused kernel finds a table by multiplication: 2 KB of operand per table), and the bias / precision of a size to be the same
    // in both layouts (one blob entry per (size, layout)).
    for (unsigned n = 1; n <= 256; n++) {
        MfmaAxisTable hz, vt;
        CHECK(build_mfma_axis_table(n, kMfmaLayoutHorizontal, hz) && build_mfma_axis_table(n, kMfmaLayoutVertical, vt), "box tables %u", n);
        CHECK(hz.ok && vt.ok, "box size %u does not fit the i8 split: the device-side letterbox route would fall back to the host plan", n);
        CHECK(hz.n_tiles == (int)((n + 63) / 64) && vt.n_tiles == hz.n_tiles && hz.precision == vt.precision && hz.precision >= 1 && hz.precision <= 15, "box shape %u", n);
        CHECK(hz.operand.size() == (size_t)hz.n_tiles * 2048 && vt.operand.size() == hz.operand.size() && hz.bias.size() == 16 && hz.bias == vt.bias, "box sizes %u", n);
        if (n <= 64) CHECK(hz.n_tiles == 1, "box size %u must be one tile", n);
    }
    if (fails == 0) std::printf("resize tables ok\n");
    return fails ? 1 : 0;
}
