// Host-only check of csrc/resize_dispatch.cpp: for every width the geometry the launchers will use must fit the kernels'
// LDS buffers, keep the operand reads aligned and conflict-free where the rules say so, and the documented sizes must land
// on the documented kernels.  Built with g++ (no HIP).
#include <cstdio>
#include <cstdint>

#include "../../vid_dup_finder_lib_amd/csrc/resize_dispatch.h"

using namespace vdf;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); fails++; } } while (0)

int main()
{
    const uint8_t *aligned = reinterpret_cast<const uint8_t *>(uintptr_t(0x10000));
    for (uint32_t w = 1; w <= 4200; w++) {
        // ---- stream kernel
        const uint32_t wp = stream_pitch(w);
        CHECK(wp % 16 == 0 && wp >= w, "pitch %u -> %u", w, wp);
        if (w % 16 != 0) {
            CHECK((wp / 16) % 2 == 1, "re-pitched rows must fall in 16 different bank groups: %u -> %u", w, wp);
            CHECK(wp >= w + (w % 4 ? 3u : 0u) && wp < w + 48, "pitch holds the row and the dword-alignment slack: %u -> %u", w, wp);
        } else if (w % 256 == 0 && w >= 768) {
            CHECK(wp == w + 16, "multiples of 256 are re-pitched: %u -> %u", w, wp);
        } else {
            CHECK(wp == w, "other multiples of 16 keep their pitch: %u -> %u", w, wp);
        }
        uint32_t nb = 0;
        const int cls = stream_class(w, &nb);
        const int n_kt = (int)((w + 63) / 64);
        if (cls) {
            const int buf = cls == 1 ? kStreamBufS : kStreamBufM;
            CHECK(nb >= (cls == 1 ? 4u : 2u) && nb <= 4, "blocks per chunk w=%u cls=%d nb=%u", w, cls, nb);
            // every DMA instruction fills a whole KB of LDS, and the last operand read may run 64 + 16 + 3 bytes past the chunk
            CHECK(((16 * nb * wp + 1023) & ~1023u) + 128 <= (uint32_t)buf, "chunk fits its buffer w=%u cls=%d nb=%u wp=%u", w, cls, nb, wp);
            CHECK(cls == 3 ? n_kt > kStreamTabM : n_kt <= (cls == 1 ? kStreamTabS : kStreamTabM), "table class w=%u cls=%d n_kt=%d", w, cls, n_kt);
            CHECK(resize_stream_wants_band(w) == resize_wavestream_applies(w), "band flag w=%u", w);
        }
        for (uint32_t h : {129u, 270u, 1080u, 1088u}) {
            const size_t fs = (size_t)w * h;
            const bool e = resize_stream_eligible(aligned, w, h, (fs + 15) & ~size_t(15), 16 * ((fs + 15) & ~size_t(15)));
            if (e) CHECK((cls == 1 || resize_wavestream_applies(w)) && w >= 64, "eligible implies a kernel w=%u", w);
            if ((cls == 1 || resize_wavestream_applies(w)) && w >= 64 && ((uint64_t)w * h) % 16 == 0) CHECK(e, "every width of the chunk or per-wave form streams w=%u", w);
            if (cls != 1 && !resize_wavestream_applies(w)) CHECK(!e, "widths beyond the per-wave buffers (pitch > 2368) do not stream w=%u", w);
            if (((uint64_t)w * h) % 16 != 0) CHECK(!e, "frames that do not end on a 16-byte boundary must not stream w=%u h=%u", w, h);
            CHECK(!resize_stream_eligible(aligned + 4, w, h, fs, 16 * fs), "misaligned base must not stream w=%u", w);
        }
        // ---- K-split kernel
        if (w % 16 == 0 && w >= 1024 && w <= 4096) {
            uint32_t kp = 0;
            const uint32_t knb = ksplit_geometry(w, &kp);
            CHECK(kp % 16 == 0 && (kp / 16) % 2 == 1 && kp >= w && kp <= w + 16, "k-split pitch %u -> %u", w, kp);
            CHECK(knb >= 1 && knb <= 4 && ((16 * knb * kp + 1023) & ~1023u) + 128 <= (uint32_t)kKsplitBuf, "k-split chunk w=%u nb=%u", w, knb);
            CHECK(n_kt <= 64, "k-split tiles per wave w=%u", w);
            CHECK(resize_ksplit_eligible(aligned, w, 1080, (size_t)w * 1080, (size_t)w * 1080 * 16), "k-split eligible w=%u", w);
        } else {
            CHECK(!resize_ksplit_eligible(aligned, w, 1080, (size_t)w * 1080, (size_t)w * 1080 * 16), "k-split must refuse w=%u", w);
        }
        // ---- cropped stream kernel: every crop box of an eligible pitch must fit with at least one block
        int ccls = 0;
        if (resize_cropped_stream_class(w, &ccls)) {
            CHECK(ccls == 1 || ccls == 2, "cropped class pitch=%u", w);
            for (uint32_t cw : {1u, 17u, w / 3 + 1, w - 1, w}) {
                for (uint32_t x0 : {0u, 1u, 5u}) {
                    if (cw == 0 || x0 + cw > w) continue;
                    uint32_t cp = 0;
                    const uint32_t cnb = resize_cropped_stream_blocks(cw, x0, w, ccls, &cp);
                    const bool linear = x0 == 0 && cw == w && w % 16 == 0 && w % 256 != 0;
                    CHECK(cp % 16 == 0 && (linear ? cp == w : (cp >= cw + 3 && (cp / 16) % 2 == 1)), "crop pitch pitch=%u cw=%u x0=%u -> %u", w, cw, x0, cp);
                    CHECK(cnb >= 1 && ((16 * cnb * cp + 1023) & ~1023u) + 128 <= (uint32_t)(ccls == 1 ? kStreamBufS : kStreamBufM),
                          "crop chunk pitch=%u cw=%u nb=%u cp=%u", w, cw, cnb, cp);
                }
            }
        }
    }
    // the sizes DESIGN.md names
    struct { uint32_t w; int cls; uint32_t nb; } want[] = {{480, 1, 4}, {426, 1, 4}, {854, 2, 4}, {960, 2, 4}, {640, 2, 4}, {768, 2, 4}, {1024, 2, 3},
                                                            {1280, 3, 3}, {1440, 3, 2}, {1920, 3, 2}, {1366, 3, 2}};
    for (auto &q : want) {
        uint32_t nb = 0;
        const int cls = stream_class(q.w, &nb);
        CHECK(cls == q.cls && nb == q.nb, "w=%u: class %d nb %u, expected %d %u", q.w, cls, nb, q.cls, q.nb);
    }
    for (uint32_t w : {480u, 854u, 640u, 768u, 1024u, 1280u, 1920u, 720u, 1440u, 240u, 160u, 128u, 1536u, 1792u})  // 1536 / 1792: per-wave block streams (round 3)
        CHECK(resize_stream_eligible(aligned, w, 1080, (size_t)w * 1080, (size_t)w * 1080 * 16), "%u wide should stream by default", w);
    for (uint32_t w : {2048u, 2560u, 3840u, 48u, 63u})
        CHECK(!resize_stream_eligible(aligned, w, 1080, (size_t)w * 1080, (size_t)w * 1080 * 16), "%u wide should not stream by default", w);
    // short frames (at most 128 rows: the fused kernel's range): the wide ones stream (round 5, measured)
    struct { uint32_t w, h; bool stream; } shorts[] = {{64, 64, false}, {128, 128, false}, {160, 90, false}, {128, 96, false}, {192, 80, false}, {512, 64, false},
                                                       {160, 120, false}, {192, 108, false}, {208, 112, false}, {208, 117, true}, {192, 128, true}, {200, 112, true}, {224, 126, true},
                                                       {256, 128, true}, {320, 96, true}, {480, 128, true},
                                                       {640, 120, true}, {854, 128, true}, {1920, 128, true}, {1920, 64, true}, {1920, 48, false}, {1920, 129, false}};
    for (auto &q : shorts) CHECK(resize_short_prefers_stream(q.w, q.h) == q.stream, "short frame %u x %u: stream %d", q.w, q.h, (int)q.stream);
    struct { uint32_t w, h; bool tiled; } talls[] = {{64, 160, true}, {64, 256, true}, {128, 256, true}, {176, 144, true}, {160, 200, true}, {16, 200, true}, {176, 208, false},
                                                     {192, 144, false}, {240, 160, false}, {64, 257, false}, {64, 128, false}, {100, 200, true}, {90, 250, true}, {144, 256, false}};
    for (auto &q : talls) CHECK(resize_tall_prefers_tiled(q.w, q.h) == q.tiled, "tall frame %u x %u: tiled %d", q.w, q.h, (int)q.tiled);
    // the per-wave block streams: every M-class width (from 462 columns) whose pitch is at most 1920, with as many waves as block buffers fit; the (whole-KB) block
    // fits the wave's buffer, the workgroup fits the CU's LDS, and the width's band table fits the table array of that wave count
    for (uint32_t w = 1; w <= 4200; w++) {
        uint32_t nb = 0;
        const int cls = stream_class(w, &nb);
        const int nw = resize_wavestream_waves(w);
        CHECK(resize_wavestream_applies(w) == (nw != 0), "applies <-> waves w=%u", w);
        if (nw) {
            CHECK(cls != 1 && w >= 256 && stream_pitch(w) <= 2368, "wave-stream width w=%u cls=%d nb=%u", w, cls, nb);
            CHECK(nw == 3 || nw == 4 || nw == 5 || nw == 6 || nw == 8, "wave count w=%u nw=%d", w, nw);
            CHECK((nw == 3) == (stream_pitch(w) > 1920) && (nw != 3 || w % 16 != 0), "three waves beyond the four-wave buffers, widths the K-split form cannot take w=%u nw=%d", w, nw);
            const int buf = nw == 3 ? kWaveStreamBuf3 : nw == 4 ? kWaveStreamBuf : nw == 5 ? kWaveStreamBuf5 : nw == 6 ? kWaveStreamBuf6 : kWaveStreamBuf8;
            const int tab = nw <= 4 ? kWaveStreamTabBytes : nw == 5 ? kWaveStreamTabMid : kWaveStreamTabSmall;
            CHECK(((16 * stream_pitch(w) + 1023) & ~1023u) + 128 <= (uint32_t)buf, "wave-stream block fits w=%u nw=%d", w, nw);
            CHECK(nw * buf + tab + 2 * (nw - 1) * 1024 <= kLdsPerCu, "wave-stream workgroup fits the LDS w=%u nw=%d", w, nw);
            MfmaAxisTable t;
            build_mfma_axis_table(w, kMfmaLayoutHorizontalBand, t);
            CHECK(t.ok && 16 * t.band_stride + 128 <= tab, "band table fits w=%u nw=%d: %d bytes of %d", w, nw, 16 * t.band_stride + 128, tab);
        } else if (w >= 256 && stream_pitch(w) <= 2368) {
            CHECK(cls == 1 || (w % 16 == 0 && w > 1920), "every width beyond the S class with a pitch up to 2368 takes the per-wave streams or the K-split form w=%u cls=%d", w, cls);
        }
    }
    for (uint32_t w : {528u, 640u, 854u, 1024u, 1280u, 1360u, 1366u, 1440u, 1536u, 1600u, 1680u, 1792u, 1904u, 1920u}) CHECK(resize_wavestream_applies(w), "%u wide takes the per-wave streams", w);
    for (uint32_t w : {480u, 320u, 1936u, 2048u, 2353u, 2560u, 3840u}) CHECK(!resize_wavestream_applies(w), "%u wide must not take the per-wave streams", w);
    CHECK(resize_wavestream_waves(640) == 8 && resize_wavestream_waves(1152) == 6 && resize_wavestream_waves(1366) == 5 && resize_wavestream_waves(1920) == 4 &&
          resize_wavestream_waves(1950) == 3 && resize_wavestream_waves(2340) == 3 && resize_wavestream_waves(2000) == 0 && resize_wavestream_waves(1915) == 3, "documented wave counts");
    // full-width crop boxes: the ROWCROP stream kernels everywhere but 2048 columns (measured)
    for (uint32_t w : {64u, 426u, 640u, 768u, 854u, 1024u, 1280u, 1366u, 1536u, 1600u, 1792u, 1920u, 2560u, 3840u, 4096u}) CHECK(resize_rowcrop_streams(w), "%u wide: row-cropped stream kernels", w);
    CHECK(!resize_rowcrop_streams(2048), "2048 wide: general cropped kernels");
    // crop boxes that share their column range, through the per-wave kernel: the LDS pitch holds the box and the up to 3 bytes in front of a
    // row that starts off a dword, is an odd multiple of 16, and the (whole-KB) block fits the buffer of the chosen wave count
    for (uint32_t fw : {640u, 854u, 1280u, 1366u, 1920u, 1921u, 2048u, 3840u})
        for (uint32_t x0 : {0u, 1u, 3u, 4u, 16u, 240u, 241u})
            for (uint32_t bw = 500; x0 + bw <= fw; bw += 37) {
                int mode = -1;
                const uint32_t wp = box_stream_pitch(fw, x0, bw, &mode);
                const bool whole = x0 == 0 && bw == fw, shifted = fw % 4 != 0 || x0 % 4 != 0;
                if (!whole) {
                    CHECK(mode == (shifted ? 2 : 1) && wp % 16 == 0 && (wp / 16) % 2 == 1 && wp >= bw + (shifted ? 3u : 0u) && wp < bw + 48,
                          "box pitch fw=%u x0=%u bw=%u -> %u mode %d", fw, x0, bw, wp, mode);
                }
                const int nw = resize_wavestream_waves_box(fw, x0, bw);
                if (nw && !whole) {
                    const int buf = nw == 3 ? kWaveStreamBuf3 : nw == 4 ? kWaveStreamBuf : nw == 5 ? kWaveStreamBuf5 : nw == 6 ? kWaveStreamBuf6 : kWaveStreamBuf8;
                    CHECK(bw >= 513 && ((16 * wp + 1023) & ~1023u) + 128 <= (uint32_t)buf, "box block fits fw=%u x0=%u bw=%u nw=%d", fw, x0, bw, nw);
                }
                if (!whole && bw >= 513 && wp <= 2368) CHECK(nw != 0, "boxes up to pitch 2368 take the per-wave kernel fw=%u x0=%u bw=%u", fw, x0, bw);
            }
    uint32_t kp = 0;
    CHECK(ksplit_geometry(3840, &kp) == 1 && kp == 3856, "4K: one 16-row block per chunk at pitch 3856");
    CHECK(ksplit_geometry(2048, &kp) == 2 && kp == 2064, "2048 wide: two blocks per chunk");
    if (fails == 0) std::printf("resize dispatch ok\n");
    return fails ? 1 : 0;
}
