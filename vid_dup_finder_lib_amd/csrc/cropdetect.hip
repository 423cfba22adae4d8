// Letterbox crop detection on the device (the step right before the hot path).
//
// Replaces vid_dup_finder_common/src/video_frames_gray.rs:38-128 (letterbox_crop, LetterboxColour::AnyColour(16))
// and :201-210 (cropdetect_letterbox: frames 0 and 8 of the 16 a clip contributes, crops united by per-edge
// minimum, crop.rs:53-68), as driven by vid_dup_finder_lib/src/video_hashing/video_hash_builder.rs:188-212.
//
// One workgroup per probed frame; wave 0/1/2/3 walks in from the left/right/top/bottom edge and stops at the
// first strip that is not letterbox, exactly like the reference's take_while.  A strip is letterbox when more
// than 90 % of its pixels are within +-tol of the strip's mode (ties: the LAST maximum, Iterator::max_by_key).
// Integer histogram per strip in LDS, one f64 compare: bit-exact with the oracle.
#include <algorithm>
#include <cstdlib>

#include "vdf_internal.h"

namespace vdf {

// Wave-wide reductions for all 64 lanes active (the walkers' code is wave-uniform).  DPP rotations inside each row of 16 lanes, then the
// four row results through readlane: ~10 instructions.  The __shfl_xor butterflies this replaces were twelve LDS round trips per strip
// (ds_bpermute) - most of a strip's wall time once its loads were batched.
template <int ROR> __device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + ROR, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, row_ror<8>(v));
    v = max(v, row_ror<4>(v));
    v = max(v, row_ror<2>(v));
    v = max(v, row_ror<1>(v));
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = min(v, row_ror<8>(v));
    v = min(v, row_ror<4>(v));
    v = min(v, row_ror<2>(v));
    v = min(v, row_ror<1>(v));
    return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

// Bytes of a dword as two packed 16-bit pairs (even bytes, odd bytes): running minima / maxima over pixels then cost one packed
// instruction per pair (v_pk_min_u16 / v_pk_max_u16).
typedef unsigned short vdf_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void minmax_bytes(uint32_t d, vdf_u16x2 &mn, vdf_u16x2 &mx)
{
    const vdf_u16x2 e = __builtin_bit_cast(vdf_u16x2, d & 0x00FF00FFu), o = __builtin_bit_cast(vdf_u16x2, (d >> 8) & 0x00FF00FFu);
    mn = __builtin_elementwise_min(__builtin_elementwise_min(mn, e), o);
    mx = __builtin_elementwise_max(__builtin_elementwise_max(mx, e), o);
}

// The reference's test is  count as f64 / len as f64 > 0.9  (video_frames_gray.rs:65,96-99).  For integers below 2^32 that is exactly
// 10 count > 9 len: a quotient other than 9/10 itself is at least 1 / (10 len) > 2^-36 away from 0.9, far more than the 2^-54 by which
// the double 0.9 and the rounding of the division can move it; and 9/10 exactly divides to the double 0.9, which is not greater than
// itself.  (The f64 division was a third of a strip's dependent chain.)
__device__ __forceinline__ bool more_than_nine_tenths(uint32_t count, uint32_t len) { return 10ull * count > 9ull * len; }

// hist: this wave's 256-bin LDS histogram.  All 64 lanes must call.
__device__ __forceinline__ bool strip_is_letterbox(const uint8_t *__restrict__ p, size_t step, uint32_t len,
                                                   uint32_t tol, uint32_t *hist)
{
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; k++) hist[lane + 64 * k] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Letterbox bars are runs of one value: 64 lanes adding to the same bin serialise in the LDS (measured: 3.5 ms to
    // probe 1000 1080p clips).  When every active lane holds the same value one lane adds the total; otherwise a lane
    // whose pixels are equal adds them at once.  Same histogram, fewer atomics.
    // The loads of a strip are issued in batches BEFORE the first of them is consumed: one dword per lane consumed at once made a
    // 3840-pixel row 15 dependent round trips to HBM (8 us per row; 2.1 ms to probe 250 letterboxed 4K clips).
    if (step == 1) {  // a row: sixteen pixels per (possibly unaligned) 16-byte load, four loads in flight
        struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };
        const uint32_t n16 = len >> 4;
        for (uint32_t i0 = 0; i0 < n16; i0 += 256) {
            U4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = i0 + 64u * k + lane;
                v[k] = i < n16 ? *reinterpret_cast<const U4 *>(p + 16 * (size_t)i) : U4{0, 0, 0, 0};
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool active = i0 + 64u * k + lane < n16;
                const uint64_t act = __builtin_amdgcn_ballot_w64(active);
                if (act == 0) break;  // wave-uniform: lane 0 is active whenever any lane is
                const uint32_t first = __builtin_amdgcn_readfirstlane(v[k].x);
                const bool same16 = v[k].x == (v[k].x & 255u) * 0x01010101u && v[k].y == v[k].x && v[k].z == v[k].x && v[k].w == v[k].x;
                if (__builtin_amdgcn_ballot_w64(active && same16 && v[k].x == first) == act) {
                    if (lane == 0) atomicAdd(&hist[first & 255u], 16u * (uint32_t)__builtin_popcountll(act));
                } else if (active) {
                    if (same16) {
                        atomicAdd(&hist[v[k].x & 255u], 16u);
                    } else {
                        const uint32_t d[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (d[j] == (d[j] & 255u) * 0x01010101u) {
                                atomicAdd(&hist[d[j] & 255u], 4u);
                            } else {
                                atomicAdd(&hist[d[j] & 255u], 1u);
                                atomicAdd(&hist[(d[j] >> 8) & 255u], 1u);
                                atomicAdd(&hist[(d[j] >> 16) & 255u], 1u);
                                atomicAdd(&hist[d[j] >> 24], 1u);
                            }
                        }
                    }
                }
            }
        }
        for (uint32_t i = 16 * n16 + lane; i < len; i += 64) atomicAdd(&hist[p[i]], 1u);
    } else {  // a column: one pixel per lane and load, eight loads in flight
        for (uint32_t i0 = 0; i0 < len; i0 += 512) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t i = i0 + 64u * k + lane;
                v[k] = i < len ? p[(size_t)i * step] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool active = i0 + 64u * k + lane < len;
                const uint64_t act = __builtin_amdgcn_ballot_w64(active);
                if (act == 0) break;
                const uint32_t first = __builtin_amdgcn_readfirstlane(v[k]);
                if (__builtin_amdgcn_ballot_w64(active && v[k] == first) == act) {
                    if (lane == 0) atomicAdd(&hist[first], (uint32_t)__builtin_popcountll(act));
                } else if (active) {
                    atomicAdd(&hist[v[k]], 1u);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t h[4], key = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        h[k] = hist[lane + 64 * k];
        key = max(key, (h[k] << 8) | (lane + 64 * k));  // max count, ties -> the larger value (the LAST maximum)
    }
    const uint32_t mode = wave_max_u32(key) & 255u;
    const uint32_t lo = mode > tol ? mode - tol : 0u, hi = min(mode + tol, 255u);
    uint32_t count = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t v = lane + 64 * k;
        if (v >= lo && v <= hi) count += h[k];
    }
    count = wave_sum_u32(count);
    __builtin_amdgcn_wave_barrier();
    return more_than_nine_tenths(count, len);
}

// Four row strips per pass (top / bottom walkers).  A row strip's wall time is not its loads any more but the dependent chain behind them
// (clear, LDS atomics, read back, two wave reductions, ~1.3 us) - and a bar is a hundred rows deep.  Four rows go through that chain
// together: sixteen lanes own a row's 256 bins (DPP reductions stay inside the 16-lane row), one ballot gives the four verdicts, judged in
// walking order (the reference's take_while, evaluated speculatively).  hist4: 4 x 256 words.  row(k) = address of strip k of the batch.
// Returns how many leading strips of the batch are letterbox (0 .. 4).  All 64 lanes must call.
template <class RowAt>
__device__ __forceinline__ uint32_t row_strips4(RowAt row, uint32_t len, uint32_t tol, uint32_t *hist4)
{
    const uint32_t lane = threadIdx.x & 63, n16 = len >> 4;
    struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };
    // A strip whose pixels all lie within `tol` of each other (max - min <= tol) is letterbox whatever its mode is: the mode is one of its
    // values, so every pixel is within tol of it and the count is the whole strip.  That is what a bar is - one value, or a few neighbouring
    // ones once a lossy codec has been over it - and it needs no histogram: four such rows are accepted from their loads alone, with packed
    // min / max per lane and two wave reductions per row.  (The histogram of a NOISY bar is the worst case the LDS has: a handful of bins,
    // every atomic a 16-way conflict - 1000 1080p clips with 129-row bars of 16 .. 19 took 1.24 ms to probe against 0.27 ms for bars of
    // exactly 16.)  Anything else - the batch where the picture begins - takes the general path below and reads its rows once more (from L2).
    if (n16 != 0) {
        vdf_u16x2 mn[4], mx[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { mn[r] = vdf_u16x2{0xFFFFu, 0xFFFFu}; mx[r] = vdf_u16x2{0u, 0u}; }
        for (uint32_t i0 = 0; i0 < n16; i0 += 128) {
            U4 v[4][2];  // all eight loads of the pass in flight together: one round trip to HBM
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    // (UNCONDITIONAL loads at a clamped index - a clamped lane looks at the row's last 16 pixels again, harmless: a
                    // `valid ? load : 0` here became a branch around every load with a full vmcnt(0) wait behind it, eight serial round trips)
                    const uint32_t i = min(i0 + 64u * k + lane, n16 - 1);
                    v[r][k] = *reinterpret_cast<const U4 *>(row(r) + 16 * (size_t)i);
                }
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    minmax_bytes(v[r][k].x, mn[r], mx[r]);
                    minmax_bytes(v[r][k].y, mn[r], mx[r]);
                    minmax_bytes(v[r][k].z, mn[r], mx[r]);
                    minmax_bytes(v[r][k].w, mn[r], mx[r]);
                }
        }
        bool narrow = true;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t lo = min((uint32_t)mn[r].x, (uint32_t)mn[r].y), hi = max((uint32_t)mx[r].x, (uint32_t)mx[r].y);
            for (uint32_t i = 16 * n16 + lane; i < len; i += 64) {  // the last len % 16 pixels of the row
                const uint32_t p = row(r)[i];
                lo = min(lo, p);
                hi = max(hi, p);
            }
            narrow = narrow & (wave_max_u32(hi) - wave_min_u32(lo) <= tol);
        }
        if (narrow) return 4;
    }
    {   // (the clear's addresses are made from a lane number the compiler cannot see through: hoisted out of the caller's walk loop they
        // stayed live across it and were the one value pass 1's 64 registers had no room for)
        uint32_t l2 = lane;
        asm volatile("" : "+v"(l2));
#pragma unroll
        for (int k = 0; k < 16; k++) hist4[l2 + 64 * k] = 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i0 = 0; i0 < n16; i0 += 128) {  // two 16-byte loads per lane and row in flight, four rows
        U4 v[4][2];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t i = i0 + 64u * k + lane;
                v[r][k] = i < n16 ? *reinterpret_cast<const U4 *>(row(r) + 16 * (size_t)i) : U4{0, 0, 0, 0};
            }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t *hist = hist4 + 256 * r;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const bool active = i0 + 64u * k + lane < n16;
                const uint64_t act = __builtin_amdgcn_ballot_w64(active);
                if (act == 0) break;  // wave-uniform
                const U4 &q = v[r][k];
                const uint32_t first = __builtin_amdgcn_readfirstlane(q.x);
                const bool same16 = q.x == (q.x & 255u) * 0x01010101u && q.y == q.x && q.z == q.x && q.w == q.x;
                if (__builtin_amdgcn_ballot_w64(active && same16 && q.x == first) == act) {
                    if (lane == 0) atomicAdd(&hist[first & 255u], 16u * (uint32_t)__builtin_popcountll(act));
                } else if (active) {
                    if (same16) {
                        atomicAdd(&hist[q.x & 255u], 16u);
                    } else {
                        const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (d[j] == (d[j] & 255u) * 0x01010101u) {
                                atomicAdd(&hist[d[j] & 255u], 4u);
                            } else {
                                atomicAdd(&hist[d[j] & 255u], 1u);
                                atomicAdd(&hist[(d[j] >> 8) & 255u], 1u);
                                atomicAdd(&hist[(d[j] >> 16) & 255u], 1u);
                                atomicAdd(&hist[d[j] >> 24], 1u);
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
        for (uint32_t i = 16 * n16 + lane; i < len; i += 64) atomicAdd(&hist4[256 * r + row(r)[i]], 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane = 16 r + q: bins 16 q .. 16 q + 15 of row r
    const uint32_t r = lane >> 4, q = lane & 15;
    const uint32_t *hr = hist4 + 256 * r + 16 * q;
    uint32_t key = 0;
#pragma unroll
    for (uint32_t b = 0; b < 16; b++) key = max(key, (hr[b] << 8) | (16u * q + b));  // max count, ties -> the larger value
    key = max(key, row_ror<8>(key));
    key = max(key, row_ror<4>(key));
    key = max(key, row_ror<2>(key));
    key = max(key, row_ror<1>(key));
    const uint32_t mode = key & 255u, lo = mode > tol ? mode - tol : 0u, hi = min(mode + tol, 255u);
    uint32_t count = 0;
#pragma unroll
    for (uint32_t b = 0; b < 16; b++) {
        const uint32_t val = 16u * q + b;
        if (val >= lo && val <= hi) count += hr[b];
    }
    count += row_ror<8>(count);
    count += row_ror<4>(count);
    count += row_ror<2>(count);
    count += row_ror<1>(count);
    const uint64_t okm = __builtin_amdgcn_ballot_w64(more_than_nine_tenths(count, len));
    __builtin_amdgcn_wave_barrier();
    uint32_t got = 0;
    while (got < 4 && ((okm >> (16 * got)) & 1ull)) got++;
    return got;
}

// NC (8 or 16) adjacent column strips at once (pillarboxed clips: 4 : 3 content in a 16 : 9 frame walks 240 columns in from each side of
// a 1080p frame).  One column strip is H single bytes, each in its own cache line: strip by strip the walk fetched every line of the
// frame's side NC times over and was bound by the address coalescer (18 ms to probe 1000 pillarboxed 1080p clips).  Here a lane reads
// the NC bytes of its row that hold columns x0 .. x0 + NC - 1 and the wave keeps NC histograms (u16 counters, two columns per LDS word:
// H < 65536); the strips are then judged in walking order - exactly the reference's take_while, evaluated speculatively.
// from_right: strip k of the batch is column x0 + NC - 1 - k.  Returns how many leading strips of the batch are letterbox (0 .. NC).
// histn: NC / 2 x kHistPitch words of this wave (the callers give NC / 2 x 256 + 64).  All 64 lanes must call.
constexpr uint32_t kHistPitch = 257;
template <int NC>
__device__ __forceinline__ uint32_t column_strips(const uint8_t *__restrict__ f, uint32_t W, uint32_t H, uint32_t x0, bool from_right,
                                                  uint32_t tol, uint32_t *histn)
{
    constexpr uint32_t HALF = NC / 2, ND = NC / 4;  // columns per counter half, dwords per row
    const uint32_t lane = threadIdx.x & 63;
    struct __attribute__((packed, aligned(1))) UN { uint32_t d[ND]; };
#pragma unroll
    for (uint32_t k = 0; k < HALF * 4 + 1; k++) histn[lane + 64 * k] = 0u;  // HALF x kHistPitch words (+ slack inside the wave's array)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto byte_of = [](const UN &v, uint32_t c) {  // (64-bit shifts and selects, not an indexed load: a lane-dependent index would put v in scratch)
        uint64_t w = (uint64_t)v.d[0] | ((uint64_t)v.d[1] << 32);
#pragma unroll
        for (uint32_t j = 1; j < ND / 2; j++) {
            const uint64_t wj = (uint64_t)v.d[2 * j] | ((uint64_t)v.d[2 * j + 1] << 32);
            w = (c >> 3) == j ? wj : w;
        }
        return (uint32_t)(w >> (8 * (c & 7))) & 255u;
    };
    // A column's 256 bins are kHistPitch = 257 words apart: at 256, bin v of every column sat in ONE bank - the bars' single value made the
    // wave's atomics 16-way conflicts and the scan below (lanes = columns) 16-way on every read: 90 % of the kernel's LDS cycles were
    // conflict cycles (rocprofv3, round 5: SQ_LDS_BANK_CONFLICT 64.7 M of SQ_LDS_IDX_ACTIVE 71.4 M per launch on 1000 pillarboxed 1080p clips).
    auto slot = [&](uint32_t c, uint32_t value) { return &histn[(c % HALF) * kHistPitch + value]; };
    // rows per lane in flight: 64 / 32 / 8 dwords of loads.  The side walk is latency-bound at two waves per SIMD (62 % of the wave cycles
    // wait; 1.1 GB per launch is only 2.3 TB/s): a pass of a 1080-row frame is 17 rows per lane, and two at a time made it nine dependent
    // round trips to HBM.  The kernel's occupancy is set by its LDS, so the registers are there.
    constexpr int INFL = NC >= 16 ? 8 : 4;
    for (uint32_t i0 = 0; i0 < H; i0 += 64u * INFL) {
        UN v[INFL];
#pragma unroll
        for (int k = 0; k < INFL; k++) {
            const uint32_t row = i0 + 64u * k + lane;
            if (row < H) v[k] = *reinterpret_cast<const UN *>(f + (size_t)row * W + x0);
            else
#pragma unroll
                for (uint32_t j = 0; j < ND; j++) v[k].d[j] = 0u;
        }
#pragma unroll
        for (int k = 0; k < INFL; k++) {
            const bool active = i0 + 64u * k + lane < H;
            const uint64_t act = __builtin_amdgcn_ballot_w64(active);
            if (act == 0) break;  // wave-uniform
            UN first;
            bool same = true;
#pragma unroll
            for (uint32_t j = 0; j < ND; j++) {
                first.d[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[k].d[j]);
                same = same && v[k].d[j] == first.d[j];
            }
            if (__builtin_amdgcn_ballot_w64(active && same) == act) {
                // every active row holds the same NC bytes: lane c adds the row count to column c's bin
                if (lane < (uint32_t)NC) atomicAdd(slot(lane, byte_of(first, lane)), (uint32_t)__builtin_popcountll(act) << (16 * (lane / HALF)));
            } else if (active) {
#pragma unroll 4
                for (uint32_t c = 0; c < (uint32_t)NC; c++) atomicAdd(slot(c, byte_of(v[k], c)), 1u << (16 * (c / HALF)));
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane = NC q + c: share q of column c's histogram (256 / (64 / NC) bins)
    constexpr uint32_t SHARES = 64 / NC, BINS = 256 / SHARES;
    const uint32_t c = lane % NC, q = lane / NC, sh = 16 * (c / HALF);
    const uint32_t *hc = histn + (c % HALF) * kHistPitch + BINS * q;
    uint32_t key = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < BINS; b++) key = max(key, (((hc[b] >> sh) & 0xFFFFu) << 8) | (BINS * q + b));  // max count, ties -> the larger value
#pragma unroll
    for (uint32_t o = NC; o < 64; o <<= 1) key = max(key, (uint32_t)__shfl_xor((int)key, (int)o, 64));
    const uint32_t mode = key & 255u, lo = mode > tol ? mode - tol : 0u, hi = min(mode + tol, 255u);
    uint32_t count = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < BINS; b++) {
        const uint32_t val = BINS * q + b;
        if (val >= lo && val <= hi) count += (hc[b] >> sh) & 0xFFFFu;
    }
#pragma unroll
    for (uint32_t o = NC; o < 64; o <<= 1) count += (uint32_t)__shfl_xor((int)count, (int)o, 64);
    const bool pass = more_than_nine_tenths(count, H);
    constexpr uint32_t ALL = NC == 32 ? 0xFFFFFFFFu : (1u << (NC & 31)) - 1u;
    uint32_t ok = (uint32_t)__builtin_amdgcn_ballot_w64(pass && lane < (uint32_t)NC) & ALL;  // bit c = column x0 + c is letterbox
    __builtin_amdgcn_wave_barrier();
    if (from_right) ok = __builtin_bitreverse32(ok) >> (32 - NC);  // bit k = strip k
    return ok == ALL ? (uint32_t)NC : (uint32_t)__builtin_ctz(~ok);
}


// True iff, in every one of the WB columns x0 .. x0 + WB - 1, all rows lie within tol / 2 of the column's pixel in row 0 - then each column's
// pixels are within tol of each other, hence of its mode whatever that is, and all WB strips are letterbox (a bar: one value per column, or
// a few neighbouring ones after a lossy codec).  False decides nothing.  No LDS, no counting: packed 16-bit adds against row 0 and one running
// maximum.  What it is for is BYTES and LDS conflicts: the side walk is bound by HBM transactions, not by its round trips (rocprofv3,
// round 5: 43 M L2 misses of 64 B per launch on 1000 pillarboxed 1080p clips = 2.8 GB at 3.5 TB/s of scattered 64-byte reads, for 1.1 GB
// of columns: a 32-byte batch fetches a 64-byte sector, and the batch next to it fetches the same sector again later) - a probe at a
// 128- or 64-byte ALIGNED window reads whole lines / sectors once; and a noisy bar's histogram is all 16-way bank conflicts.
// All 64 lanes must call.
template <int WB>
__device__ __forceinline__ bool columns_narrow(const uint8_t *__restrict__ f, uint32_t W, uint32_t H, uint32_t x0, uint32_t tol)
{
    constexpr uint32_t ND = WB / 4;
    struct __attribute__((packed, aligned(1))) UW { uint32_t d[ND]; };
    const uint32_t lane = threadIdx.x & 63;
    constexpr int INFL = WB >= 128 ? 4 : 8;  // rows per lane in flight (128 / 128 / 64 / 32 / 16 dwords of loads)
    const uint32_t half = tol / 2;
    vdf_u16x2 ce[ND], co[ND];                // half - (row 0's pixel), per even / odd byte of each dword (scalars)
#pragma unroll
    for (uint32_t j = 0; j < ND; j++) ce[j] = co[j] = vdf_u16x2{0u, 0u};
    vdf_u16x2 worst = {0u, 0u};              // max over everything seen of (pixel - row 0's pixel + half) mod 2^16: <= tol iff inside the window
    auto look = [&](const UW &v) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t j = 0; j < ND; j++) {
            const uint32_t d = v.d[j];
            const vdf_u16x2 e = __builtin_bit_cast(vdf_u16x2, d & 0x00FF00FFu), o = __builtin_bit_cast(vdf_u16x2, (d >> 8) & 0x00FF00FFu);
            worst = __builtin_elementwise_max(__builtin_elementwise_max(worst, e + ce[j]), o + co[j]);
        }
    };
    auto outside = [&]() __attribute__((always_inline)) {  // wave-uniform
        const uint32_t wx = worst.x, wy = worst.y;
        return __builtin_amdgcn_ballot_w64((wx > wy ? wx : wy) > tol) != 0ull;
    };
    {   // The first 64 rows on their own, one load per lane: the window where the picture begins - every walk ends in three or four of them -
        // fails here for a quarter or an eighth of what a full first round reads (the side walk is bound by its HBM transactions).
        // (unconditional loads at a clamped row: a clamped lane looks at the last row again)
        const UW v = *reinterpret_cast<const UW *>(f + (size_t)min(lane, H - 1) * W + x0);
#pragma unroll
        for (uint32_t j = 0; j < ND; j++) {  // lane 0: row 0
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.d[j]);
            ce[j] = vdf_u16x2{(unsigned short)(half - (r0 & 255u)), (unsigned short)(half - ((r0 >> 16) & 255u))};
            co[j] = vdf_u16x2{(unsigned short)(half - ((r0 >> 8) & 255u)), (unsigned short)(half - (r0 >> 24))};
        }
        look(v);
        if (outside()) return false;
    }
    for (uint32_t i0 = 64; i0 < H; i0 += 64u * INFL) {
        UW v[INFL];
#pragma unroll
        for (int k = 0; k < INFL; k++) v[k] = *reinterpret_cast<const UW *>(f + (size_t)min(i0 + 64u * k + lane, H - 1) * W + x0);
#pragma unroll
        for (int k = 0; k < INFL; k++) look(v[k]);
        if (outside()) return false;
    }
    return true;
}

// Pass 1: one workgroup per probed frame.  Waves 2 / 3 walk in from the top / bottom; waves 0 / 1 judge only the FIRST column strip of their
// side.  Most frames have no side bars and are finished here (4 KB of LDS, the launch rate of 40 000 small workgroups matters at 64 x 64).
// A frame whose left or right first strip IS letterbox goes on the work list of pass 2 with its top / bottom result.
// work: 64 sub-lists (one shared counter took 40 000 same-address atomics 0.6 ms when every 64 x 64 clip of a batch had side bars): counts[64],
// then 64 x cap entries {frame index (clip * n_probe + probe), top, bottom, left | right << 1 first-strip flags}; frame i appends to sub-list
// i % 64, so cap = ceil(frames / 64) entries always suffice.
constexpr uint32_t kWorkLists = 64;
__global__ __launch_bounds__(256, 8) void letterbox_kernel(const uint8_t *__restrict__ frames, uint32_t W, uint32_t H,
                                                        size_t frame_stride, size_t clip_stride, uint32_t n_probe,
                                                        uint32_t tol, uint32_t *__restrict__ crops, uint32_t *__restrict__ work)
{
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint32_t s_hist4[2][4 * 256];  // the row walkers' four-strip batches
    __shared__ uint32_t s_edge[4];
    __shared__ uint32_t s_prog[4];  // strips each walker has confirmed so far
    const size_t clip = blockIdx.x / n_probe;
    const uint32_t probe = blockIdx.x % n_probe;  // frame 8 * probe
    const uint8_t *f = frames + clip * clip_stride + (size_t)(8 * probe) * frame_stride;
    // (readfirstlane: the wave index is uniform, and saying so keeps every per-wave LDS address - histograms, s_prog, s_edge - in SGPRs; as a
    // per-lane value they cost the 64-register budget of eight waves per SIMD four VGPRs it did not have: 4 spills to scratch until round 5)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *hist = s_hist[wave];
    if (threadIdx.x < 4) s_prog[threadIdx.x] = 0;
    __syncthreads();
    // A frame whose opposite walkers meet (a fade-in: uniformly black, every strip of every edge is letterbox) means "no crop" whatever the
    // other edges find (video_frames_gray.rs:119-127), and each walker alone would go through the whole frame first - 1.5 ms for ONE
    // such 1080p frame in a batch whose other thousand clips take 0.1 ms.  The walkers publish their progress; once top + bottom
    // reach H the result is fixed and they stop.  (The counts only grow, so a partial sum that reaches the extent implies the final one does.)
    auto publish = [&](uint32_t n) {
        if ((threadIdx.x & 63) == 0) __atomic_store_n(&s_prog[wave], n, __ATOMIC_RELAXED);
    };
    auto converged = [&]() {
        const uint32_t t = __atomic_load_n(&s_prog[2], __ATOMIC_RELAXED), b = __atomic_load_n(&s_prog[3], __ATOMIC_RELAXED);
        return __builtin_amdgcn_readfirstlane((int)((uint64_t)t + b >= H)) != 0;
    };
    uint32_t n = 0;
    if (wave == 0) {
        n = strip_is_letterbox(f, W, H, tol, hist) ? 1u : 0u;
    } else if (wave == 1) {
        n = strip_is_letterbox(f + (W - 1), W, H, tol, hist) ? 1u : 0u;
    } else {
        // top (wave 2) / bottom (wave 3): strip k is row k / H - 1 - k.  The first strip alone (most frames have no bar and stop here), then
        // four at a time while a whole batch is inside the frame
        const bool bottom = wave == 3;
        auto strip = [&](uint32_t k) { return f + (size_t)(bottom ? H - 1 - k : k) * W; };
        if (strip_is_letterbox(strip(0), 1, W, tol, hist)) {
            n = 1;
            publish(n);
            bool walking = true;
            while (walking && n + 4 <= H && !converged()) {
                const uint32_t n0 = n;
                const uint32_t got = row_strips4([&](uint32_t k) { return strip(n0 + k); }, W, tol, s_hist4[wave - 2]);
                n += got;
                publish(n);
                walking = got == 4;
            }
            if (walking)
                while (n < H && !converged() && strip_is_letterbox(strip(n), 1, W, tol, hist)) publish(++n);
        }
    }
    if ((threadIdx.x & 63) == 0) s_edge[wave] = n;
    __syncthreads();
    const uint32_t l = s_edge[0], r = s_edge[1], t = s_edge[2], b = s_edge[3];
    // side bars to walk (and the frame is not already decided: top + bottom met, or a frame of one or two columns that the first strips cover)
    const bool deferred = (l | r) != 0 && (uint64_t)t + b < H && (uint64_t)l + r < W;
    if (deferred) {
        if (threadIdx.x == 0) {
            const uint32_t k = blockIdx.x % kWorkLists, cap = (gridDim.x + kWorkLists - 1) / kWorkLists;
            const uint32_t at = atomicAdd(&work[k], 1u);
            uint32_t *e = work + kWorkLists + 4 * ((size_t)k * cap + at);
            e[0] = blockIdx.x; e[1] = t; e[2] = b; e[3] = l | (r << 1);
        }
    } else if (threadIdx.x < 4) {
        // video_frames_gray.rs:119-127: converging edges (e.g. a uniform frame) mean "no crop"
        const bool ok = (long long)W - l - r >= 1 && (long long)H - t - b >= 1;
        atomicMin(&crops[clip * 4 + threadIdx.x], ok ? s_edge[threadIdx.x] : 0u);  // union = per-edge minimum
    }
}

// Pass 2: the frames with side bars.  Persistent workgroups of two waves (left / right walker) take entries off the work list and walk
// kColumnBatch column strips per pass; the walkers of a frame stop when left + right reach W.  Then the frame's four edges are final.
template <int kColumnBatch>
__global__ __launch_bounds__(128) void letterbox_sides_kernel(const uint8_t *__restrict__ frames, uint32_t W, uint32_t H,
                                                              size_t frame_stride, size_t clip_stride, uint32_t n_probe,
                                                              uint32_t tol, uint32_t *__restrict__ crops, const uint32_t *__restrict__ work,
                                                              uint32_t n_frames)
{
    __shared__ uint32_t s_histn[2][kColumnBatch / 2 * 256 + 64];  // batches of strips (pitch kHistPitch), or one strip in the first KB
    __shared__ uint32_t s_edge[2];
    __shared__ uint32_t s_prog[2];
    // workgroup b serves sub-list b % 64 from entry b / 64 in steps of gridDim.x / 64 (the grid is a multiple of 64)
    const uint32_t wave = threadIdx.x >> 6, list = blockIdx.x % kWorkLists, cap = (n_frames + kWorkLists - 1) / kWorkLists;
    const uint32_t n_work = work[list];
    const bool right = wave == 1;
    uint32_t *hist = s_histn[wave];
    for (uint32_t at = blockIdx.x / kWorkLists; at < n_work; at += gridDim.x / kWorkLists) {
        const uint32_t *e = work + kWorkLists + 4 * ((size_t)list * cap + at);
        const uint32_t frame = e[0], t = e[1], b = e[2], first = (e[3] >> wave) & 1u;
        const size_t clip = frame / n_probe;
        const uint8_t *f = frames + clip * clip_stride + (size_t)(8 * (frame % n_probe)) * frame_stride;
        if (threadIdx.x < 2) s_prog[threadIdx.x] = (e[3] >> threadIdx.x) & 1u;
        __syncthreads();
        auto publish = [&](uint32_t n) {
            if ((threadIdx.x & 63) == 0) __atomic_store_n(&s_prog[wave], n, __ATOMIC_RELAXED);
        };
        auto converged = [&]() {
            const uint32_t l = __atomic_load_n(&s_prog[0], __ATOMIC_RELAXED), r = __atomic_load_n(&s_prog[1], __ATOMIC_RELAXED);
            return __builtin_amdgcn_readfirstlane((int)((uint64_t)l + r >= W)) != 0;
        };
        uint32_t n = first;
        if (first) {
            bool walking = true;
            // Aligned probes for bars (columns_narrow), where rows and frame are line-aligned: a window of 128 bytes, after its first
            // failure 64, after that the counted 32-column batches for good.  The window that holds strip n starts at or before it: the strips
            // in front of n inside it were accepted already, and if they are not constant columns (a noisy bar) the probe just fails.
            const uint32_t align = (uint32_t)(reinterpret_cast<uintptr_t>(f) | W);  // rows start where the frame does, W bytes apart
            uint32_t probe = kColumnBatch != 32 ? 0u : (align & 127u) == 0 ? 128u : (align & 63u) == 0 ? 64u : 0u;
            while (walking && n + kColumnBatch <= W && H < 65536u && !converged()) {
                if (probe) {
                    const uint32_t edge = right ? W - n : n;                                              // first column not yet accepted (left) / one past it (right)
                    const uint32_t a = right ? (edge + probe - 1) / probe * probe - probe : edge / probe * probe;  // the aligned window that holds it
                    const bool clean = probe == 128u ? columns_narrow<128>(f, W, H, a, tol) : columns_narrow<64>(f, W, H, a, tol);
                    if (clean) {
                        n = right ? W - a : a + probe;
                        publish(n);
                        continue;
                    }
                    probe = probe == 128u ? 64u : 0u;
                    continue;
                }
                // a batch of narrow columns (a bar, clean or noisy) is accepted without a histogram; the batch where the picture begins fails
                // this at its first row group and is counted
                const uint32_t x0 = right ? W - n - kColumnBatch : n;
                const uint32_t got = columns_narrow<kColumnBatch>(f, W, H, x0, tol) ? (uint32_t)kColumnBatch
                                                                                   : column_strips<kColumnBatch>(f, W, H, x0, right, tol, s_histn[wave]);
                n += got;
                publish(n);
                walking = got == (uint32_t)kColumnBatch;
            }
            if (walking)
                while (n < W && !converged() && strip_is_letterbox(right ? f + (W - n - 1) : f + n, W, H, tol, hist)) publish(++n);
        }
        if ((threadIdx.x & 63) == 0) s_edge[wave] = n;
        __syncthreads();
        if (threadIdx.x < 4) {
            const uint32_t l = s_edge[0], r = s_edge[1];
            const uint32_t edge[4] = {l, r, t, b};
            const bool ok = (long long)W - l - r >= 1 && (long long)H - t - b >= 1;
            atomicMin(&crops[clip * 4 + threadIdx.x], ok ? edge[threadIdx.x] : 0u);
        }
        __syncthreads();  // s_edge / s_prog are rewritten for the next entry
    }
}

size_t letterbox_work_bytes(size_t n_clips, uint32_t frames_per_clip)
{
    const uint32_t nf = frames_per_clip < VDF_DCT_SIZE ? frames_per_clip : VDF_DCT_SIZE;
    const size_t frames = n_clips * ((nf + 7) / 8), cap = (frames + kWorkLists - 1) / kWorkLists;
    return (kWorkLists + 4 * kWorkLists * cap) * sizeof(uint32_t);
}

hipError_t launch_letterbox(const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w, uint32_t h,
                            size_t frame_stride, size_t clip_stride, uint32_t *crops, uint32_t *work, hipStream_t stream, int side_strips)
{
    if (n_clips == 0) return hipSuccess;
    const uint32_t nf = frames_per_clip < VDF_DCT_SIZE ? frames_per_clip : VDF_DCT_SIZE;  // the builder keeps 16 frames
    const uint32_t n_probe = (nf + 7) / 8;                                                // frames 0, 8
    hipError_t e = hipMemsetAsync(crops, 0xFF, n_clips * 4 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(work, 0, kWorkLists * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(letterbox_kernel, dim3((uint32_t)(n_clips * n_probe)), dim3(256), 0, stream, frames, w, h,
                       frame_stride, clip_stride, n_probe, 16u, crops, work);
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // Sixteen strips per pass (16 KB of LDS per workgroup, nine per CU); small frames, where a strip is a handful of loads and the
    // fixed work per pass (clearing and scanning the histograms) dominates, take eight (twice the workgroups per CU: 64 x 64 x 20 000 clips
    // with side bars 0.86 -> 0.45 ms).  An empty work list costs one pass of trivial workgroups.
    const uint32_t n_frames = (uint32_t)(n_clips * n_probe);
    auto grid_for = [&](uint32_t per_cu) {  // a multiple of 64 (the sub-lists), no more than the frames can fill
        const size_t want = std::min<size_t>(n_frames, (size_t)cus * per_cu);
        return (uint32_t)((want + kWorkLists - 1) / kWorkLists * kWorkLists);
    };
    if (h >= 512 && side_strips != 16)
        hipLaunchKernelGGL(letterbox_sides_kernel<32>, dim3(grid_for(4)), dim3(128), 0, stream, frames, w, h, frame_stride, clip_stride, n_probe, 16u, crops,
                           work, n_frames);
    else if (h >= 256)
        hipLaunchKernelGGL(letterbox_sides_kernel<16>, dim3(grid_for(9)), dim3(128), 0, stream, frames, w, h, frame_stride, clip_stride, n_probe, 16u, crops,
                           work, n_frames);
    else
        hipLaunchKernelGGL(letterbox_sides_kernel<8>, dim3(grid_for(16)), dim3(128), 0, stream, frames, w, h, frame_stride, clip_stride, n_probe, 16u, crops,
                           work, n_frames);
    return hipGetLastError();
}

}  // namespace vdf
