// VideoHash construction kernels for gfx950 (MI355X), wave64.
//
// Replaces, per clip (vid_dup_finder_lib/src/video_hashing/video_hash.rs:45-73):
//   crop_resize_buf           vid_dup_finder_common/src/resize_gray.rs:11-54 (fast_image_resize 5.1,
//                             Convolution(Lanczos3), u8 fixed point, horizontal then vertical pass)
//   Dct3d::from_images        vid_dup_finder_lib/src/video_hashing/dct_3d.rs:15-53 (pix - 128, f64 cube)
//   dct_3d                    vid_dup_finder_lib/src/video_hashing/raw_dct_ops.rs:107-142 (3 x DCT-II, size 16)
//   hash_bits + Lsb0 pack     dct_3d.rs:55-66, video_hash.rs:64-68 (low 10^3 corner, coef > 0.0)
//
// Two kernels: resize (one workgroup per frame, W x H -> 16 x 16 u8 into a small cube buffer) and
// dct_hash (one workgroup per clip: f64 DCT-II along x, y, t through LDS, pruned to the 10 outputs
// per axis that are consumed, sign test, __ballot pack: ballot of wave-word w IS hash word w).
#include <algorithm>
#include <cstdlib>

#include "resize_dispatch.h"
#include "resize_tables.h"
#include "vdf_internal.h"

namespace vdf {

typedef const __attribute__((address_space(4))) double *const_f64_ptr;

__device__ __forceinline__ uint8_t clip8(int32_t v, int precision)
{
    int32_t s = v >> precision;
    s = s < 0 ? 0 : s;
    s = s > 255 ? 255 : s;
    return (uint8_t)s;
}

// Generic (any W, H) resize of one frame per workgroup.  Correct for every size the tables describe;
// the 64 x 64 fast path lives in resize64_dct_hash_kernel (when enabled).
__global__ __launch_bounds__(256) void resize_generic_kernel(
    const uint8_t *__restrict__ frames, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
    ResizeAxisTable th, ResizeAxisTable tv, int need_h, int need_v, int32_t y_first, int32_t tmp_rows,
    uint8_t *__restrict__ small)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tmp[];  // [tmp_rows][16]
    const size_t clip = blockIdx.x >> 4;
    const uint32_t f = blockIdx.x & 15;
    const uint8_t *src = frames + clip * clip_stride + (size_t)f * frame_stride;
    uint8_t *dst = small + (clip * 16 + f) * 256;

    // horizontal pass into the temporary (only the rows the vertical pass reads)
    for (int32_t idx = threadIdx.x; idx < tmp_rows * 16; idx += 256) {
        const int32_t y = idx >> 4, o = idx & 15;
        const uint8_t *row = src + (size_t)(y + y_first) * w;
        uint8_t v;
        if (need_h) {
            int32_t ss = 1 << (th.precision - 1);
            const int16_t *k = th.w + (size_t)o * th.window;
            const int32_t s0 = th.start[o], n = th.size[o];
            for (int32_t t = 0; t < n; t++) ss += (int32_t)row[s0 + t] * (int32_t)k[t];
            v = clip8(ss, th.precision);
        } else {
            v = row[o];
        }
        s_tmp[idx] = v;
    }
    __syncthreads();
    {
        const int32_t oy = threadIdx.x >> 4, x = threadIdx.x & 15;
        uint8_t v;
        if (need_v) {
            int32_t ss = 1 << (tv.precision - 1);
            const int16_t *k = tv.w + (size_t)oy * tv.window;
            const int32_t s0 = tv.start[oy] - y_first, n = tv.size[oy];
            for (int32_t t = 0; t < n; t++) ss += (int32_t)s_tmp[(s0 + t) * 16 + x] * (int32_t)k[t];
            v = clip8(ss, tv.precision);
        } else {
            v = s_tmp[threadIdx.x];
        }
        dst[threadIdx.x] = v;
    }
}

// 16-point unnormalised DCT-II,  X[k] = sum_n v[n] cos(pi k (2n + 1) / 32),  outputs 0..9.
// Only the SIGN of a coefficient is consumed (dct_3d.rs:55-62), so the order of the floating-point operations decides the
// bit wherever the coefficient is mathematically zero - and that is not exotic: every temporal AC coefficient of a static
// clip, every AC coefficient of a black frame, every odd coefficient of a mirror-symmetric line.  rustdct 0.7's
// plan_dct2(16) is Type2And3Butterfly16: split-radix steps (16 -> 8 + 4 + 4, 8 -> 4 + 2 + 2, 4 -> 2 + 1 + 1) in which
// even outputs come from the sums v[n] + v[N-1-n] and odd outputs from the differences v[n] - v[N-1-n] rotated by
// e^{i pi (2n+1) / 2N}; a constant or symmetric line then yields exact +-0.0 there (bit 0).  This is the same operation
// sequence as the CPU checker under oracle/ (dct2_len16), with every product and sum rounded separately like Rust does
// (contraction off), so the coefficients - not just their signs - are bit-identical to the oracle's.
// The 15 constants (twiddles + sqrt(1/2)) are wave-uniform and live in SGPR pairs; unused outputs fall to dead-code elimination.
#pragma clang fp contract(off)
struct DctTw {
    double t16[4][2], t8[2][2], t4[2], h;  // (cos, sin) of pi (2 i + 1) / (2 N); h = sqrt(1/2)
};

__device__ __forceinline__ void dct2_len2(double &a, double &b, const DctTw &T)
{
    const double sum = a + b;
    b = (a - b) * T.h;
    a = sum;
}

__device__ __forceinline__ void dct2_len4(double (&b)[4], const DctTw &T)
{
    double i0 = b[3] + b[0], i1 = b[1] + b[2];
    const double lower = b[0] - b[3], upper = b[1] - b[2];
    const double cos_in = lower * T.t4[0] + upper * T.t4[1];
    const double sin_in = upper * T.t4[0] - lower * T.t4[1];
    dct2_len2(i0, i1, T);
    b[0] = i0; b[1] = cos_in; b[2] = i1; b[3] = -sin_in;
}

__device__ __forceinline__ void dct2_len8(double (&b)[8], const DctTw &T)
{
    double in2[4], ev[2], od[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const double bottom = b[i], top = b[7 - i], hb = b[3 - i], ht = b[4 + i];
        in2[i] = top + bottom;
        in2[3 - i] = hb + ht;
        const double lower = bottom - top, upper = hb - ht;
        ev[i] = lower * T.t8[i][0] + upper * T.t8[i][1];
        const double sin_in = upper * T.t8[i][0] - lower * T.t8[i][1];
        od[1 - i] = (i % 2 == 0) ? sin_in : -sin_in;
    }
    dct2_len4(in2, T);
    dct2_len2(ev[0], ev[1], T);
    dct2_len2(od[0], od[1], T);
    b[0] = in2[0]; b[1] = ev[0]; b[2] = in2[1];
    b[3] = ev[1] + od[1]; b[4] = in2[2]; b[5] = ev[1] - od[1]; b[6] = in2[3];
    b[7] = -od[0];
}

__device__ __forceinline__ void dct16_pruned(const double (&v)[16], double (&out)[10], const DctTw &T)
{
    double in2[8], ev[4], od[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double bottom = v[i], top = v[15 - i], hb = v[7 - i], ht = v[8 + i];
        in2[i] = top + bottom;
        in2[7 - i] = hb + ht;
        const double lower = bottom - top, upper = hb - ht;
        ev[i] = lower * T.t16[i][0] + upper * T.t16[i][1];
        const double sin_in = upper * T.t16[i][0] - lower * T.t16[i][1];
        od[3 - i] = (i % 2 == 0) ? sin_in : -sin_in;
    }
    dct2_len8(in2, T);
    dct2_len4(ev, T);
    dct2_len4(od, T);
    out[0] = in2[0]; out[1] = ev[0]; out[2] = in2[1];
    // i = 1: (i + 4) odd -> +od[3];  i = 2: (i + 4) even -> -od[2]
    out[3] = ev[1] + od[3]; out[4] = in2[2]; out[5] = ev[1] - od[3]; out[6] = in2[3];
    out[7] = ev[2] + (-od[2]); out[8] = in2[4]; out[9] = ev[2] - (-od[2]);
}

constexpr int kPadY = 17;  // [t][ky][x] rows padded to 17 doubles: pass-y writes and pass-x reads are conflict-free
// Second-pass output [t][10 kx + ky] with a t-stride of 106 doubles: the pass-x writes (16-lane groups over consecutive
// (t, ky)) and the pass-t reads (lane = 10 kx + ky, consecutive doubles) are both free of bank conflicts; strides 100..131
// were enumerated against the bank model of MI355X_MICROARCH.md (only 106 and 122 are clean).  Overwriting the first-pass
// rows in place (round 1) cost 48-72 conflict cycles per clip in pass t (17 % of the kernel's LDS-active cycles).
constexpr int kStrideT = 106;

struct DctShared {
    union {                         // the second-pass output overlays the first-pass rows (a barrier separates the last read of b
        double b[16 * 10 * kPadY];  // first-pass output [t][ky][x]                 from the first write of c): 26 KB per workgroup
        double c[16 * kStrideT];    // second-pass output [t][10 kx + ky]           instead of 40, i.e. 6 instead of 4 workgroups per CU
    };                              //                                               for the DCT-bound 16 x 16 input path
    uint32_t words[32];         // the 1024 hash bits, OR-assembled from per-kt ballots
    uint32_t dc[4];
    // Resized clip as CENTRED bytes (pix - 128 as i8), one dword per (t, g, x): byte r = pixel (y = 4 g + r, x).
    // This is exactly what a lane of the vertical resize MFMA holds (C layout), so a frame is stored with one
    // conflict-free ds_write_b32 per lane, and the thread (t, x) of the first DCT pass finds its 16 y-values
    // in 4 dwords.
    __attribute__((aligned(16))) uint32_t cube[16 * 64];
};

// 256 threads: 3-D DCT-II of the clip in sh.cube (f64), sign test, ballot pack.  Pass order y, x, t (the
// reference's order, raw_dct_ops.rs:118-132).  Caller has filled sh.cube and sh.cosv and synchronised.
__device__ __forceinline__ void dct_hash_block(DctShared &sh, const_f64_ptr cosv, size_t clip,
                                               uint64_t *__restrict__ out_hashes, uint32_t *__restrict__ out_dontcare)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DctTw cm;  // wave-uniform constants (scalar loads): cos_table[256 ..] = t16, t8, t4, h in this order
#pragma unroll
    for (int i = 0; i < 4; i++) { cm.t16[i][0] = cosv[256 + 2 * i]; cm.t16[i][1] = cosv[257 + 2 * i]; }
#pragma unroll
    for (int i = 0; i < 2; i++) { cm.t8[i][0] = cosv[264 + 2 * i]; cm.t8[i][1] = cosv[265 + 2 * i]; }
    cm.t4[0] = cosv[268]; cm.t4[1] = cosv[269]; cm.h = cosv[270];
    // pass y: thread (t, x) owns one 16-pixel column
    {
        const uint32_t t = tid >> 4, x = tid & 15;
        double v[16], o[10];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int32_t w = (int32_t)sh.cube[t * 64 + g * 16 + x];
#pragma unroll
            for (int r = 0; r < 4; r++) v[4 * g + r] = (double)((w << (24 - 8 * r)) >> 24);  // sign-extending byte extract
        }
        dct16_pruned(v, o, cm);
#pragma unroll
        for (int ky = 0; ky < 10; ky++) sh.b[(t * 10 + ky) * kPadY + x] = o[ky];
    }
    __syncthreads();
    // pass x: thread (t, ky), 160 lines
    double ox[10];
    if (tid < 160) {
        double v[16];
#pragma unroll
        for (int x = 0; x < 16; x++) v[x] = sh.b[tid * kPadY + x];
        dct16_pruned(v, ox, cm);
    } else if (tid >= 192 && tid < 224) {
        sh.words[tid - 192] = 0u;  // idle lanes clear the ballot words (previous clip's words were read two barriers ago)
    }
    __syncthreads();  // every row of b has been read: c may overwrite it
    if (tid < 160) {
        const uint32_t t = tid / 10, ky = tid - 10 * t;
#pragma unroll
        for (int kx = 0; kx < 10; kx++) sh.c[t * kStrideT + 10 * kx + ky] = ox[kx];
    }
    __syncthreads();
    // pass t + sign + pack: thread rem = 10 kx + ky (100 threads) reads its 16 t-values once and produces the ten
    // kt outputs; the ballot of output kt over wave w is the 64 (or 36) hash bits starting at bit 100 kt + 64 w
    // (dct_3d.rs:55-66: bit i = 100 kt + 10 kx + ky), OR-ed into the LDS words by lane kt of the wave.
    uint32_t dc = 0;
    if (wave < 2) {
        const uint32_t rem = tid;  // < 128; lanes with rem >= 100 idle
        const bool live = rem < 100;
        const uint32_t at = live ? rem : 0;
        double v[16], o[10];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = sh.c[t * kStrideT + at];
        dct16_pruned(v, o, cm);
        unsigned long long piece = 0;
#pragma unroll
        for (int kt = 0; kt < 10; kt++) {
            const unsigned long long bits = __builtin_amdgcn_ballot_w64(live && o[kt] > 0.0);  // 0.0 and NaN -> 0
            const unsigned long long tiny = __builtin_amdgcn_ballot_w64(live && fabs(o[kt]) < 1e-6);
            dc += (uint32_t)__builtin_popcountll(tiny);
            if (lane == (uint32_t)kt) piece = bits;
        }
        if (lane < 10) {
            const uint32_t off = 100u * lane + 64u * wave, wi = off >> 5, sh_l = off & 31u;
            const uint32_t lo = (uint32_t)piece, hi = (uint32_t)(piece >> 32);
            atomicOr(&sh.words[wi], lo << sh_l);
            atomicOr(&sh.words[wi + 1], (hi << sh_l) | (sh_l ? lo >> (32u - sh_l) : 0u));
            if (sh_l && wi + 2 < 32) atomicOr(&sh.words[wi + 2], hi >> (32u - sh_l));
        }
        if (lane == 0) sh.dc[wave] = dc;
    }
    __syncthreads();
    if (tid < 16) {
        const unsigned long long w = (unsigned long long)sh.words[2 * tid] | ((unsigned long long)sh.words[2 * tid + 1] << 32);
        out_hashes[clip * 16 + tid] = w;
    }
    if (out_dontcare && tid == 0) out_dontcare[clip] = sh.dc[0] + sh.dc[1];
}

// 16 x 16 x 16 u8 cubes (row-major frames) -> hashes.  The cube is re-laid into sh.cube's dword layout.
__global__ __launch_bounds__(256) void dct_hash_kernel(const uint8_t *__restrict__ small, size_t clip_stride,
                                                       size_t frame_stride, const double *__restrict__ cos_table,
                                                       uint64_t *__restrict__ out_hashes,
                                                       uint32_t *__restrict__ out_dontcare)
{
    __shared__ DctShared sh;
    const size_t clip = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    if (tid < 32) sh.words[tid] = 0u;
    {
        // thread (t, g, xq): rows 4g..4g+3, columns 4xq..4xq+3 of frame t -> a 4x4 byte transpose in registers
        const uint32_t t = tid >> 4, g = (tid >> 2) & 3, xq = tid & 3;
        const uint8_t *src = small + clip * clip_stride + (size_t)t * frame_stride + (4 * g) * 16 + 4 * xq;
        uint32_t row[4];
#pragma unroll
        for (int r = 0; r < 4; r++) row[r] = *reinterpret_cast<const uint32_t *>(src + r * 16) ^ 0x80808080u;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint32_t w = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) w |= ((row[r] >> (8 * c)) & 255u) << (8 * r);
            sh.cube[t * 64 + g * 16 + 4 * xq + c] = w;
        }
    }
    __syncthreads();
    dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, clip, out_hashes, out_dontcare);
}

// ---- resize on the matrix cores -----------------------------------------------------------------
// A Lanczos resize is two banded integer matrix products per frame:
//     tmp[y][o]  = clip8((sum_x P[y][x]  * Ch[o][x]  + 2^(ph-1)) >> ph)      rows x taps
//     out[oy][o] = clip8((sum_y Cv[oy][y] * tmp[y][o] + 2^(pv-1)) >> pv)
// v_mfma_i32_16x16x64_i8 computes a 16 x 16 x 64 block of either exactly (i8 x i8 -> i32).  Pixels are
// centred (p ^ 0x80 = p - 128 as i8), each i16 coefficient is split 256 hi + lo, and the bias table restores
// the unsigned sum: result bit-identical to the scalar fixed-point loop (resize_generic_kernel, the oracle).
// Lane maps (probed, tools/probe_mfma_i8.hip): A row / B col / C col = lane & 15; the 16 operand bytes of lane
// group g = lane >> 4 are 16 k-values, any order as long as A and B agree; C row = 4 g + reg.
//   horizontal: A = pixels  (byte j <-> x = 64 kt + 16 g + j: one 16-byte load per lane), B = Ch table
//   vertical:   B = tmp     (byte 4 m + r <-> y = 64 rg + 16 m + 4 g + r: exactly the C layout of the four
//               horizontal blocks m = 0..3 of a 64-row group, so tmp never leaves registers), A = Cv table
// Cache policy of the frame streams: every byte is read exactly once, so the LDS-DMA loads carry the gfx950 `nt` bit
// (aux = 2) and stay out of L2 / the Infinity Cache.  Same box, plain -> nt, TB/s of frame bytes: 1280 x 720 6.04 -> 6.40-6.61,
// 1024 x 576 5.99 -> 6.63, 3840 x 2160 6.24 -> 6.60, 480 x 270 5.96 -> 6.29, 1920 x 1080 6.08 -> 6.19, 854 x 480 6.04 -> 6.15
// (profiles/r03_hash_nt_ab.txt).  -DVDF_STREAM_AUX=0 builds the plain form.
#ifndef VDF_STREAM_AUX
#define VDF_STREAM_AUX 2
#endif
typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));

struct MfmaResizeTables {
    const v4i *bh;          // [kt][2][64]   horizontal B operand (hi, lo)
    const v4i *av;          // [rg][2][64]   vertical A operand (hi, lo)
    const int32_t *bias_h;  // [16]
    const int32_t *bias_v;  // [16]
    int32_t prec_h, prec_v, n_kt, n_rg;
    const int32_t *band_meta;  // bh in band form (linear-stream kernel, wide frames): kt_lo[16], nt[16]
    int32_t band_stride;
};

// NT: non-temporal load.  Frames are read exactly once; where one wave instruction consumes whole 128-byte lines (the
// persistent 64 x 64 kernel: 16 rows x 64 B = 1 KB contiguous) keeping them out of L2 / the Infinity Cache is worth 7 % of the
// stream (1.120 -> 1.043 ms per 100 k clips in one run).  Where two instructions share a line (128-wide frames read as
// 16 rows x 64 B) it is 30 % SLOWER (1.058 -> 1.376 ms per 20 k clips): those kernels keep plain loads.
template <bool CAREFUL, bool NT = false>
__device__ __forceinline__ v4i load_pixels16(const uint8_t *p, const uint8_t *buf_end)
{
    if (!CAREFUL || p + 16 <= buf_end) {
        u32x4_unaligned v;
        if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_unaligned *>(p));
        else v = *reinterpret_cast<const u32x4_unaligned *>(p);
        v4i r = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        return r;
    }
    uint32_t w[4] = {0, 0, 0, 0};  // last bytes of the buffer: never read past the end
    for (int i = 0; i < 16; i++)
        if (p + i < buf_end) w[i >> 2] |= (uint32_t)p[i] << ((i & 3) * 8);
    v4i r = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    return r;
}

// v_ashr_pk_u8_i32 d, a, b, s: d[7:0] = sat_u8(a >> s), d[15:8] = sat_u8(b >> s) - shift, clamp and pack for two values in
// one VALU op.  Only bits 15:0 of the result are defined (hipcc's own use of it trusts the upper half to be zero, which it is
// not: found by the parity tests), so the two halves are joined with a byte permute that never looks at the upper bits.
__device__ __forceinline__ uint32_t ashr_pk_u8(int32_t a, int32_t b, int shift)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(shift));
    return d;
}

__device__ __forceinline__ uint32_t finalize4(v4i hi, v4i lo, int prec)
{  // four (256 hi + lo) >> prec, clamped to u8, packed little-endian, re-centred for the next i8 product
    const uint32_t p01 = ashr_pk_u8((hi[0] << 8) + lo[0], (hi[1] << 8) + lo[1], prec);
    const uint32_t p23 = ashr_pk_u8((hi[2] << 8) + lo[2], (hi[3] << 8) + lo[3], prec);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u;  // bytes: p01.b0, p01.b1, p23.b0, p23.b1
}

// Row groups rg_begin, rg_begin + rg_step, ... of one frame; accumulates the vertical partial sums.
// CAREFUL = this frame's 16-byte loads may reach past the end of the caller's buffer (last frame only).
template <bool CAREFUL>
__device__ __forceinline__ void resize_row_groups(const uint8_t *__restrict__ src, uint32_t W, uint32_t H,
                                                  const uint8_t *buf_end, const MfmaResizeTables &T, int rg_begin,
                                                  int rg_step, v4i &acc_vh, v4i &acc_vl, size_t pitch = 0)
{
    if (pitch == 0) pitch = W;  // tightly packed rows unless a crop box is read in place
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
    const int32_t bias_h = T.bias_h[r16];
    for (int rg = rg_begin; rg < T.n_rg; rg += rg_step) {
        v4i ah[4], al[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { ah[m] = (v4i){0, 0, 0, 0}; al[m] = (v4i){bias_h, bias_h, bias_h, bias_h}; }
        // K-tile loop, software pipelined by hand: the four 16-byte pixel loads (and the two table fragments) of tile
        // kt + 1 are issued before the eight MFMAs of tile kt, so a wave always has a tile's worth of loads in flight
        auto load_tile = [&](int kt, v4i (&px)[4], v4i &tbh, v4i &tbl) {
            tbh = T.bh[(kt * 2 + 0) * 64 + lane];
            tbl = T.bh[(kt * 2 + 1) * 64 + lane];
            const uint32_t x = 64u * kt + 16u * g;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t row = 64u * rg + 16u * m + r16;
                px[m] = (v4i){0, 0, 0, 0};
                if (row < H && x < W) px[m] = load_pixels16<CAREFUL>(src + (size_t)row * pitch + x, buf_end);
            }
        };
        const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
        v4i cur[4], nxt[4], cbh, cbl, nbh, nbl;
        load_tile(0, cur, cbh, cbl);
        for (int kt = 0; kt < T.n_kt; kt++) {
            const bool more = kt + 1 < T.n_kt;
            if (more) load_tile(kt + 1, nxt, nbh, nbl);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const v4i a = cur[m] ^ x80;
                ah[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, cbh, ah[m], 0, 0, 0);
                al[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, cbl, al[m], 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int m = 0; m < 4; m++) cur[m] = nxt[m];
                cbh = nbh; cbl = nbl;
            }
        }
        v4i b;
#pragma unroll
        for (int m = 0; m < 4; m++) b[m] = (int)finalize4(ah[m], al[m], T.prec_h);
        acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 0) * 64 + lane], b, acc_vh, 0, 0, 0);
        acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 1) * 64 + lane], b, acc_vl, 0, 0, 0);
    }
}

// Large frames: the same arithmetic over a contiguous range of 16-row BLOCKS [blk_begin, blk_end) of one frame, so that
// the four waves of a workgroup get equal shares whatever H is (1080 rows = 68 blocks = 17 each; 64-row groups dealt
// round-robin gave 5/4/4/4, and 270 rows gave 2/1/1/1).  Blocks of a 64-row group that belong to another wave enter this
// wave's vertical product as i8 zeros (contribution 0; the +128 re-centring bias is added once, by the caller).
template <bool CAREFUL>
__device__ __forceinline__ void resize_row_blocks(const uint8_t *__restrict__ src, uint32_t W, uint32_t H,
                                                  const uint8_t *buf_end, const MfmaResizeTables &T, int blk_begin,
                                                  int blk_end, v4i &acc_vh, v4i &acc_vl, size_t pitch = 0)
{
    if (pitch == 0) pitch = W;
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
    const int32_t bias_h = T.bias_h[r16];
    for (int rg = blk_begin >> 2; 4 * rg < blk_end && rg < T.n_rg; rg++) {
        const int m_lo = max(blk_begin - 4 * rg, 0), m_hi = min(blk_end - 4 * rg, 4);  // wave-uniform
        v4i ah[4], al[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { ah[m] = (v4i){0, 0, 0, 0}; al[m] = (v4i){bias_h, bias_h, bias_h, bias_h}; }
        auto load_tile = [&](int kt, v4i (&px)[4], v4i &tbh, v4i &tbl) {
            tbh = T.bh[(kt * 2 + 0) * 64 + lane];
            tbl = T.bh[(kt * 2 + 1) * 64 + lane];
            const uint32_t x = 64u * kt + 16u * g;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t row = 64u * rg + 16u * m + r16;
                px[m] = (v4i){0, 0, 0, 0};
                if (m >= m_lo && m < m_hi && row < H && x < W)
                    px[m] = load_pixels16<CAREFUL>(src + (size_t)row * pitch + x, buf_end);
            }
        };
        const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
        v4i cur[4], nxt[4], cbh, cbl, nbh, nbl;
        load_tile(0, cur, cbh, cbl);
        for (int kt = 0; kt < T.n_kt; kt++) {
            const bool more = kt + 1 < T.n_kt;
            if (more) load_tile(kt + 1, nxt, nbh, nbl);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (m >= m_lo && m < m_hi) {
                    const v4i a = cur[m] ^ x80;
                    ah[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, cbh, ah[m], 0, 0, 0);
                    al[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, cbl, al[m], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int m = 0; m < 4; m++) cur[m] = nxt[m];
                cbh = nbh; cbl = nbl;
            }
        }
        v4i b;
#pragma unroll
        for (int m = 0; m < 4; m++) b[m] = (m >= m_lo && m < m_hi) ? (int)finalize4(ah[m], al[m], T.prec_h) : 0;
        acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 0) * 64 + lane], b, acc_vh, 0, 0, 0);
        acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 1) * 64 + lane], b, acc_vl, 0, 0, 0);
    }
}

// Small frames: one workgroup per clip, one wave per frame (4 frames each), resized frames go to LDS and the
// DCT runs in the same kernel: HBM traffic = the frames once + 128 B of hash.
// ONE_TILE (W, H <= 64): the wave's 16 pixel loads (4 frames x 4 row blocks, 16 KB) are all issued before the
// first MFMA so a workgroup keeps its whole 64 KB clip in flight; tables stay in registers.
template <bool ONE_TILE>
__global__ __launch_bounds__(256) void resize_dct_hash_fused_kernel(
    const uint8_t *__restrict__ frames, uint32_t W, uint32_t H, size_t frame_stride, size_t clip_stride,
    const uint8_t *buf_end, MfmaResizeTables T, const double *__restrict__ cos_table,
    uint64_t *__restrict__ out_hashes, uint32_t *__restrict__ out_dontcare)
{
    __shared__ DctShared sh;
    const size_t clip = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    if (tid < 32) sh.words[tid] = 0u;
    const uint8_t *clip_base = frames + clip * clip_stride;
    // only the last frames of the buffer can see a 16-byte load cross its end (wave-uniform test)
    const bool careful = clip_base + 15 * frame_stride + (size_t)W * H + 64 > buf_end;
    v4i bias_v;
#pragma unroll
    for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
    if (ONE_TILE && !careful) {
        const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
        v4i px[4][4];
        const bool col_ok = 16u * g < W;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint8_t *src = clip_base + (size_t)(wave + 4 * q) * frame_stride;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t row = 16u * m + r16;
                px[q][m] = (v4i){0, 0, 0, 0};
                if (row < H && col_ok) px[q][m] = load_pixels16<false>(src + (size_t)row * W + 16u * g, buf_end);
            }
        }
        const v4i bh = T.bh[lane], bl = T.bh[64 + lane], avh = T.av[lane], avl = T.av[64 + lane];
        const int32_t bias_h = T.bias_h[r16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v4i b;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const v4i a = px[q][m] ^ x80;
                v4i ah = {0, 0, 0, 0}, al = {bias_h, bias_h, bias_h, bias_h};
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bh, ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bl, al, 0, 0, 0);
                b[m] = (int)finalize4(ah, al, T.prec_h);
            }
            v4i vh = {0, 0, 0, 0}, vl = bias_v;
            vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(avh, b, vh, 0, 0, 0);
            vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(avl, b, vl, 0, 0, 0);
            // centred bytes of out[oy = 4 g + r][x = r16], r = 0..3: one dword per lane
            sh.cube[(wave + 4 * q) * 64 + g * 16 + r16] = finalize4(vh, vl, T.prec_v);
        }
    } else {
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const uint32_t f = wave + 4 * q;
            v4i vh = {0, 0, 0, 0}, vl = bias_v;
            if (careful) resize_row_groups<true>(clip_base + (size_t)f * frame_stride, W, H, buf_end, T, 0, 1, vh, vl);
            else resize_row_groups<false>(clip_base + (size_t)f * frame_stride, W, H, buf_end, T, 0, 1, vh, vl);
            sh.cube[f * 64 + g * 16 + r16] = finalize4(vh, vl, T.prec_v);
        }
    }
    __syncthreads();
    dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, clip, out_hashes, out_dontcare);
}

// Persistent form of the ONE_TILE fused kernel (W, H <= 64, W % 16 == 0 so no load crosses the buffer end):
// a workgroup loops over clips and issues the NEXT clip's 16 loads per lane (64 KB per workgroup) as soon as the
// resize has consumed the current pixels, so HBM streams underneath the DCT instead of after it.
// FULL = W == H == 64: every lane's loads are in range, so the zero fill and the per-lane predicates (exec-mask
// juggling, 58 register moves per wave and clip) disappear.
template <bool FULL>
__global__ __launch_bounds__(256) void resize_dct_hash_persistent_kernel(
    const uint8_t *__restrict__ frames, uint32_t W, uint32_t H, size_t frame_stride, size_t clip_stride,
    MfmaResizeTables T, const double *__restrict__ cos_table, uint64_t *__restrict__ out_hashes,
    uint32_t *__restrict__ out_dontcare, uint32_t n_clips)
{
    __shared__ DctShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    if (tid < 32) sh.words[tid] = 0u;
    v4i bias_v;
#pragma unroll
    for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    const v4i bh = T.bh[lane], bl = T.bh[64 + lane], avh = T.av[lane], avl = T.av[64 + lane];
    const int32_t bias_h = T.bias_h[r16];
    const bool col_ok = 16u * g < W;
    const size_t lane_off = (size_t)r16 * W + 16u * g;  // row r16 of row block m is at + 16 m W

#ifndef VDF_HASH_STRIDED  // wave w takes frames 4 w .. 4 w + 3: 16 KB contiguous per wave (frames w, w + 4, .. measured 2 % slower)
    const uint32_t f0 = 4 * wave, fstep = 1;
#else
    const uint32_t f0 = wave, fstep = 4;
#endif
#ifdef VDF_HASH_NO_NT
    constexpr bool kNT = false;
#else
    constexpr bool kNT = true;
#endif
    v4i px[4][4];
    auto issue_loads = [&](size_t clip) {
        const uint8_t *base = frames + clip * clip_stride + (size_t)f0 * frame_stride + lane_off;
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (!FULL) px[q][m] = (v4i){0, 0, 0, 0};
                if (FULL || (16u * m + r16 < H && col_ok))
                    px[q][m] = load_pixels16<false, kNT>(base + (size_t)(fstep * q) * frame_stride + (size_t)(16 * m) * W, nullptr);
            }
        }
    };
    auto issue_loads_q = [&](size_t clip, int q) {
        const uint8_t *base = frames + clip * clip_stride + (size_t)f0 * frame_stride + lane_off;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            if (!FULL) px[q][m] = (v4i){0, 0, 0, 0};
            if (FULL || (16u * m + r16 < H && col_ok))
                px[q][m] = load_pixels16<false, kNT>(base + (size_t)(fstep * q) * frame_stride + (size_t)(16 * m) * W, nullptr);
        }
    };
    uint32_t clip = blockIdx.x;
    if (clip < n_clips) issue_loads(clip);
    while (clip < n_clips) {
        // The resize and the loads it releases are this workgroup's memory-critical path: while it waits behind the DCTs of the
        // other two workgroups on its SIMDs nothing of its next clip is in flight.  So the resize runs at raised priority and each
        // frame quartet's loads go out as soon as its pixels are consumed (not after the whole resize): 1.152-1.187 -> 1.117-1.127 ms
        // per 100 k clips in one run (without any DCT the load + resize stream alone does 1.08).
        __builtin_amdgcn_s_setprio(3);
        const uint32_t next = clip + gridDim.x;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v4i b;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const v4i a = px[q][m] ^ x80;
                v4i ah = {0, 0, 0, 0}, al = {bias_h, bias_h, bias_h, bias_h};
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bh, ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bl, al, 0, 0, 0);
                b[m] = (int)finalize4(ah, al, T.prec_h);
            }
            v4i vh = {0, 0, 0, 0}, vl = bias_v;
            vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(avh, b, vh, 0, 0, 0);
            vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(avl, b, vl, 0, 0, 0);
            sh.cube[(f0 + fstep * q) * 64 + g * 16 + r16] = finalize4(vh, vl, T.prec_v);
            if (next < n_clips) issue_loads_q(next, q);  // in flight during the whole DCT below
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
#ifdef VDF_ABL_NO_DCT  // ablation build: what the load + resize stream reaches without the DCT (results are wrong)
        if (tid < 16) out_hashes[(size_t)clip * 16 + tid] = sh.cube[tid * 64];
        __syncthreads();
#else
        dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, clip, out_hashes, out_dontcare);
#endif
        clip = next;
    }
}

// Persistent form for frames of up to 128 x 128 (NKT x NRG tiles of 64 x 64, W % 16 == 0; round 5).  The one-workgroup-per-clip kernel above keeps
// ONE tile's loads in flight per wave (4 KB) and its frames' streams end at every DCT: 96 x 64 read at 4.2 TB/s, 64 x 128 at 4.6, 32 x 128 at 2.7.
// Here a wave issues its loads in UNITS of eight 16-byte loads per lane (8 KB per wave: a whole frame of up to two tiles, or one 64-row group
// of a 2 x 2-tile frame) into one of two register buffers: unit u + 1 goes out before the products of unit u run, and the next clip's first
// unit before the DCT - as in the 64 x 64 kernel, nothing of the stream waits for the DCT.  Tables (8 fragments at most) stay in registers.
// (Whole 2 x 2-tile frames as units - 16 loads, 241 registers, two workgroups per CU - measured no better than the per-clip kernel: 128 x 128
// 4.66 against 4.77 TB/s; the unit of eight keeps all three shapes at three workgroups per CU.)
// Plain loads: two instructions share a 128-byte line here (see load_pixels16).  Same products, exact integer sums: bit-identical.
// (WAVES = the register bound, waves per SIMD: the 2 x 2 shape needs it - 168 registers keep it at three workgroups per CU; put on the
// two-tile shapes, which fit anyway, it changed their schedule for the worse: 80 x 48 5.3 -> 4.1 TB/s, 32 x 128 5.7 -> 3.8)
// Three or four K tiles (129 ... 256 columns): a unit is TWO of the four 16-row blocks of a row group (2 x NKT loads); the block results wait
// in `b` for the row group's second unit, whose end is the vertical product.
template <int NKT, int NRG, int WAVES>
__global__ __launch_bounds__(256, WAVES) void resize_dct_hash_tiled_kernel(
    const uint8_t *__restrict__ frames, uint32_t W, uint32_t H, size_t frame_stride, size_t clip_stride,
    MfmaResizeTables T, const double *__restrict__ cos_table, uint64_t *__restrict__ out_hashes,
    uint32_t *__restrict__ out_dontcare, uint32_t n_clips)
{
    constexpr int MU = NKT > 2 ? 2 : 4;                 // 16-row blocks per unit
    constexpr int RGU = NKT == 1 && NRG >= 2 ? 2 : 1;   // row groups per unit: two where a row group is only four loads
    constexpr int UPF = (NRG / RGU) * (4 / MU);         // units per frame
    static_assert(NRG % RGU == 0, "whole units");
    constexpr int NU = 4 * UPF;                         // units per wave and clip (four frames)
    constexpr bool FETCH_AV = UPF > 1;                  // vertical fragments fetched per unit instead of living in registers
    static_assert(NU % 2 == 0, "the two buffers alternate: the next clip's first unit lands in buffer 0");
    __shared__ DctShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    if (tid < 32) sh.words[tid] = 0u;
    v4i bias_v;
#pragma unroll
    for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    v4i bh[NKT][2], av[NRG][2];
#pragma unroll
    for (int kt = 0; kt < NKT; kt++) { bh[kt][0] = T.bh[(kt * 2 + 0) * 64 + lane]; bh[kt][1] = T.bh[(kt * 2 + 1) * 64 + lane]; }
    static_assert(FETCH_AV || NRG < 3, "fragments in registers: every row group exists");
    if constexpr (!FETCH_AV) {
#pragma unroll
        for (int rg = 0; rg < NRG; rg++) { av[rg][0] = T.av[(rg * 2 + 0) * 64 + lane]; av[rg][1] = T.av[(rg * 2 + 1) * 64 + lane]; }
    }
    const int32_t bias_h = T.bias_h[r16];
    const uint32_t f0 = 4 * wave;  // this wave's frames: 4 wave .. 4 wave + 3
    const size_t lane_off = (size_t)r16 * W + 16u * g;
    typedef v4i UnitPx[RGU][NKT][MU];
    auto issue = [&](size_t clip, int u, UnitPx &px) __attribute__((always_inline)) {
        const int q = u / UPF, r = u % UPF, rg0 = (r / (4 / MU)) * RGU, m0 = (r % (4 / MU)) * MU;
        const uint8_t *base = frames + clip * clip_stride + (size_t)(f0 + q) * frame_stride + lane_off;
#pragma unroll
        for (int ri = 0; ri < RGU; ri++)
#pragma unroll
            for (int kt = 0; kt < NKT; kt++)
#pragma unroll
                for (int mi = 0; mi < MU; mi++) {
                    const uint32_t row0 = 64u * (rg0 + ri) + 16u * (m0 + mi);
                    px[ri][kt][mi] = (v4i){0, 0, 0, 0};
                    if (row0 + r16 < H && 64u * kt + 16u * g < W) px[ri][kt][mi] = load_pixels16<false>(base + (size_t)row0 * W + 64 * kt, nullptr);
                }
    };
    v4i vh = {0, 0, 0, 0}, vl = bias_v;  // the frame's vertical sums, carried over its units
    v4i b = {0, 0, 0, 0};                // the row group's four block results, carried over its units (MU == 2)
    auto products = [&](int u, const UnitPx &px) __attribute__((always_inline)) {
        const int q = u / UPF, r = u % UPF, rg0 = (r / (4 / MU)) * RGU, m0 = (r % (4 / MU)) * MU;
#pragma unroll
        for (int ri = 0; ri < RGU; ri++) {
            // (several units per frame: the vertical fragments are fetched per row group - L1 hits, consumed behind the unit's products - instead
            // of living in up to 32 registers: that is what keeps the 2 x 2 shape at three workgroups per CU without a spill.  NRG = 4 also
            // serves frames of three row groups: the fourth has no rows - no loads - and no fragments)
            const bool live = NRG < 3 || rg0 + ri < T.n_rg;  // workgroup-uniform
            v4i avh_u = {0, 0, 0, 0}, avl_u = {0, 0, 0, 0};
            if constexpr (FETCH_AV) {
                if (m0 + MU == 4 && live) {
                    avh_u = T.av[((rg0 + ri) * 2 + 0) * 64 + lane];
                    avl_u = T.av[((rg0 + ri) * 2 + 1) * 64 + lane];
                }
            }
#pragma unroll
            for (int mi = 0; mi < MU; mi++) {
                v4i ah = {0, 0, 0, 0}, al = {bias_h, bias_h, bias_h, bias_h};
#pragma unroll
                for (int kt = 0; kt < NKT; kt++) {
                    const v4i a = px[ri][kt][mi] ^ x80;
                    ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bh[kt][0], ah, 0, 0, 0);
                    al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bh[kt][1], al, 0, 0, 0);
                }
                b[m0 + mi] = (int)finalize4(ah, al, T.prec_h);
            }
            if (m0 + MU == 4 && live) {  // the row group's last unit
                vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(FETCH_AV ? avh_u : av[rg0 + ri][0], b, vh, 0, 0, 0);
                vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(FETCH_AV ? avl_u : av[rg0 + ri][1], b, vl, 0, 0, 0);
            }
        }
        if (rg0 + RGU == NRG && m0 + MU == 4) {  // the frame's last unit
            sh.cube[(f0 + q) * 64 + g * 16 + r16] = finalize4(vh, vl, T.prec_v);
            vh = (v4i){0, 0, 0, 0};
            vl = bias_v;
        }
    };
    UnitPx pa, pb;
    uint32_t clip = blockIdx.x;
    if (clip < n_clips) issue(clip, 0, pa);
    while (clip < n_clips) {
        const uint32_t next = clip + gridDim.x;
        __builtin_amdgcn_s_setprio(3);  // as in the 64 x 64 kernel: the resize releases the loads the stream lives on
#pragma unroll
        for (int u = 0; u < NU; u += 2) {
            issue(clip, u + 1, pb);
            products(u, pa);
            if (u + 2 < NU) issue(clip, u + 2, pa);
            else if (next < n_clips) issue(next, 0, pa);  // in flight during the whole DCT below
            products(u + 1, pb);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, clip, out_hashes, out_dontcare);
        clip = next;
    }
}

// ---- large frames, coalesced ------------------------------------------------------------------------------
// Large frames, one workgroup per frame; the four waves' vertical partial sums (exact i32) are added through LDS and the 16 x 16
// u8 frame goes to `small`.  The natural A-operand shape (lane = row, 16 bytes of one K tile: resize_row_blocks) makes every
// wave load touch 16 rows x 64 B: half lines at the frame's pitch, which caps
// a 1080p stream at 5.3 TB/s (tools/ubench_rowload.hip; 8 rows x 128 B reads at 6.2, a linear sweep at 6.24).
// Here one wave load covers 8 rows x 128 B (whole lines) and still lands directly in MFMA operand registers: lane
// (a = lane & 15, g = lane >> 4) loads row R0 + (a >> 1), bytes 128 T + 64 (a & 1) + 16 g .. +15, i.e. the 16 A rows of
// the product are 8 image rows x 2 halves of the 128-byte window.  Half 0 is a row's K-tile 2T and half 1 its K-tile
// 2T + 1, so the window takes one product with tile 2T's coefficients (right for the even A rows, don't-care for the
// odd ones) and one with tile 2T + 1's (the reverse): twice the MFMAs of the uncoalesced form, still < 20 % of the
// matrix pipe at 6 TB/s.  C rows = 4 (lane >> 4) + reg, so the two halves of an image row sit in adjacent registers
// of one lane: row R0 + 2 G + j = even[2 j] + odd[2 j + 1], no cross-lane traffic.  A lane therefore ends up with
// rows 8 oct + 2 G + {0, 1} of every octet of a 64-row group - the k order of the vertical product is free, its
// coefficient operand is simply built in that order (kMfmaLayoutVerticalWide).  Bit-identical to the other kernels.
template <bool CAREFUL>
__device__ __forceinline__ void resize_row_quads(const uint8_t *__restrict__ src, uint32_t W, uint32_t H,
                                                 const uint8_t *buf_end, const MfmaResizeTables &T, int q_begin,
                                                 int q_end, v4i &acc_vh, v4i &acc_vl, size_t pitch = 0)
{
    if (q_begin >= q_end) return;
    if (pitch == 0) pitch = W;  // tightly packed rows unless a crop box is read in place
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, a16 = lane & 15;
    const uint32_t row8 = a16 >> 1, half = a16 & 1;
    const int32_t bias_h = T.bias_h[a16];  // C column = lane & 15 = output o
    const int n_win = (T.n_kt + 1) / 2;
    const v4i zero4 = {0, 0, 0, 0};
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    // one step = one 128-byte window of one 32-row quad: 4 pixel loads (one per octet) + the 4 coefficient fragments
    auto load_step = [&](int q, int w, v4i (&px)[4], v4i (&tb)[4]) {
        tb[0] = T.bh[((2 * w) * 2 + 0) * 64 + lane];
        tb[1] = T.bh[((2 * w) * 2 + 1) * 64 + lane];
        tb[2] = zero4; tb[3] = zero4;
        if (2 * w + 1 < T.n_kt) {
            tb[2] = T.bh[((2 * w + 1) * 2 + 0) * 64 + lane];
            tb[3] = T.bh[((2 * w + 1) * 2 + 1) * 64 + lane];
        }
        const uint32_t x = 128u * w + 64u * half + 16u * g;
#pragma unroll
        for (int oct = 0; oct < 4; oct++) {
            const uint32_t row = 32u * q + 8u * oct + row8;
            px[oct] = zero4;
#ifndef VDF_WIDE_NO_NT  // whole 128-byte lines per instruction, read once: non-temporal (1536 x 864: 5.66 -> 5.96 TB/s)
            if (row < H && x < W) px[oct] = load_pixels16<CAREFUL, true>(src + (size_t)row * pitch + x, buf_end);
#else
            if (row < H && x < W) px[oct] = load_pixels16<CAREFUL>(src + (size_t)row * pitch + x, buf_end);
#endif
        }
    };
    v4i eh[4], el[4], oh[4], ol[4];
    auto reset_acc = [&]() {
#pragma unroll
        for (int oct = 0; oct < 4; oct++) {
            eh[oct] = zero4; oh[oct] = zero4; ol[oct] = zero4;
            el[oct] = (v4i){bias_h, bias_h, bias_h, bias_h};
        }
    };
    reset_acc();
    v4i b = zero4;  // quads of a 64-row group that another wave owns stay i8 zeros: no contribution
    v4i cur[4], nxt[4], ct[4], nt[4];
    int q = q_begin, w = 0;
    load_step(q, w, cur, ct);
    // the (quad, window) steps form ONE software-pipelined stream: the next step's loads are issued before this step's
    // MFMAs also across a quad boundary, so narrow frames (few windows per quad) keep loads in flight too
    for (;;) {
        int qn = q, wn = w + 1;
        if (wn == n_win) { wn = 0; qn = q + 1; }
        const bool more = qn < q_end;
        if (more) load_step(qn, wn, nxt, nt);
#pragma unroll
        for (int oct = 0; oct < 4; oct++) {
            const v4i a = cur[oct] ^ x80;
            eh[oct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, ct[0], eh[oct], 0, 0, 0);
            el[oct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, ct[1], el[oct], 0, 0, 0);
            oh[oct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, ct[2], oh[oct], 0, 0, 0);
            ol[oct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, ct[3], ol[oct], 0, 0, 0);
        }
        if (w == n_win - 1) {  // quad complete: its 32 tmp rows -> two bytes per octet in this lane's vertical operand
            const int qh = q & 1;
#pragma unroll
            for (int oct = 0; oct < 4; oct++) {
                // rows 2G and 2G + 1 of the octet: even-tile part in registers 0 / 2, odd-tile part in registers 1 / 3
                const int32_t s0 = ((eh[oct][0] + oh[oct][1]) << 8) + el[oct][0] + ol[oct][1];
                const int32_t s1 = ((eh[oct][2] + oh[oct][3]) << 8) + el[oct][2] + ol[oct][3];
                uint32_t two = ashr_pk_u8(s0, s1, T.prec_h) & 0xFFFFu;  // only bits 15:0 of the packed result are defined
                two ^= 0x8080u;  // re-centred for the next i8 product
                // octet o8 = 4 qh + oct of the 64-row group: bytes 2 o8, 2 o8 + 1 of the operand (static register indices)
                const int val = (int)(two << (16 * (oct & 1)));
                if (qh) b[2 + (oct >> 1)] |= val; else b[oct >> 1] |= val;
            }
            reset_acc();
            if (qh == 1 || !more) {  // last quad this wave owns in the 64-row group
                const int rg = q >> 1;
                acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 0) * 64 + lane], b, acc_vh, 0, 0, 0);
                acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 1) * 64 + lane], b, acc_vl, 0, 0, 0);
                b = zero4;
            }
        }
        if (!more) break;
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2]; cur[3] = nxt[3];
        ct[0] = nt[0]; ct[1] = nt[1]; ct[2] = nt[2]; ct[3] = nt[3];
        q = qn; w = wn;
    }
}

// One workgroup per frame; the four waves take equal contiguous shares of the frame's 32-row quads.  T.av is in
// kMfmaLayoutVerticalWide order.
__global__ __launch_bounds__(256) void resize_mfma_frame_wide_kernel(const uint8_t *__restrict__ frames, uint32_t W,
                                                                     uint32_t H, size_t frame_stride,
                                                                     size_t clip_stride, const uint8_t *buf_end,
                                                                     MfmaResizeTables T, uint8_t *__restrict__ small)
{
    __shared__ int32_t s_part[3][2][64][4];
    const size_t clip = blockIdx.x >> 4;
    const uint32_t f = blockIdx.x & 15;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    v4i vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
    const uint8_t *src = frames + clip * clip_stride + (size_t)f * frame_stride;
    const int n_q = (int)((H + 31) / 32), q0 = n_q * (int)wave / 4, q1 = n_q * ((int)wave + 1) / 4;
    if (src + (size_t)W * H + 128 > buf_end) resize_row_quads<true>(src, W, H, buf_end, T, q0, q1, vh, vl);
    else resize_row_quads<false>(src, W, H, buf_end, T, q0, q1, vh, vl);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) { s_part[wave - 1][0][lane][r] = vh[r]; s_part[wave - 1][1][lane][r] = vl[r]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            vl[r] += T.bias_v[4 * g + r];
#pragma unroll
            for (int w = 0; w < 3; w++) { vh[r] += s_part[w][0][lane][r]; vl[r] += s_part[w][1][lane][r]; }
        }
        const uint32_t px = finalize4(vh, vl, T.prec_v) ^ 0x80808080u;
        uint8_t *dst = small + (clip * 16 + f) * 256;
#pragma unroll
        for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + r16] = (uint8_t)(px >> (8 * r));
    }
}

static MfmaResizeTables make_tables(const MfmaResizeArgs &a);

// ---- per-clip row ranges (letterboxed clips with top / bottom bars only) ---------------------------------------------------
// The linear-stream kernels below take, as ROWCROP instantiations, a per-clip first row and height: the crop box of a clip
// with full-width bars is a contiguous run of rows at the frame's own pitch, so the box streams exactly like a (shorter) frame
// and only the vertical table changes from clip to clip.  The geometry comes from the clip's descriptor (y0, h, v_table of
// CropStreamClip; the vertical CropStreamTable entry) by scalar loads, a frame ahead of its use.
typedef const __attribute__((address_space(4))) CropStreamClip *const_clip_ptr;
typedef const __attribute__((address_space(4))) CropStreamTable *const_table_ptr;
typedef const __attribute__((address_space(1))) v4i *global_v4i;
typedef const __attribute__((address_space(1))) int32_t *global_i32;
struct RowGeo {
    uint32_t y0, h, src;  // src: the clip's index in the caller's batch
    int32_t n_rg, prec_v;
    global_v4i av;
    global_i32 bias_v;
};
// workgroup-uniform values pinned to SGPRs; pointers rebuilt in the GLOBAL address space (a pointer made from integers is a generic
// one: its loads become flat_load, and the compiler puts a vmcnt(0) in front of a flat load while an LDS-DMA is in flight)
__device__ __forceinline__ uint32_t sgpr_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ const __attribute__((address_space(1))) void *sgpr_global_ptr(const void *p)
{
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return (const __attribute__((address_space(1))) void *)(uintptr_t)(((uint64_t)sgpr_u32((uint32_t)(a >> 32)) << 32) | sgpr_u32((uint32_t)a));
}
__device__ __forceinline__ RowGeo row_geo_of(const_clip_ptr clips, const_table_ptr tables, uint32_t F)
{
    const uint32_t clip = sgpr_u32(F >> 4);
    RowGeo q;
    q.y0 = sgpr_u32(clips[clip].y0);
    q.h = sgpr_u32(clips[clip].h);
    q.src = sgpr_u32(clips[clip].src_clip);
    const uint32_t vt = sgpr_u32(clips[clip].v_table);
    q.av = (global_v4i)sgpr_global_ptr(tables[vt].operand);
    q.bias_v = (global_i32)sgpr_global_ptr(tables[vt].bias);
    q.n_rg = (int32_t)sgpr_u32((uint32_t)tables[vt].n_tiles);
    q.prec_v = (int32_t)sgpr_u32((uint32_t)tables[vt].precision);
    return q;
}

// ---- large frames, linear-stream form ------------------------------------------------------------------------
// With W % 128 != 0 every row-shaped wave load above (16 x 64 B or 8 x 128 B) straddles lines: 480 x 270 read at 4.6 TB/s
// while the same bytes read linearly stream at 6.2 (tools/ubench_rowload.hip), and even line-aligned rows cap at 5.7.
// Here the global side IS linear: a persistent workgroup copies its frames chunk by chunk (16 nb rows = 16 nb W contiguous
// bytes) into LDS with LDS-DMA, 1 KB per wave instruction, and the MFMA operands are read back from LDS at (row, x), where
// the pitch costs nothing.  Chunks are double buffered: the DMA of chunk s + 1 is in flight while the waves run the
// products of chunk s, wave m taking 16-row block m of the chunk (the vertical partial sums of the waves are added through
// LDS at the end of a frame, as in the kernels above).  Nothing inside the loop waits on a global load: the horizontal
// table lives in LDS, the two vertical fragments of the NEXT chunk are requested before its DMA, and the frame's result
// is written after the next barrier - so the only vmcnt wait is the one in front of the barrier that hands a chunk over.
// The buffer resource is sized to the frame, so the last DMA of a frame cannot read past it (no CAREFUL variant).
// The two pixel buffers and the table are separate arrays so the compiler knows a DMA into one does not alias reads of
// the others.  Bit-identical to the other kernels (same products, same order of the exact integer sums).
//
// MODE 0 (W % 16 == 0): LDS holds the chunk as it is in memory, pitch W; one DMA instruction = 1 KB of the frame.
// MODE 1, 2 (any other width): rows are RE-PITCHED on the way in.  LDS holds them at a pitch Wp that is an odd multiple of
// 16 bytes - every operand read is an aligned ds_read_b128 and the 16 rows of a block fall into 16 different bank groups
// (at the frame's own pitch 854 wide put six rows on the same banks and the LDS pipe became the limit: 4.5 TB/s) - and
// each DMA lane fetches the 16 global bytes that belong at its LDS position: position row * Wp + x  <-  frame byte
// row * W + x.  The global side is still a linear sweep (a row's tail lanes run into the next row).  LDS-DMA ignores the
// low two bits of a global address, so MODE 2 (W % 4 != 0) starts each row at the dword below it and the operand read
// takes one more dword and shifts by the row's 0..3 bytes (v_alignbyte, a per-lane constant).
template <int BUF_BYTES, int TAB_TILES, int MODE, bool ROWCROP = false>
__global__ __launch_bounds__(256) void resize_mfma_frame_stream_kernel(const uint8_t *__restrict__ frames, uint32_t W,
                                                                       uint32_t H, size_t frame_stride,
                                                                       size_t clip_stride, uint32_t n_frames,
                                                                       MfmaResizeTables T, uint32_t nb, uint32_t Wp,
                                                                       uint8_t *__restrict__ small,
                                                                       const CropStreamClip *__restrict__ clips_g = nullptr,
                                                                       const CropStreamTable *__restrict__ tables_g = nullptr)
{
    __shared__ __attribute__((aligned(16))) uint4 s_tab[TAB_TILES * 2 * 64];
    __shared__ __attribute__((aligned(16))) uint4 s_px0[BUF_BYTES / 16];
    __shared__ __attribute__((aligned(16))) uint4 s_px1[BUF_BYTES / 16];
    __shared__ int32_t s_part[3][64][4];  // 256 hi + lo of waves 1..3 (the sums are exact in i32, as in finalize4)
    const uint32_t tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t rpc = 16u * nb, frame_bytes = W * H;
    // ROWCROP: rows y0 .. y0 + h of each frame of a clip, its own vertical table; geo = the frame whose products run next, geo_n = the
    // workgroup's frame after it (fetched when geo becomes current), geo_p = the frame whose result is pending
    const_clip_ptr clips = (const_clip_ptr)(uintptr_t)clips_g;
    const_table_ptr tables = (const_table_ptr)(uintptr_t)tables_g;
    RowGeo geo = {0u, H, 0u, T.n_rg, T.prec_v, (global_v4i)T.av, (global_i32)T.bias_v}, geo_n = geo;
    int32_t pend_prec = T.prec_v;
    // MODE 1, 2: where this lane's first DMA instruction of a chunk lands (LDS position P0 = 1024 wave + 16 lane = row * Wp + x) and
    // how far an instruction (4096 bytes of LDS further) moves it; the loop below only adds and compares - with a multiply
    // high / two multiplies per instruction (16 cycles each) the ISSUE of a chunk's DMA cost 0.4 us of a 2.5 us step
    uint32_t lane_x0 = 0, lane_ro0 = 0;
    const uint32_t step_rows = MODE ? 4096u / Wp : 0u, step_x = MODE ? 4096u - step_rows * Wp : 0u;
    if constexpr (MODE != 0) {
        const uint32_t P0 = 1024u * wave + 16u * lane, row0 = P0 / Wp;
        lane_x0 = P0 - row0 * Wp;
        lane_ro0 = row0 * W;
    }
    const int32_t bias_h = T.bias_h[r16];
    v4i bias_v = {0, 0, 0, 0};
    if constexpr (!ROWCROP) {
#pragma unroll
        for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
    }
    // horizontal table -> LDS, all K tiles
    const uint32_t tab_vecs = (uint32_t)T.n_kt * 128u;
    for (uint32_t i = tid; i < tab_vecs; i += 256u) {
        const v4i v = T.bh[i];
        s_tab[i] = uint4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
    }
    const v4i zero4 = {0, 0, 0, 0};
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};

    auto issue_dma = [&](uint32_t F, uint32_t c, uint4 *dst, const RowGeo &q) __attribute__((always_inline)) {
        const uint8_t *src = frames + (size_t)(ROWCROP ? q.src : F >> 4) * clip_stride + (size_t)(F & 15u) * frame_stride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, frame_bytes, 0x00020000);
        const uint32_t start = (q.y0 + c * rpc) * W, rows = min(rpc, q.h - c * rpc), bytes = rows * Wp;
        uint32_t x = lane_x0, ro = lane_ro0;  // MODE 1, 2: this lane's position in the chunk: column, row * W
        for (uint32_t off = 1024u * wave; off < bytes; off += 4096u) {
            auto *lds = (__attribute__((address_space(3))) void *)&dst[off >> 4];
            if constexpr (MODE == 0) {
                // the whole byte offset goes into the VGPR offset: that is the field the frame-sized range check surely covers
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, (int)(start + off + 16u * lane), 0, 0, VDF_STREAM_AUX);
            } else {
                // (MODE 2: the dword below the row's first byte; an uncropped chunk starts on a multiple of 16 bytes, a box need not)
                const uint32_t at = MODE != 2 ? start + ro : ROWCROP ? (start + ro) & ~3u : start + (ro & ~3u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, (int)(at + x), 0, 0, VDF_STREAM_AUX);
                x += step_x;
                ro += step_rows * W;
                if (x >= Wp) { x -= Wp; ro += W; }
            }
        }
    };
    // vertical fragments of the 64-row group that holds this wave's block of chunk c
    auto load_av = [&](uint32_t c, v4i &h, v4i &l, const RowGeo &q) __attribute__((always_inline)) {
        uint32_t rg = (c * nb + wave) >> 2;
        rg = rg < (uint32_t)q.n_rg ? rg : (uint32_t)q.n_rg - 1u;
        h = q.av[(rg * 2 + 0) * 64 + lane];
        l = q.av[(rg * 2 + 1) * 64 + lane];
    };

    uint32_t F = blockIdx.x, c = 0;  // the chunk whose products run next
    v4i acc_vh = zero4, acc_vl = zero4, pend_vh = zero4, pend_vl = zero4;
    bool out_pending = false;
    uint32_t out_F = 0;
    auto write_pending = [&]() __attribute__((always_inline)) {  // after a barrier: wave 0 adds the partial sums of the frame that ended
        if (out_pending && wave == 0) {
            v4i vh = pend_vh, vl = pend_vl;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                vl[r] += bias_v[r];
#pragma unroll
                for (int w = 0; w < 3; w++) vl[r] += s_part[w][lane][r];
            }
            const uint32_t px = finalize4(vh, vl, pend_prec) ^ 0x80808080u;
            uint8_t *dst = small + (size_t)out_F * 256;
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + r16] = (uint8_t)(px >> (8 * r));
        }
        out_pending = false;
    };
    auto step = [&](const uint4 *cur, uint4 *nxt, const v4i &avh, const v4i &avl, v4i &avh_n, v4i &avl_n) __attribute__((always_inline)) {
        // Each wave's own DMA instructions of the chunk must have landed BEFORE it arrives at the barrier (the rows of a block come from all
        // four waves).  The fence of __syncthreads() used to bring that vmcnt(0) along, but it is the compiler's to drop: in the ROWCROP
        // instantiation the barrier at the loop header came out without it (seen in the ISA; wrong hashes for a few clips per launch).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk (F, c) has landed in `cur` (vmcnt) and every wave is done with `nxt`
        const uint32_t n_chunks = (geo.h + rpc - 1) / rpc;
        const bool last = c + 1 == n_chunks;
        uint32_t Fn = F, cn = c + 1;
        if (last) { cn = 0; Fn = F + gridDim.x; }
        const RowGeo &gq = ROWCROP && last ? geo_n : geo;  // of the chunk to fetch
        if (Fn < n_frames) issue_dma(Fn, cn, nxt, gq);  // first: with one chunk in flight per workgroup its issue time is on the critical path
        write_pending();
        if (Fn < n_frames) load_av(cn, avh_n, avl_n, gq);  // consumed in the next step, behind the barrier's vmcnt wait
        const uint32_t rows = min(rpc, geo.h - c * rpc);
        if (16u * wave < rows) {
            v4i ah = zero4, al = {bias_h, bias_h, bias_h, bias_h};
            const uint32_t row = 16u * wave + r16;  // in the chunk; chunks start on multiples of 16 rows, so (row * W) & 3 is the frame row's
            const uint8_t *base = reinterpret_cast<const uint8_t *>(cur) + row * Wp + 16u * g;
            const uint32_t shift = MODE == 2 ? ((ROWCROP ? geo.y0 * W : 0u) + row * W) & 3u : 0u;
            auto tile = [&](int kt) __attribute__((always_inline)) {
                const uint4 p = *reinterpret_cast<const uint4 *>(base + 64 * kt);
                v4i a = {(int)p.x, (int)p.y, (int)p.z, (int)p.w};
                if constexpr (MODE == 2) {
                    const uint32_t nx = *reinterpret_cast<const uint32_t *>(base + 64 * kt + 16);
                    a[0] = (int)__builtin_amdgcn_alignbyte(p.y, p.x, shift);
                    a[1] = (int)__builtin_amdgcn_alignbyte(p.z, p.y, shift);
                    a[2] = (int)__builtin_amdgcn_alignbyte(p.w, p.z, shift);
                    a[3] = (int)__builtin_amdgcn_alignbyte(nx, p.w, shift);
                }
                a = a ^ x80;
                const uint4 th = s_tab[(kt * 2 + 0) * 64 + lane], tl = s_tab[(kt * 2 + 1) * 64 + lane];
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, (v4i){(int)th.x, (int)th.y, (int)th.z, (int)th.w}, ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, (v4i){(int)tl.x, (int)tl.y, (int)tl.z, (int)tl.w}, al, 0, 0, 0);
            };
            int kt = 0;
            for (; kt + 3 < T.n_kt; kt += 4) { tile(kt); tile(kt + 1); tile(kt + 2); tile(kt + 3); }  // four tiles' LDS reads per wait
            for (; kt < T.n_kt; kt++) tile(kt);
            const int val = (int)finalize4(ah, al, T.prec_h);
            const uint32_t mb = (c * nb + wave) & 3u;  // block of the 64-row group: bytes 4 mb .. 4 mb + 3 of the operand
            v4i b;
#pragma unroll
            for (int m = 0; m < 4; m++) b[m] = mb == (uint32_t)m ? val : 0;
            acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(avh, b, acc_vh, 0, 0, 0);
            acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(avl, b, acc_vl, 0, 0, 0);
        }
        if (last) {  // frame complete: partial sums to LDS, the result is written after the next barrier
            // A frame of ONE chunk ends in the step that began with wave 0 reading the PREVIOUS frame's partial sums (write_pending):
            // nothing orders that read before the writes below - a wave without a block in a short chunk gets here at once (seen, round 5:
            // 1 - 2 clips in 30 000 of 64 x 48 through this kernel by force; in the product path only a letterbox box of at most one chunk
            // can get here).  Frames of two chunks and more have the next step's barrier in between.  LDS-only: the DMA stays in flight.
#ifndef VDF_ABL_NO_ONE_CHUNK_BARRIER  // (ablation build: the library before ce37e43, for showing that the sweeps and the ISA check see the race)
            if (c == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) s_part[wave - 1][lane][r] = (acc_vh[r] << 8) + acc_vl[r];
                // The barrier that publishes these words is the one at the top of the next step, across the loop's back
                // edge, and the compiler emitted it with a vmcnt wait only (seen in the ISA; wave 0 then read stale partial
                // sums once in ~20 launches when two workgroups shared a CU): retire the LDS writes here.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            pend_vh = acc_vh; pend_vl = acc_vl;
            acc_vh = zero4; acc_vl = zero4;
            out_pending = true;
            out_F = ROWCROP ? geo.src * 16u + (F & 15u) : F;
            if constexpr (ROWCROP) {  // the finished clip's bias and precision; the next barrier's vmcnt wait covers the loads
                pend_prec = geo.prec_v;
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) bias_v[r] = geo.bias_v[4 * g + r];
                }
                geo = geo_n;
                if (Fn + gridDim.x < n_frames) geo_n = row_geo_of(clips, tables, Fn + gridDim.x);
            }
        }
        F = Fn; c = cn;
    };
    v4i av0h = zero4, av0l = zero4, av1h = zero4, av1l = zero4;
    if constexpr (ROWCROP) {
        if (F < n_frames) geo = row_geo_of(clips, tables, F);
        geo_n = geo;
        if (F + gridDim.x < n_frames) geo_n = row_geo_of(clips, tables, F + gridDim.x);
    }
    if (F < n_frames) {
        load_av(0, av0h, av0l, geo);
        issue_dma(F, 0, s_px0, geo);
    }
    while (F < n_frames) {
        step(s_px0, s_px1, av0h, av0l, av1h, av1l);
        if (!(F < n_frames)) break;
        step(s_px1, s_px0, av1h, av1l, av0h, av0l);
    }
    __syncthreads();
    write_pending();
}

// ---- linear-stream form, one 16-row block stream PER WAVE (round 3; wide band-class frames such as 1920 x 1080) -------------
// The kernel above keeps ONE chunk per workgroup in flight.  For frames whose chunk holds only two 16-row blocks (1440..1984
// columns: 32 rows = 60 KB) that is the bandwidth-delay product and nothing more - two of the four waves idle through every
// product phase, and every bubble (barrier, DMA issue) shows: 1920 x 1080 stayed at 6.0-6.2 TB/s while 1280 x 720 (three blocks per
// chunk) reached 6.4-6.6 with the same non-temporal loads.  Here every wave owns one 30 KB buffer and walks the frame's blocks
// wave, wave + 4, ...: compute block i (30 K tiles from its own buffer), then DMA block i + 4 over it, wait for it alone
// (s_waitcnt vmcnt(0): no workgroup barrier), compute ...  Up to four blocks (120 KB) are in flight per CU, the waves drift out of
// phase by themselves, and the only barrier left is the LDS-only one at the end of a frame where the four vertical partial
// sums meet (double-buffered by frame parity, so wave 0 adds and writes frame F while the others already stream frame F + G).
// Same products in the same order per output as the other kernels (exact integers): bit-identical.  W % 16 == 0, rows at the
// frame's own pitch (MODE 0 addressing), table in band form.
// MODE as in the kernel above: 0 = rows at the frame's own pitch (W % 16 == 0), 1 / 2 = rows re-pitched by the DMA to Wp (an odd
// multiple of 16 bytes; 2 = row starts that are not dword-aligned: one more dword per operand read and a per-lane byte shift).
// ROWCROP: per-clip row ranges (see row_geo_of): the box's first row, block count and vertical table change from clip to clip; a wave
// without a block in a short box still issues its first block of the next frame.  A launch may also cover boxes with side bars that
// share their column range (x0, width): W stays the FRAME's pitch (every global address is in frame coordinates), x0 shifts the rows'
// first byte, and the box's width is in the band table and in Wp (MODE 1 / 2: the DMA gathers the box's bytes of each row).
// NW waves per workgroup, each with its own block buffer: four for the widest frames (30 KB blocks), more for narrower ones, so that the
// blocks in flight per CU stay near 120 KB whatever the width.
template <int NW, int BUF_BYTES, int TAB_BYTES, int MODE, bool ROWCROP = false>
__global__ __launch_bounds__(64 * NW) void resize_mfma_frame_wavestream_kernel(const uint8_t *__restrict__ frames, uint32_t W,
                                                                           uint32_t H, size_t frame_stride,
                                                                           size_t clip_stride, uint32_t n_frames,
                                                                           MfmaResizeTables T, uint32_t Wp, uint8_t *__restrict__ small,
                                                                           const CropStreamClip *__restrict__ clips_g = nullptr,
                                                                           const CropStreamTable *__restrict__ tables_g = nullptr, uint32_t x0 = 0)
{
    __shared__ __attribute__((aligned(16))) uint4 s_tab[TAB_BYTES / 16];
    __shared__ __attribute__((aligned(16))) uint4 s_pxw[NW][BUF_BYTES / 16];
    __shared__ int32_t s_part[2][NW - 1][64][4];  // [frame parity][wave 1..]: 256 hi + lo
    const uint32_t tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t frame_bytes = W * H;
    const_clip_ptr clips = (const_clip_ptr)(uintptr_t)clips_g;
    const_table_ptr tables = (const_table_ptr)(uintptr_t)tables_g;
    RowGeo geo = {0u, H, 0u, T.n_rg, T.prec_v, (global_v4i)T.av, (global_i32)T.bias_v}, geo_n = geo;  // of frame F, of the workgroup's next frame
    const int32_t bias_h = T.bias_h[r16];
    v4i bias_v = {0, 0, 0, 0};
    if constexpr (!ROWCROP) {
#pragma unroll
        for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
    }
    const uint32_t tab_vecs = (uint32_t)T.band_stride;  // 16 outputs x stride / 16 bytes
    for (uint32_t i = tid; i < tab_vecs; i += 64u * NW) {
        const v4i v = T.bh[i];
        s_tab[i] = uint4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
    }
    if (tid < 8) s_tab[tab_vecs + tid] = uint4{0, 0, 0, 0};
    const uint32_t band_zero = 16u * tab_vecs;
    const int32_t band_lo = T.band_meta[r16];
    const uint32_t band_nt = (uint32_t)T.band_meta[16 + r16];
    const uint32_t band_base = r16 * (uint32_t)T.band_stride + 16u * g;
    const v4i zero4 = {0, 0, 0, 0};
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    uint4 *my = s_pxw[wave];
    __syncthreads();  // the table is in place

    // MODE 1, 2: where this lane's first DMA instruction of a block lands (LDS position 16 lane = row * Wp + x) and how far an
    // instruction (1024 bytes of LDS further) moves it
    uint32_t lane_x0 = 0, lane_ro0 = 0;
    const uint32_t step_rows = MODE ? 1024u / Wp : 0u, step_x = MODE ? 1024u - step_rows * Wp : 0u;
    if constexpr (MODE != 0) {
        const uint32_t P0 = 16u * lane, row0 = P0 / Wp;
        lane_x0 = P0 - row0 * Wp;
        lane_ro0 = row0 * W;
    }
    const uint32_t shift0 = MODE == 2 ? (r16 * W) & 3u : 0u;  // blocks start on multiples of 16 rows: (row * W) & 3 is the frame row's
    auto issue_dma = [&](uint32_t F, uint32_t b, const RowGeo &q) __attribute__((always_inline)) {
        const uint8_t *src = frames + (size_t)(ROWCROP ? q.src : F >> 4) * clip_stride + (size_t)(F & 15u) * frame_stride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, frame_bytes, 0x00020000);
        const uint32_t start = (q.y0 + 16u * b) * W + (ROWCROP ? x0 : 0u), bytes = min(16u, q.h - 16u * b) * Wp;
        uint32_t x = lane_x0, ro = lane_ro0;
        for (uint32_t off = 0; off < bytes; off += 1024u) {
            auto *lds = (__attribute__((address_space(3))) void *)&my[off >> 4];
            if constexpr (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, (int)(start + off + 16u * lane), 0, 0, VDF_STREAM_AUX);
            } else {
                const uint32_t at = MODE != 2 ? start + ro : ROWCROP ? (start + ro) & ~3u : start + (ro & ~3u);  // (a box need not start on a dword)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, (int)(at + x), 0, 0, VDF_STREAM_AUX);
                x += step_x;
                ro += step_rows * W;
                if (x >= Wp) { x -= Wp; ro += W; }
            }
        }
    };
    uint32_t parity = 0;
    uint32_t F = blockIdx.x;
    if constexpr (ROWCROP) {
        if (F < n_frames) geo = row_geo_of(clips, tables, F);
    }
    if (F < n_frames && 16u * wave < geo.h) issue_dma(F, wave, geo);
    for (; F < n_frames; F += gridDim.x) {
        v4i acc_vh = zero4, acc_vl = zero4;
        const uint32_t Fn = F + gridDim.x;
        const uint32_t n_blk = (geo.h + 15u) / 16u;
        uint32_t shift = shift0;
        bool next_issued = false;
        if constexpr (ROWCROP) {
            if (Fn < n_frames) geo_n = row_geo_of(clips, tables, Fn);
            if (wave == 0) {  // consumed at the end of the frame
#pragma unroll
                for (int r = 0; r < 4; r++) bias_v[r] = geo.bias_v[4 * g + r];
            }
            if constexpr (MODE == 2) shift = (geo.y0 * W + x0 + r16 * W) & 3u;
        }
        for (uint32_t b = wave; b < n_blk; b += NW) {
            // vertical fragments of this block's 64-row group (global loads: issued before the wait, consumed after the products)
            const uint32_t rg = min(b >> 2, (uint32_t)geo.n_rg - 1u);
            const v4i avh = geo.av[(rg * 2 + 0) * 64 + lane], avl = geo.av[(rg * 2 + 1) * 64 + lane];
            // This wave's own block has landed once the two fragments have: they were requested AFTER its DMA and VMEM returns in
            // order.  An empty asm that reads them makes the COMPILER place the vmcnt wait here and know the fragments are in -
            // with a hand-written s_waitcnt ALONE it kept its own wait in front of the vertical products below, i.e. behind the NEXT
            // block's DMA, and the next fragments' load latency was exposed every block.  The explicit vmcnt(0) inside the same asm
            // costs nothing (nothing else of this wave is in flight here) and no longer leaves the block's arrival to the order the
            // compiler happens to keep between the DMA and the two loads (tools/check_isa_barriers.py looks for it in every instantiation).
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(avh), "v"(avl) : "memory");
            v4i ah = zero4, al = {bias_h, bias_h, bias_h, bias_h};
            const uint8_t *base = reinterpret_cast<const uint8_t *>(my) + r16 * Wp + 16u * g;
            auto tile = [&](int kt) __attribute__((always_inline)) {
                const uint4 p = *reinterpret_cast<const uint4 *>(base + 64 * kt);
                v4i a = {(int)p.x, (int)p.y, (int)p.z, (int)p.w};
                if constexpr (MODE == 2) {
                    const uint32_t nx = *reinterpret_cast<const uint32_t *>(base + 64 * kt + 16);
                    a[0] = (int)__builtin_amdgcn_alignbyte(p.y, p.x, shift);
                    a[1] = (int)__builtin_amdgcn_alignbyte(p.z, p.y, shift);
                    a[2] = (int)__builtin_amdgcn_alignbyte(p.w, p.z, shift);
                    a[3] = (int)__builtin_amdgcn_alignbyte(nx, p.w, shift);
                }
                a = a ^ x80;
                const uint32_t j = (uint32_t)(kt - band_lo);
                const uint8_t *q = reinterpret_cast<const uint8_t *>(s_tab) + (j < band_nt ? band_base + j * 128u : band_zero);
                const uint4 th = *reinterpret_cast<const uint4 *>(q), tl = *reinterpret_cast<const uint4 *>(q + 64);
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, (v4i){(int)th.x, (int)th.y, (int)th.z, (int)th.w}, ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, (v4i){(int)tl.x, (int)tl.y, (int)tl.z, (int)tl.w}, al, 0, 0, 0);
            };
            int kt = 0;
            for (; kt + 3 < T.n_kt; kt += 4) { tile(kt); tile(kt + 1); tile(kt + 2); tile(kt + 3); }
            for (; kt < T.n_kt; kt++) tile(kt);
            // rows past the frame's end hold the previous block's bytes: their vertical coefficients are zero (as in every kernel)
            const int val = (int)finalize4(ah, al, T.prec_h);
            // the buffer is free: the products above consumed every LDS read of it.  Next block of this frame, or this wave's
            // first block of the next frame - before the frame-end barrier, so the stream never drains
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (b + NW < n_blk) issue_dma(F, b + NW, geo);
            else if (Fn < n_frames && 16u * wave < geo_n.h) { issue_dma(Fn, wave, geo_n); next_issued = true; }
            const uint32_t mb = b & 3u;
            v4i bb;
#pragma unroll
            for (int m = 0; m < 4; m++) bb[m] = mb == (uint32_t)m ? val : 0;
            acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(avh, bb, acc_vh, 0, 0, 0);
            acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(avl, bb, acc_vl, 0, 0, 0);
        }
        if constexpr (ROWCROP) {  // a wave without a block in this (short) box still owes its first block of the next frame
            if (!next_issued && Fn < n_frames && 16u * wave < geo_n.h) issue_dma(Fn, wave, geo_n);
        }
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) s_part[parity][wave - 1][lane][r] = (acc_vh[r] << 8) + acc_vl[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: the DMA of the next frame's first blocks stays in flight
        if (wave == 0) {
            v4i vl = acc_vl;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                vl[r] += bias_v[r];
#pragma unroll
                for (int w = 0; w < NW - 1; w++) vl[r] += s_part[parity][w][lane][r];
            }
            const uint32_t px = finalize4(acc_vh, vl, geo.prec_v) ^ 0x80808080u;
            uint8_t *dst = small + (size_t)(ROWCROP ? geo.src * 16u + (F & 15u) : F) * 256;
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + r16] = (uint8_t)(px >> (8 * r));
        }
        parity ^= 1u;  // wave 0 reads parity p while the others may already write p ^ 1; p is written again only after the next barrier
        if constexpr (ROWCROP) geo = geo_n;
    }
}

static void launch_stream_mode(uint32_t grid, hipStream_t stream, const uint8_t *frames, uint32_t w, uint32_t h,
                               size_t frame_stride, size_t clip_stride, uint32_t n_frames, const MfmaResizeTables &T,
                               uint32_t nb, uint8_t *small, const CropStreamClip *clips, const CropStreamTable *tables)
{
    const uint32_t wp = stream_pitch(w);
    const int mode = wp == w ? 0 : w % 4 == 0 ? 1 : 2;
#define VDF_CS_LAUNCH(M, RC)                                                                                                               \
    hipLaunchKernelGGL((resize_mfma_frame_stream_kernel<kStreamBufS, kStreamTabS, M, RC>), dim3(grid), dim3(256), 0, stream, frames, w, h, \
                       frame_stride, clip_stride, n_frames, T, nb, wp, small, clips, tables)
    if (clips) {  // per-clip row ranges
        if (mode == 0) VDF_CS_LAUNCH(0, true);
        else if (mode == 1) VDF_CS_LAUNCH(1, true);
        else VDF_CS_LAUNCH(2, true);
    } else {
        if (mode == 0) VDF_CS_LAUNCH(0, false);
        else if (mode == 1) VDF_CS_LAUNCH(1, false);
        else VDF_CS_LAUNCH(2, false);
    }
#undef VDF_CS_LAUNCH
}

// per-wave block streams: NW by the (box) width (resize_wavestream_waves), MODE by the pitch and the box's first column, ROWCROP when clips
// carry row ranges
template <int NW, int BUF, int TAB>
static void launch_wavestream_nw(uint32_t grid, hipStream_t stream, const uint8_t *frames, uint32_t w, uint32_t h, size_t frame_stride,
                                 size_t clip_stride, uint32_t n_frames, const MfmaResizeTables &T, uint8_t *small,
                                 const CropStreamClip *clips, const CropStreamTable *tables, uint32_t wp, int mode, uint32_t x0)
{
#define VDF_WS_LAUNCH(M, RC)                                                                                                              \
    hipLaunchKernelGGL((resize_mfma_frame_wavestream_kernel<NW, BUF, TAB, M, RC>), dim3(grid), dim3(64 * NW), 0, stream, frames, w, h, \
                       frame_stride, clip_stride, n_frames, T, wp, small, clips, tables, x0)
    if (clips) {
        if (mode == 0) VDF_WS_LAUNCH(0, true);
        else if (mode == 1) VDF_WS_LAUNCH(1, true);
        else VDF_WS_LAUNCH(2, true);
    } else {
        if (mode == 0) VDF_WS_LAUNCH(0, false);
        else if (mode == 1) VDF_WS_LAUNCH(1, false);
        else VDF_WS_LAUNCH(2, false);
    }
#undef VDF_WS_LAUNCH
}

// w = the frames' pitch; box_w / x0 = the column range every clip of the launch keeps (box_w == w, x0 == 0: whole rows)
static hipError_t launch_wavestream(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                    const MfmaResizeArgs &a, const CropStreamClip *clips, const CropStreamTable *tables, uint8_t *small,
                                    hipStream_t stream, uint32_t box_w, uint32_t x0)
{
    const int nw = resize_wavestream_waves_box(w, x0, box_w, a.wavestream_knob);
    if (n_clips * 16 > 0xFFFFFFFFull || (uint64_t)w * h >= (1ull << 31) || !a.band_meta || nw == 0 || (box_w != w && !clips) || (uint64_t)x0 + box_w > w)
        return hipErrorInvalidValue;
    if (!resize_wavestream_table_fits(nw, a.band_stride)) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t n_frames = (uint32_t)(n_clips * 16), grid = std::min<uint32_t>(n_frames, (uint32_t)cus);
    const MfmaResizeTables T = make_tables(a);
    int mode = 0;
    const uint32_t wp = box_stream_pitch(w, x0, box_w, &mode);
    if (nw == 3) launch_wavestream_nw<3, kWaveStreamBuf3, kWaveStreamTabBytes>(grid, stream, frames, w, h, frame_stride, clip_stride, n_frames, T, small, clips, tables, wp, mode, x0);
    else if (nw == 4) launch_wavestream_nw<4, kWaveStreamBuf, kWaveStreamTabBytes>(grid, stream, frames, w, h, frame_stride, clip_stride, n_frames, T, small, clips, tables, wp, mode, x0);
    else if (nw == 5) launch_wavestream_nw<5, kWaveStreamBuf5, kWaveStreamTabMid>(grid, stream, frames, w, h, frame_stride, clip_stride, n_frames, T, small, clips, tables, wp, mode, x0);
    else if (nw == 6) launch_wavestream_nw<6, kWaveStreamBuf6, kWaveStreamTabSmall>(grid, stream, frames, w, h, frame_stride, clip_stride, n_frames, T, small, clips, tables, wp, mode, x0);
    else launch_wavestream_nw<8, kWaveStreamBuf8, kWaveStreamTabSmall>(grid, stream, frames, w, h, frame_stride, clip_stride, n_frames, T, small, clips, tables, wp, mode, x0);
    return hipGetLastError();
}

hipError_t launch_resize_mfma_box_wavestream(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                             size_t clip_stride, const MfmaResizeArgs &a, uint32_t x0, uint32_t box_w, const CropStreamClip *clips,
                                             const CropStreamTable *tables, uint8_t *small, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    if (!clips || !tables) return hipErrorInvalidValue;
    return launch_wavestream(frames, n_clips, w, h, frame_stride, clip_stride, a, clips, tables, small, stream, box_w, x0);
}

hipError_t launch_resize_mfma_frames_stream(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h,
                                            size_t frame_stride, size_t clip_stride, const MfmaResizeArgs &a,
                                            uint8_t *small, hipStream_t stream, const CropStreamClip *clips,
                                            const CropStreamTable *tables)
{
    if (n_clips == 0) return hipSuccess;
    uint32_t nb = 0;
    const int cls = stream_class(w, &nb);
    if (n_clips * 16 > 0xFFFFFFFFull) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t n_frames = (uint32_t)(n_clips * 16);
    if (resize_stream_wants_band(w, a.wavestream_knob) != (a.band_meta != nullptr)) return hipErrorInvalidValue;  // the caller picks the table form by resize_stream_wants_band
    if (resize_wavestream_applies(w, a.wavestream_knob))
        return launch_wavestream(frames, n_clips, w, h, frame_stride, clip_stride, a, clips, tables, small, stream, w, 0);
    if (cls != 1 || a.band_meta) return hipErrorInvalidValue;  // the chunk form serves the S class only (frames up to 512 wide, two workgroups per CU)
    launch_stream_mode(std::min<uint32_t>(n_frames, (uint32_t)cus * 2u), stream, frames, w, h, frame_stride, clip_stride, n_frames, make_tables(a),
                       nb, small, clips, tables);
    return hipGetLastError();
}

// ---- wide frames, linear-stream form with the K tiles split over the waves -----------------------------------
// The stream kernel above keeps the horizontal table in LDS (2 KB per K tile, or its band form) and gives each wave its own
// 16-row block.  For wide frames that leaves too little LDS for the chunks in flight (1024 / 1536 wide: 48 KB chunks) or does
// not fit at all (wider than 1984).  Here the table lives in REGISTERS: all four waves work on the same 16-row block, wave w
// taking K tiles w, w + 4, ... - always the same tiles, so its B fragments (2 x 4 VGPRs per tile, at most 16 tiles = 128
// VGPRs for 4096 columns) are loaded once per launch - and the four partial sums of a block meet in LDS (exact integers,
// hi and lo folded into one i32 as in finalize4) behind an LDS-only barrier that does not wait for the DMA in flight.
// The wave that sums a block (block index & 3) also owns its byte column of the vertical operand and does the vertical
// product, so the vertical partial sums are spread over the waves as before.  LDS holds nothing but the two chunk
// buffers (75 KB each: 16 rows of 4096 columns, 32 of 2048, 48 of 1536, 64 of 1024), rows re-pitched to an odd multiple of
// 16 bytes by the DMA (at the frame's own pitch the 16 rows of a block would share one bank group: 3840 = 240 x 16).
template <int MAXT, bool ROWCROP = false>
__global__ __launch_bounds__(256) void resize_mfma_frame_ksplit_kernel(const uint8_t *__restrict__ frames, uint32_t W,
                                                                       uint32_t H, size_t frame_stride,
                                                                       size_t clip_stride, uint32_t n_frames,
                                                                       MfmaResizeTables T, uint32_t nb, uint32_t Wp,
                                                                       uint8_t *__restrict__ small,
                                                                       const CropStreamClip *__restrict__ clips_g = nullptr,
                                                                       const CropStreamTable *__restrict__ tables_g = nullptr)
{
    constexpr int kBuf = kKsplitBuf;
    __shared__ __attribute__((aligned(16))) uint4 s_px0[kBuf / 16];
    __shared__ __attribute__((aligned(16))) uint4 s_px1[kBuf / 16];
    __shared__ __attribute__((aligned(16))) v4i s_red[2][3][64];  // partial sums of the three waves that do not own the block, by block parity
    __shared__ int32_t s_part[3][64][4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t rpc = 16u * nb, frame_bytes = W * H;
    // ROWCROP: as in resize_mfma_frame_stream_kernel
    const_clip_ptr clips = (const_clip_ptr)(uintptr_t)clips_g;
    const_table_ptr tables = (const_table_ptr)(uintptr_t)tables_g;
    RowGeo geo = {0u, H, 0u, T.n_rg, T.prec_v, (global_v4i)T.av, (global_i32)T.bias_v}, geo_n = geo;
    int32_t pend_prec = T.prec_v;
    const v4i zero4 = {0, 0, 0, 0};
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
    // this wave's K tiles, for the whole launch
    v4i bt[MAXT][2];
#pragma unroll
    for (int i = 0; i < MAXT; i++) {
        const int kt = (int)wave + 4 * i;
        bt[i][0] = zero4; bt[i][1] = zero4;
        if (kt < T.n_kt) { bt[i][0] = T.bh[(kt * 2 + 0) * 64 + lane]; bt[i][1] = T.bh[(kt * 2 + 1) * 64 + lane]; }
    }
    const int32_t bias_h = T.bias_h[r16];
    v4i bias_v = {0, 0, 0, 0};
    if constexpr (!ROWCROP) {
#pragma unroll
        for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
    }
    // gather DMA: where this lane's first instruction of a chunk lands (LDS position 1024 wave + 16 lane = row * Wp + x)
    const uint32_t step_rows = 4096u / Wp, step_x = 4096u - step_rows * Wp;
    uint32_t lane_x0, lane_ro0;
    {
        const uint32_t P0 = 1024u * wave + 16u * lane, row0 = P0 / Wp;
        lane_x0 = P0 - row0 * Wp;
        lane_ro0 = row0 * W;
    }
    auto issue_dma = [&](uint32_t F, uint32_t c, uint4 *dst, const RowGeo &q) __attribute__((always_inline)) {
        const uint8_t *src = frames + (size_t)(ROWCROP ? q.src : F >> 4) * clip_stride + (size_t)(F & 15u) * frame_stride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, frame_bytes, 0x00020000);
        const uint32_t start = (q.y0 + c * rpc) * W, rows = min(rpc, q.h - c * rpc), bytes = rows * Wp;
        uint32_t x = lane_x0, ro = lane_ro0;
        for (uint32_t off = 1024u * wave; off < bytes; off += 4096u) {
            auto *lds = (__attribute__((address_space(3))) void *)&dst[off >> 4];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, (int)(start + ro + x), 0, 0, VDF_STREAM_AUX);
            x += step_x;
            ro += step_rows * W;
            if (x >= Wp) { x -= Wp; ro += W; }
        }
    };
    // the block of chunk c this wave owns (block index & 3 == wave), if the chunk has one: its vertical fragments
    auto load_av = [&](uint32_t c, v4i &h, v4i &l, const RowGeo &q) __attribute__((always_inline)) {
        const uint32_t j = (wave - c * nb) & 3u;  // block j of the chunk has index c * nb + j
        uint32_t rg = (c * nb + j) >> 2;
        rg = rg < (uint32_t)q.n_rg ? rg : (uint32_t)q.n_rg - 1u;
        h = q.av[(rg * 2 + 0) * 64 + lane];
        l = q.av[(rg * 2 + 1) * 64 + lane];
    };

    uint32_t F = blockIdx.x, c = 0;
    v4i acc_vh = zero4, acc_vl = zero4, pend_vh = zero4, pend_vl = zero4;
    bool out_pending = false;
    uint32_t out_F = 0;
    auto write_pending = [&]() __attribute__((always_inline)) {
        if (out_pending && wave == 0) {
            v4i vh = pend_vh, vl = pend_vl;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                vl[r] += bias_v[r];
#pragma unroll
                for (int w = 0; w < 3; w++) vl[r] += s_part[w][lane][r];
            }
            const uint32_t px = finalize4(vh, vl, pend_prec) ^ 0x80808080u;
            uint8_t *dst = small + (size_t)out_F * 256;
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + r16] = (uint8_t)(px >> (8 * r));
        }
        out_pending = false;
    };
    auto step = [&](const uint4 *cur, uint4 *nxt, const v4i &avh, const v4i &avl, v4i &avh_n, v4i &avl_n) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // see resize_mfma_frame_stream_kernel
        __syncthreads();  // chunk (F, c) has landed in `cur` (vmcnt) and every wave is done with `nxt`
        const uint32_t n_chunks = (geo.h + rpc - 1) / rpc;
        const bool last = c + 1 == n_chunks;
        uint32_t Fn = F, cn = c + 1;
        if (last) { cn = 0; Fn = F + gridDim.x; }
        const RowGeo &gq = ROWCROP && last ? geo_n : geo;  // of the chunk to fetch
        if (Fn < n_frames) issue_dma(Fn, cn, nxt, gq);
        write_pending();
        if (Fn < n_frames) load_av(cn, avh_n, avl_n, gq);
        const uint32_t rows = min(rpc, geo.h - c * rpc), n_blocks = (rows + 15u) >> 4;
        for (uint32_t j = 0; j < n_blocks; j++) {
            const uint32_t owner = (c * nb + j) & 3u;
            v4i ah = zero4, al = zero4;
            const uint8_t *base = reinterpret_cast<const uint8_t *>(cur) + (16u * j + r16) * Wp + 64u * wave + 16u * g;
#pragma unroll
            for (int i = 0; i < MAXT; i++) {
                if ((int)wave + 4 * i < T.n_kt) {  // wave-uniform
                    const uint4 p = *reinterpret_cast<const uint4 *>(base + 256 * i);
                    const v4i a = (v4i){(int)p.x, (int)p.y, (int)p.z, (int)p.w} ^ x80;
                    ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bt[i][0], ah, 0, 0, 0);
                    al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bt[i][1], al, 0, 0, 0);
                }
            }
            v4i part;
#pragma unroll
            for (int r = 0; r < 4; r++) part[r] = (ah[r] << 8) + al[r];
            if (wave != owner) s_red[j & 1][(wave - owner - 1u) & 3u][lane] = part;
            // LDS-only barrier: the partial sums are LDS writes; a __syncthreads() here would also wait for the DMA in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (wave == owner) {
#pragma unroll
                for (int w = 0; w < 3; w++) {
                    const v4i o = s_red[j & 1][w][lane];
#pragma unroll
                    for (int r = 0; r < 4; r++) part[r] += o[r];
                }
                const v4i hi0 = zero4;
#pragma unroll
                for (int r = 0; r < 4; r++) part[r] += bias_h;
                const int val = (int)finalize4(hi0, part, T.prec_h);
                v4i b;
#pragma unroll
                for (int m = 0; m < 4; m++) b[m] = owner == (uint32_t)m ? val : 0;
                acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(avh, b, acc_vh, 0, 0, 0);
                acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(avl, b, acc_vl, 0, 0, 0);
            }
        }
        if (last) {
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) s_part[wave - 1][lane][r] = (acc_vh[r] << 8) + acc_vl[r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // see resize_mfma_frame_stream_kernel
            }
            pend_vh = acc_vh; pend_vl = acc_vl;
            acc_vh = zero4; acc_vl = zero4;
            out_pending = true;
            out_F = ROWCROP ? geo.src * 16u + (F & 15u) : F;
            if constexpr (ROWCROP) {
                pend_prec = geo.prec_v;
                if (wave == 0) {
#pragma unroll
                    for (int r = 0; r < 4; r++) bias_v[r] = geo.bias_v[4 * g + r];
                }
                geo = geo_n;
                if (Fn + gridDim.x < n_frames) geo_n = row_geo_of(clips, tables, Fn + gridDim.x);
            }
        }
        F = Fn; c = cn;
    };
    v4i av0h = zero4, av0l = zero4, av1h = zero4, av1l = zero4;
    if constexpr (ROWCROP) {
        if (F < n_frames) geo = row_geo_of(clips, tables, F);
        geo_n = geo;
        if (F + gridDim.x < n_frames) geo_n = row_geo_of(clips, tables, F + gridDim.x);
    }
    if (F < n_frames) {
        load_av(0, av0h, av0l, geo);
        issue_dma(F, 0, s_px0, geo);
    }
    while (F < n_frames) {
        step(s_px0, s_px1, av0h, av0l, av1h, av1l);
        if (!(F < n_frames)) break;
        step(s_px1, s_px0, av1h, av1l, av0h, av0l);
    }
    __syncthreads();
    write_pending();
}

hipError_t launch_resize_mfma_frames_ksplit(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h,
                                            size_t frame_stride, size_t clip_stride, const MfmaResizeArgs &a,
                                            uint8_t *small, hipStream_t stream, const CropStreamClip *clips,
                                            const CropStreamTable *tables)
{
    if (n_clips == 0) return hipSuccess;
    uint32_t wp = 0;
    const uint32_t nb = ksplit_geometry(w, &wp);
    if (n_clips * 16 > 0xFFFFFFFFull || nb == 0 || a.n_kt > 64 || a.band_meta) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t n_frames = (uint32_t)(n_clips * 16);
    const dim3 grid(std::min<uint32_t>(n_frames, (uint32_t)cus));
    if (clips) {  // per-clip row ranges
        if (a.n_kt <= 16)
            hipLaunchKernelGGL((resize_mfma_frame_ksplit_kernel<4, true>), grid, dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride,
                               n_frames, make_tables(a), nb, wp, small, clips, tables);
        else if (a.n_kt <= 32)
            hipLaunchKernelGGL((resize_mfma_frame_ksplit_kernel<8, true>), grid, dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride,
                               n_frames, make_tables(a), nb, wp, small, clips, tables);
        else
            hipLaunchKernelGGL((resize_mfma_frame_ksplit_kernel<16, true>), grid, dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride,
                               n_frames, make_tables(a), nb, wp, small, clips, tables);
        return hipGetLastError();
    }
    if (a.n_kt <= 16)
        hipLaunchKernelGGL((resize_mfma_frame_ksplit_kernel<4>), grid, dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride,
                           n_frames, make_tables(a), nb, wp, small);
    else if (a.n_kt <= 32)
        hipLaunchKernelGGL((resize_mfma_frame_ksplit_kernel<8>), grid, dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride,
                           n_frames, make_tables(a), nb, wp, small);
    else
        hipLaunchKernelGGL((resize_mfma_frame_ksplit_kernel<16>), grid, dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride,
                           n_frames, make_tables(a), nb, wp, small);
    return hipGetLastError();
}

// ---- cropped clips, linear-stream form ------------------------------------------------------------------------
// The letterbox crop box is read in place, as in resize_mfma_cropped_kernel, but through the stream kernel's machinery:
// rows x0 .. x0 + w of frame rows y0 .. y0 + h go to LDS by gather DMA at the clip's own conflict-free pitch (always the
// re-pitch + byte-shift form: a crop's row starts have any alignment), the horizontal table is the band form of the
// crop's width, and every per-frame quantity (box, pitch, chunk geometry, tables, precisions) comes from the clip's
// descriptor by scalar loads.  A workgroup reloads its LDS table only when the next frame's clip uses another one
// (clips of one source share their box); the biases and the vertical fragments are requested before the DMA of the
// chunk that needs them, like the vertical fragments above.

template <int BUF_BYTES, int TAB_TILES, bool SHIFT>
__global__ __launch_bounds__(256) void resize_mfma_cropped_stream_kernel(const uint8_t *__restrict__ frames, uint32_t pitch,
                                                                         uint32_t frame_bytes, size_t frame_stride,
                                                                         size_t clip_stride, uint32_t n_frames,
                                                                         const CropStreamClip *__restrict__ clips_g,
                                                                         const CropStreamTable *__restrict__ tables_g,
                                                                         uint8_t *__restrict__ small)
{
    __shared__ __attribute__((aligned(16))) uint4 s_tab[TAB_TILES * 2 * 64];
    __shared__ __attribute__((aligned(16))) uint4 s_px0[BUF_BYTES / 16];
    __shared__ __attribute__((aligned(16))) uint4 s_px1[BUF_BYTES / 16];
    __shared__ int32_t s_part[3][64][4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const_clip_ptr clips = (const_clip_ptr)(uintptr_t)clips_g;
    const_table_ptr tables = (const_table_ptr)(uintptr_t)tables_g;
    const v4i zero4 = {0, 0, 0, 0};
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};

    // what the per-chunk code needs of a frame, in SGPRs (scalar loads, fetched a frame ahead of its first DMA); the rest
    // (table and bias pointers) is read from the table entries where it is used, once per frame, off the critical path.
    // Kept small on purpose: with the pointers in here the two copies spilled and the scalar loads turned into VMEM loads.
    struct Geo {
        uint32_t x0, y0, h, wp, step_rows, step_x, nb, n_chunks, h_table, v_table, src;  // step_*: 4096 = step_rows * wp + step_x; src: the clip's index in the caller's batch
        const __attribute__((address_space(1))) v4i *av;
        int32_t n_kt, n_rg, prec_h, prec_v;
    };
    // Every value below is workgroup-uniform; the readfirstlane pins it to an SGPR.  Without that the compiler treated one of
    // the descriptor loads as a vector load, and from there the chunk counter, the table index and the branches on them all
    // became per-lane values (v_cmp + exec masks, table entries fetched through VGPR addresses, waits on the DMA).
    auto sgpr = [](uint32_t v) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // (the result is cast back to the GLOBAL address space: as a generic pointer the loads through it became flat loads, and the
    // compiler puts a vmcnt(0) in front of a flat load while an LDS-DMA is in flight - flat may read LDS)
    auto sgpr_ptr = [&](const void *p) __attribute__((always_inline)) {
        const uint64_t a = (uint64_t)(uintptr_t)p;
        return (const __attribute__((address_space(1))) void *)(uintptr_t)(((uint64_t)sgpr((uint32_t)(a >> 32)) << 32) | sgpr((uint32_t)a));
    };
    typedef const __attribute__((address_space(1))) v4i *global_v4i;
    typedef const __attribute__((address_space(1))) int32_t *global_i32;
    auto geo_of = [&](uint32_t F) __attribute__((always_inline)) {
        const uint32_t clip = sgpr(F >> 4);
        Geo q;
        q.x0 = sgpr(clips[clip].x0); q.y0 = sgpr(clips[clip].y0); q.h = sgpr(clips[clip].h); q.src = sgpr(clips[clip].src_clip);
        q.wp = sgpr(clips[clip].wp); q.nb = sgpr(clips[clip].nb);
        q.step_rows = sgpr(clips[clip].step_rows); q.step_x = sgpr(clips[clip].step_x);
        q.n_chunks = sgpr(clips[clip].n_chunks);
        q.h_table = sgpr(clips[clip].h_table);
        q.v_table = sgpr(clips[clip].v_table);
        q.n_kt = (int32_t)sgpr((uint32_t)tables[q.h_table].n_tiles);
        q.prec_h = (int32_t)sgpr((uint32_t)tables[q.h_table].precision);
        q.av = (global_v4i)sgpr_ptr(tables[q.v_table].operand);
        q.n_rg = (int32_t)sgpr((uint32_t)tables[q.v_table].n_tiles);
        q.prec_v = (int32_t)sgpr((uint32_t)tables[q.v_table].precision);
        return q;
    };
    // lx0, lro0: this lane's column and row * pitch at its first DMA instruction of a chunk (LDS position 1024 wave + 16 lane)
    auto issue_dma = [&](uint32_t F, const Geo &q, uint32_t lx0, uint32_t lro0, uint32_t c, uint4 *dst) __attribute__((always_inline)) {
        const uint8_t *src = frames + (size_t)q.src * clip_stride + (size_t)(F & 15u) * frame_stride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, frame_bytes, 0x00020000);
        const uint32_t rpc = 16u * q.nb, rows = min(rpc, q.h - c * rpc), bytes = rows * q.wp;
        const uint32_t first = (q.y0 + c * rpc) * pitch + q.x0;  // frame byte of the chunk's first pixel
        // this lane's position in the chunk (column, row * pitch), advanced by adds and one compare per instruction: a multiply
        // high and two multiplies per instruction made the ISSUE of a chunk's DMA cost 0.4 us of a 2.5 us step
        if (q.wp == pitch) {  // LDS pitch == frame pitch (a full-width box, or a narrower one whose padded width happens to be the
            // pitch): LDS position P <- frame byte first + P, a linear copy from the dword at or below the box's first pixel
            const uint32_t first_al = SHIFT ? first & ~3u : first;
            for (uint32_t off = 1024u * wave; off < bytes; off += 4096u) {
                auto *lds = (__attribute__((address_space(3))) void *)&dst[off >> 4];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, (int)(first_al + off + 16u * lane), 0, 0, VDF_STREAM_AUX);
            }
            return;
        }
        uint32_t x = lx0, ro = first + lro0;
        for (uint32_t off = 1024u * wave; off < bytes; off += 4096u) {
            auto *lds = (__attribute__((address_space(3))) void *)&dst[off >> 4];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, (int)((SHIFT ? ro & ~3u : ro) + x), 0, 0, VDF_STREAM_AUX);
            x += q.step_x;
            ro += q.step_rows * pitch;
            if (x >= q.wp) { x -= q.wp; ro += pitch; }
        }
    };
    auto load_av = [&](const Geo &q, uint32_t c, v4i &h, v4i &l) __attribute__((always_inline)) {
        uint32_t rg = (c * q.nb + wave) >> 2;
        rg = rg < (uint32_t)q.n_rg ? rg : (uint32_t)q.n_rg - 1u;
        h = q.av[(rg * 2 + 0) * 64 + lane];
        l = q.av[(rg * 2 + 1) * 64 + lane];
    };
    // biases of the NEXT frame of this workgroup: requested a step ahead (any global load consumed in the step that issues a
    // DMA would have to wait for that DMA - VMEM returns in order - so nothing is consumed in the step that requests it)
    int32_t next_bias_h = 0;
    v4i next_bias_v = {0, 0, 0, 0};
    auto load_biases = [&](const Geo &q) __attribute__((always_inline)) {
        const global_i32 bh_bias = (global_i32)sgpr_ptr(tables[q.h_table].bias), bv_bias = (global_i32)sgpr_ptr(tables[q.v_table].bias);
        next_bias_h = bh_bias[r16];
#pragma unroll
        for (int r = 0; r < 4; r++) next_bias_v[r] = bv_bias[4 * g + r];
    };

    uint32_t cur_x0 = 0, cur_ro0 = 0, nf_x0 = 0, nf_ro0 = 0;
    auto lane_start = [&](const Geo &q, uint32_t &lx0, uint32_t &lro0) __attribute__((always_inline)) {
        const uint32_t P0 = 1024u * wave + 16u * lane, row0 = P0 / q.wp;  // one division per frame, not per chunk
        lx0 = P0 - row0 * q.wp;
        lro0 = row0 * pitch;
    };
    uint32_t F = blockIdx.x, c = 0;
    Geo cur = {}, nf = {};  // the frame whose chunks are being multiplied; this workgroup's next frame
    if (F < n_frames) { cur = geo_of(F); load_biases(cur); lane_start(cur, cur_x0, cur_ro0); }
    uint32_t loaded_table = 0xFFFFFFFFu;
    int32_t bias_h = 0, band_lo = 0, pend_prec_v = 0;
    uint32_t band_nt = 0, band_base = 0, band_zero = 0;
    v4i bias_v = zero4, pend_bias_v = zero4;
    v4i acc_vh = zero4, acc_vl = zero4, pend_vh = zero4, pend_vl = zero4;
    bool out_pending = false;
    uint32_t out_F = 0;
    auto write_pending = [&]() __attribute__((always_inline)) {
        if (out_pending && wave == 0) {
            v4i vh = pend_vh, vl = pend_vl;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                vl[r] += pend_bias_v[r];
#pragma unroll
                for (int w = 0; w < 3; w++) vl[r] += s_part[w][lane][r];
            }
            const uint32_t px = finalize4(vh, vl, pend_prec_v) ^ 0x80808080u;
            uint8_t *dst = small + (size_t)out_F * 256;
#pragma unroll
            for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + r16] = (uint8_t)(px >> (8 * r));
        }
        out_pending = false;
    };
    auto step = [&](const uint4 *cur_px, uint4 *nxt_px, const v4i &avh, const v4i &avl, v4i &avh_n, v4i &avl_n) __attribute__((always_inline)) {
        // Each wave's own DMA instructions of the chunk must have landed BEFORE it arrives at the barrier (the rows of a block come from all
        // four waves).  The fence of __syncthreads() used to bring that vmcnt(0) along, but it is the compiler's to drop: in the ROWCROP
        // instantiation the barrier at the loop header came out without it (seen in the ISA; wrong hashes for a few clips per launch).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk (F, c) has landed and every wave is done with the other buffer (and, at c == 0, with the old table)
        uint32_t Fn = F, cn = c + 1;
        const bool wraps = cn == cur.n_chunks;
        if (wraps) { cn = 0; Fn = F + gridDim.x; }
        // the next chunk's DMA first: with one chunk in flight per workgroup its issue time is on the critical path
        if (Fn < n_frames) {
            if (wraps && c == 0) { nf = geo_of(Fn); lane_start(nf, nf_x0, nf_ro0); }  // one-chunk frames: could not be fetched a step ahead
            if (wraps) issue_dma(Fn, nf, nf_x0, nf_ro0, 0, nxt_px);
            else issue_dma(Fn, cur, cur_x0, cur_ro0, cn, nxt_px);
        }
        write_pending();
        if (c == 0) {  // first chunk of a frame: its biases (requested a frame ago), its table if it differs, the next frame's geometry
            bias_h = next_bias_h;
            bias_v = next_bias_v;
            if (cur.h_table != loaded_table) {  // workgroup-uniform; rare (clips of one source share their box)
                const uint32_t stride = sgpr((uint32_t)tables[cur.h_table].band_stride);
                const global_v4i bh = (global_v4i)sgpr_ptr(tables[cur.h_table].operand);
                const global_i32 meta = (global_i32)sgpr_ptr(tables[cur.h_table].meta);
                for (uint32_t i = tid; i < stride; i += 256u) {
                    const v4i v = bh[i];
                    s_tab[i] = uint4{(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
                }
                if (tid < 8) s_tab[stride + tid] = uint4{0, 0, 0, 0};
                band_lo = meta[r16];
                band_nt = (uint32_t)meta[16 + r16];
                band_base = r16 * stride + 16u * g;
                band_zero = 16u * stride;
                loaded_table = cur.h_table;
                // retire the two loads HERE: a barrier does not wait for loads, and a wait at their first use - inside the K
                // loop, behind this step's DMA - would be a wait for that DMA in every step (seen in the ISA)
                asm volatile("" ::"v"(band_lo), "v"(band_nt));
                __syncthreads();
            }
            if (F + gridDim.x < n_frames) { nf = geo_of(F + gridDim.x); lane_start(nf, nf_x0, nf_ro0); }
        }
        if (Fn < n_frames) {
            if (wraps) load_av(nf, 0, avh_n, avl_n);
            else load_av(cur, cn, avh_n, avl_n);
        }
        const uint32_t rpc = 16u * cur.nb, rows = min(rpc, cur.h - c * rpc);
        if (16u * wave < rows) {
            v4i ah = zero4, al = {bias_h, bias_h, bias_h, bias_h};
            const uint32_t row = 16u * wave + r16;
            const uint8_t *base = reinterpret_cast<const uint8_t *>(cur_px) + row * cur.wp + 16u * g;
            const uint32_t shift = ((cur.y0 + c * rpc + row) * pitch + cur.x0) & 3u;
            auto tile = [&](int kt) __attribute__((always_inline)) {
                const uint4 p = *reinterpret_cast<const uint4 *>(base + 64 * kt);
                v4i a = {(int)p.x, (int)p.y, (int)p.z, (int)p.w};
                if constexpr (SHIFT) {  // some row of some box starts off a dword boundary
                    const uint32_t nx = *reinterpret_cast<const uint32_t *>(base + 64 * kt + 16);
                    a[0] = (int)__builtin_amdgcn_alignbyte(p.y, p.x, shift);
                    a[1] = (int)__builtin_amdgcn_alignbyte(p.z, p.y, shift);
                    a[2] = (int)__builtin_amdgcn_alignbyte(p.w, p.z, shift);
                    a[3] = (int)__builtin_amdgcn_alignbyte(nx, p.w, shift);
                }
                a = a ^ x80;
                const uint32_t j = (uint32_t)(kt - band_lo);
                const uint8_t *q = reinterpret_cast<const uint8_t *>(s_tab) + (j < band_nt ? band_base + j * 128u : band_zero);
                const uint4 th = *reinterpret_cast<const uint4 *>(q), tl = *reinterpret_cast<const uint4 *>(q + 64);
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, (v4i){(int)th.x, (int)th.y, (int)th.z, (int)th.w}, ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, (v4i){(int)tl.x, (int)tl.y, (int)tl.z, (int)tl.w}, al, 0, 0, 0);
            };
            int kt = 0;
            for (; kt + 3 < cur.n_kt; kt += 4) { tile(kt); tile(kt + 1); tile(kt + 2); tile(kt + 3); }
            for (; kt < cur.n_kt; kt++) tile(kt);
            const int val = (int)finalize4(ah, al, cur.prec_h);
            const uint32_t mb = (c * cur.nb + wave) & 3u;
            v4i b;
#pragma unroll
            for (int m = 0; m < 4; m++) b[m] = mb == (uint32_t)m ? val : 0;
            acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(avh, b, acc_vh, 0, 0, 0);
            acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(avl, b, acc_vl, 0, 0, 0);
        }
        if (wraps) {  // frame complete
#ifndef VDF_ABL_NO_ONE_CHUNK_BARRIER
            if (c == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // a box of one chunk: see resize_mfma_frame_stream_kernel
#endif
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 4; r++) s_part[wave - 1][lane][r] = (acc_vh[r] << 8) + acc_vl[r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // see resize_mfma_frame_stream_kernel
            }
            pend_vh = acc_vh; pend_vl = acc_vl; pend_bias_v = bias_v; pend_prec_v = cur.prec_v;
            acc_vh = zero4; acc_vl = zero4;
            out_pending = true;
            out_F = cur.src * 16u + (F & 15u);
            // the next frame's biases, consumed at its first chunk = the next step, behind the barrier (requested here and not in
            // the c == 0 block above, where the same registers are read: the compiler then loaded into temporaries and waited)
            if (Fn < n_frames) load_biases(nf);
            cur = nf; cur_x0 = nf_x0; cur_ro0 = nf_ro0;
        }
        F = sgpr(Fn); c = sgpr(cn);
    };
    v4i av0h = zero4, av0l = zero4, av1h = zero4, av1l = zero4;
    if (F < n_frames) {
        load_av(cur, 0, av0h, av0l);
        issue_dma(F, cur, cur_x0, cur_ro0, 0, s_px0);
    }
    while (F < n_frames) {
        step(s_px0, s_px1, av0h, av0l, av1h, av1l);
        if (!(F < n_frames)) break;
        step(s_px1, s_px0, av1h, av1l, av0h, av0l);
    }
    __syncthreads();
    write_pending();
}


hipError_t launch_resize_mfma_cropped_stream(const uint8_t *frames, size_t n_clips, uint32_t pitch, uint32_t frame_rows,
                                             size_t frame_stride, size_t clip_stride, const CropStreamClip *clips,
                                             const CropStreamTable *tables, int cls, bool shift, uint8_t *small,
                                             hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    if (n_clips * 16 > 0xFFFFFFFFull || (uint64_t)pitch * frame_rows >= (1ull << 31)) return hipErrorInvalidValue;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t n_frames = (uint32_t)(n_clips * 16), frame_bytes = pitch * frame_rows;
    const dim3 grid_s(std::min<uint32_t>(n_frames, (uint32_t)cus * 2u)), grid_m(std::min<uint32_t>(n_frames, (uint32_t)cus));
#define VDF_LAUNCH_CROPPED(BUF, TAB, SH, GRID)                                                                              \
    hipLaunchKernelGGL((resize_mfma_cropped_stream_kernel<BUF, TAB, SH>), GRID, dim3(256), 0, stream, frames, pitch,        \
                       frame_bytes, frame_stride, clip_stride, n_frames, clips, tables, small)
    if (cls == 1 && shift) VDF_LAUNCH_CROPPED(kStreamBufS, kStreamTabS, true, grid_s);
    else if (cls == 1) VDF_LAUNCH_CROPPED(kStreamBufS, kStreamTabS, false, grid_s);
    else if (shift) VDF_LAUNCH_CROPPED(kStreamBufM, kStreamTabM, true, grid_m);
    else VDF_LAUNCH_CROPPED(kStreamBufM, kStreamTabM, false, grid_m);
#undef VDF_LAUNCH_CROPPED
    return hipGetLastError();
}

hipError_t launch_resize_generic(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                 size_t clip_stride, ResizeAxisTable th, ResizeAxisTable tv, int need_h, int need_v,
                                 int32_t y_first, int32_t tmp_rows, uint8_t *small, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    const size_t lds = (size_t)tmp_rows * 16;
    hipLaunchKernelGGL(resize_generic_kernel, dim3((uint32_t)(n_clips * 16)), dim3(256), lds, stream, frames, w, h,
                       frame_stride, clip_stride, th, tv, need_h, need_v, y_first, tmp_rows, small);
    return hipGetLastError();
}

hipError_t launch_dct_hash(const uint8_t *small, size_t small_clip_stride, size_t small_frame_stride, size_t n_clips,
                           const double *cos_table, uint64_t *out_hashes, uint32_t *out_dontcare, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    hipLaunchKernelGGL(dct_hash_kernel, dim3((uint32_t)n_clips), dim3(256), 0, stream, small, small_clip_stride,
                       small_frame_stride, cos_table, out_hashes, out_dontcare);
    return hipGetLastError();
}

// Cropped clips (letterbox crop box read in place: no cropped copy of the frames is ever made).  Every clip may have
// its own box, hence its own coefficient tables: a per-clip descriptor names the box and the two table entries.
template <bool WIDE>
__global__ __launch_bounds__(256) void resize_mfma_cropped_kernel(const uint8_t *__restrict__ frames, uint32_t pitch,
                                                                  size_t frame_stride, size_t clip_stride,
                                                                  const uint8_t *buf_end,
                                                                  const CropClipDesc *__restrict__ desc,
                                                                  const CropTableEntry *__restrict__ tables,
                                                                  uint8_t *__restrict__ small)
{
    __shared__ int32_t s_part[3][2][64][4];
    const size_t clip = blockIdx.x >> 4;
    const uint32_t f = blockIdx.x & 15;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    const CropClipDesc d = desc[clip];
    const CropTableEntry th = tables[d.h_table], tv = tables[d.v_table];
    MfmaResizeTables T;
    T.bh = reinterpret_cast<const v4i *>(th.operand);
    T.av = reinterpret_cast<const v4i *>(tv.operand);
    T.bias_h = th.bias;
    T.bias_v = tv.bias;
    T.prec_h = th.precision;
    T.prec_v = tv.precision;
    T.n_kt = th.n_tiles;
    T.n_rg = tv.n_tiles;
    T.band_meta = nullptr;
    T.band_stride = 0;
    v4i vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
    const uint8_t *src = frames + (size_t)d.src_clip * clip_stride + (size_t)f * frame_stride + (size_t)d.y0 * pitch + d.x0;
    // 16-byte loads may run past the crop box into the rest of the frame (zero coefficients there); only the very
    // end of the buffer needs the careful loader
    if (WIDE) {  // whole-line loads as in resize_mfma_frame_wide_kernel (vertical tables in kMfmaLayoutVerticalWide order)
        const int n_q = (int)((d.h + 31) / 32), q0 = n_q * (int)wave / 4, q1 = n_q * ((int)wave + 1) / 4;
        if (src + (size_t)d.h * pitch + 128 > buf_end) resize_row_quads<true>(src, d.w, d.h, buf_end, T, q0, q1, vh, vl, pitch);
        else resize_row_quads<false>(src, d.w, d.h, buf_end, T, q0, q1, vh, vl, pitch);
    } else {     // narrow frames: a 128-byte window would be half empty
        const int n_blk = (int)((d.h + 15) / 16), b0 = n_blk * (int)wave / 4, b1 = n_blk * ((int)wave + 1) / 4;
        if (src + (size_t)d.h * pitch + 64 > buf_end) resize_row_blocks<true>(src, d.w, d.h, buf_end, T, b0, b1, vh, vl, pitch);
        else resize_row_blocks<false>(src, d.w, d.h, buf_end, T, b0, b1, vh, vl, pitch);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) { s_part[wave - 1][0][lane][r] = vh[r]; s_part[wave - 1][1][lane][r] = vl[r]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            vl[r] += T.bias_v[4 * g + r];
#pragma unroll
            for (int w = 0; w < 3; w++) { vh[r] += s_part[w][0][lane][r]; vl[r] += s_part[w][1][lane][r]; }
        }
        const uint32_t px = finalize4(vh, vl, T.prec_v) ^ 0x80808080u;
        uint8_t *dst = small + ((size_t)d.src_clip * 16 + f) * 256;
#pragma unroll
        for (int r = 0; r < 4; r++) dst[(4 * g + r) * 16 + r16] = (uint8_t)(px >> (8 * r));
    }
}

// Small frames (at most 128 rows, 256 columns; round 5): one workgroup per CLIP - a wave takes four of its frames, a frame's box whole (the
// one-workgroup-per-clip kernel's resize_row_groups at the frame's pitch) - instead of one per frame: a 64 x 64 frame is one tile, and sixteen
// workgroups of 256 threads per clip each did a quarter of one (20 000 letterboxed clips of 64 x 64: crop + hash 1.25 ms against 0.33 ms for
// the same clips without bars).  Vertical tables in the plain layout (kMfmaLayoutVertical).  The DCT runs in the same workgroup (as in
// resize_dct_hash_fused_kernel): no 16 x 16 frames through HBM, no second launch.
// DEVICE_BOX (round 6): no descriptors from the host - the workgroup reads its clip's box {left, right, top, bottom} where the detect kernels
// left it (`boxes`, device memory) and finds the tables of that box size in the per-(W, H) set of ALL box sizes (`tables`: horizontal
// table of box width bw at [bw], vertical of box height bh at [pitch + 1 + bh]; api.cpp: box_table_set), so nothing of the detect's result visits the host
// between the two launches (the reference crops and hashes in one pass per clip: video_hash_builder.rs:188-204).
template <bool DEVICE_BOX>
__global__ __launch_bounds__(256) void resize_dct_hash_cropped_small_kernel(const uint8_t *__restrict__ frames, uint32_t pitch,
                                                                            size_t frame_stride, size_t clip_stride,
                                                                            const uint8_t *buf_end,
                                                                            const CropClipDesc *__restrict__ desc,
                                                                            const CropTableEntry *__restrict__ tables,
                                                                            const double *__restrict__ cos_table,
                                                                            uint64_t *__restrict__ out_hashes,
                                                                            uint32_t *__restrict__ out_dontcare,
                                                                            const uint32_t *__restrict__ boxes, uint32_t frame_rows)
{
    __shared__ DctShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    if (tid < 32) sh.words[tid] = 0u;
    CropClipDesc d;
    if (DEVICE_BOX) {
        const uint4 b = reinterpret_cast<const uint4 *>(boxes)[blockIdx.x];  // left, right, top, bottom (the detect never leaves an empty box)
        d.x0 = b.x; d.y0 = b.z; d.w = pitch - b.x - b.y; d.h = frame_rows - b.z - b.w;
        d.h_table = d.w; d.v_table = pitch + 1 + d.h; d.src_clip = blockIdx.x; d.pad = 0;
    } else {
        d = desc[blockIdx.x];
    }
    const CropTableEntry th = tables[d.h_table], tv = tables[d.v_table];
    MfmaResizeTables T;
    T.bh = reinterpret_cast<const v4i *>(th.operand);
    T.av = reinterpret_cast<const v4i *>(tv.operand);
    T.bias_h = th.bias;
    T.bias_v = tv.bias;
    T.prec_h = th.precision;
    T.prec_v = tv.precision;
    T.n_kt = th.n_tiles;
    T.n_rg = tv.n_tiles;
    T.band_meta = nullptr;
    T.band_stride = 0;
    v4i bias_v;
#pragma unroll
    for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
    const uint8_t *clip0 = frames + (size_t)d.src_clip * clip_stride + (size_t)d.y0 * pitch + d.x0;
    // a box of one tile (at most 64 x 64: every box of the bench's 64 x 64 stacks) well inside the buffer: the wave's sixteen loads - four frames
    // x four row blocks - all go out before the first product, as in resize_dct_hash_fused_kernel<true> (16 KB in flight per wave instead of 4)
    if (T.n_kt == 1 && T.n_rg == 1 && clip0 + 15 * frame_stride + (size_t)d.h * pitch + 64 <= buf_end) {  // workgroup-uniform
        const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
        v4i px[4][4];
        const bool col_ok = 16u * g < d.w;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint8_t *src = clip0 + (size_t)(4 * wave + q) * frame_stride;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t row = 16u * m + r16;
                px[q][m] = (v4i){0, 0, 0, 0};
                if (row < d.h && col_ok) px[q][m] = load_pixels16<false>(src + (size_t)row * pitch + 16u * g, buf_end);
            }
        }
        const v4i bh = T.bh[lane], bl = T.bh[64 + lane], avh = T.av[lane], avl = T.av[64 + lane];
        const int32_t bias_h = T.bias_h[r16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v4i b;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const v4i a = px[q][m] ^ x80;
                v4i ah = {0, 0, 0, 0}, al = {bias_h, bias_h, bias_h, bias_h};
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bh, ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bl, al, 0, 0, 0);
                b[m] = (int)finalize4(ah, al, T.prec_h);
            }
            v4i vh = {0, 0, 0, 0}, vl = bias_v;
            vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(avh, b, vh, 0, 0, 0);
            vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(avl, b, vl, 0, 0, 0);
            sh.cube[(4 * wave + q) * 64 + g * 16 + r16] = finalize4(vh, vl, T.prec_v);
        }
        __syncthreads();
        dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, d.src_clip, out_hashes, out_dontcare);
        return;
    }
#pragma unroll 1
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t f = 4 * wave + q;
        const uint8_t *src = clip0 + (size_t)f * frame_stride;
        v4i vh = {0, 0, 0, 0}, vl = bias_v;
        // 16-byte loads may run past the crop box into the rest of the frame (zero coefficients there); only the very end of the buffer
        // needs the careful loader (wave-uniform test)
        if (src + (size_t)d.h * pitch + 64 > buf_end) resize_row_groups<true>(src, d.w, d.h, buf_end, T, 0, 1, vh, vl, pitch);
        else resize_row_groups<false>(src, d.w, d.h, buf_end, T, 0, 1, vh, vl, pitch);
        sh.cube[f * 64 + g * 16 + r16] = finalize4(vh, vl, T.prec_v);  // centred bytes of out[oy = 4 g + r][x = r16], r = 0..3
    }
    __syncthreads();
    dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, d.src_clip, out_hashes, out_dontcare);
}

hipError_t launch_resize_dct_cropped_small(const uint8_t *frames, size_t n_clips, uint32_t pitch, size_t frame_stride,
                                           size_t clip_stride, const uint8_t *buf_end, const CropClipDesc *desc,
                                           const CropTableEntry *tables, const double *cos_table, uint64_t *out_hashes,
                                           uint32_t *out_dontcare, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    hipLaunchKernelGGL(resize_dct_hash_cropped_small_kernel<false>, dim3((uint32_t)n_clips), dim3(256), 0, stream, frames, pitch, frame_stride,
                       clip_stride, buf_end, desc, tables, cos_table, out_hashes, out_dontcare, (const uint32_t *)nullptr, 0u);
    return hipGetLastError();
}

hipError_t launch_resize_dct_cropped_small_boxes(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                                 size_t clip_stride, const uint8_t *buf_end, const uint32_t *boxes,
                                                 const CropTableEntry *tables, const double *cos_table, uint64_t *out_hashes,
                                                 uint32_t *out_dontcare, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    hipLaunchKernelGGL(resize_dct_hash_cropped_small_kernel<true>, dim3((uint32_t)n_clips), dim3(256), 0, stream, frames, w, frame_stride,
                       clip_stride, buf_end, (const CropClipDesc *)nullptr, tables, cos_table, out_hashes, out_dontcare, boxes, h);
    return hipGetLastError();
}

// ---- small frames (W, H <= 64): letterbox detect + crop + resize + DCT + hash in ONE persistent kernel (round 6) -----------------------
// Replaces, for one-tile frames, the chain  letterbox_kernel -> letterbox_sides_kernel -> copy of the boxes to the host -> host plan ->
// resize_dct_hash_cropped_small_kernel  (the reference detects, crops and hashes clip by clip: video_hash_builder.rs:188-204,
// video_frames_gray.rs:38-128,201-210).  A workgroup owns a clip: the two probed frames (0 and 8) go to LDS, wave e walks in from edge e
// (left / right / top / bottom) of BOTH probes at LDS latency, the box is the per-edge minimum of the two probes' crops (crop.rs:53-68),
// and the sixteen frames' box pixels are then loaded straight into the MFMA operand registers (unaligned 16-byte loads at the box origin;
// the probes come back from L2).  Software pipeline per workgroup: detect of clip i + 1 runs BETWEEN the resize and the DCT of clip i, so
// clip i + 1's pixel loads and clip i + 2's probe loads are in flight under the f64 DCT, as in resize_dct_hash_persistent_kernel.
// Strip test (exact, all integer): four strips per wave instruction stream, one per 16-lane DPP row, two consecutive strips of each
// probe judged speculatively in walking order (the reference's take_while):
//   accept  max - min <= tol      (every pixel within tol of any value of the strip, so of its mode: count = len)
//   reject  no two ADJACENT value bins of width 32 hold more than 9/10 of the strip (the window mode +- tol <= 16 spans at most 33 values,
//           so at most two adjacent bins: its count cannot exceed the best pair)
//   else    the strip's 256-bin histogram in LDS (mode = last maximum; count over [mode - tol, mode + tol]; 10 count > 9 len), one strip per wave.
namespace lbs {
constexpr uint32_t kPitch = 68;                 // LDS row pitch of a probe: 17 dwords (odd), so a column walk touches 16 different banks
constexpr uint32_t kProbeBytes = 64 * kPitch;
constexpr uint32_t kTableStride = 2176;         // bytes per one-tile table of the box-size set: operand hi | lo (2 x 1024), bias[16], precision

struct Shared {
    __attribute__((aligned(16))) uint8_t probe[2 * kProbeBytes];
    uint32_t hist[4][256];  // the exact test's histogram, one per wave
    uint32_t edge[2][4];    // [probe][left, right, top, bottom] strips found
    // the tables of the current and the next clip's box (LDS-DMA right after a clip's detect; in registers across the DCT they were 23 more than
    // three workgroups per CU have): [buffer][0 .. 127] horizontal hi | lo, [128 .. 255] vertical hi | lo; tail: [buffer][h | v] bias[16], precision
    __attribute__((aligned(16))) v4i tab[2][256];
    __attribute__((aligned(16))) v4i tail[2][2][8];
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
template <int ROR> __device__ __forceinline__ uint32_t row_ror(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + ROR, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = max(v, row_ror<8>(v)); v = max(v, row_ror<4>(v)); v = max(v, row_ror<2>(v)); v = max(v, row_ror<1>(v));
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// One strip of len <= 64 pixels at s[i * step], exactly as video_frames_gray.rs:52-99 counts it.  All 64 lanes must call.
__device__ __forceinline__ bool strip_exact(const uint8_t *s, uint32_t step, uint32_t len, uint32_t tol, uint32_t *hist)
{
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; k++) hist[lane + 64 * k] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < len) atomicAdd(&hist[s[lane * step]], 1u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t h[4], key = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        h[k] = hist[lane + 64 * k];
        key = max(key, (h[k] << 8) | (lane + 64 * k));  // max count, ties -> the larger value (Iterator::max_by_key keeps the LAST maximum)
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
    return 10u * count > 9u * len;  // count / len > 0.9 in f64, exactly (cropdetect.hip: more_than_nine_tenths)
}

// Edge e (0 left, 1 right, 2 top, 3 bottom) of both probes: n0 / n1 = how many strips from that edge inwards are letterbox.
// (Eight strips per pass - eight lanes and eight pixels per lane each, a 7-strip bar in two passes instead of four - was built, passed the
// same fuzz, and measured no better: 20 000 clips without bars 0.227 -> 0.237 ms, top / bottom bars 0.254 -> 0.252, side bars 0.268 -> 0.274;
// a pass's longer dependent chain costs what the saved passes gain.  gpurun_out/r6o, LABNOTES R6.1.)
__device__ __forceinline__ void walk_edge(const uint8_t *probe, uint32_t e, uint32_t W, uint32_t H, uint32_t tol, uint32_t *hist,
                                          uint32_t &n0_out, uint32_t &n1_out)
{
    // (everything a lane derives from its number is re-derived per call, behind an asm the compiler cannot see through: hoisted out of the
    // caller's persistent loop these values stayed live across the resize and the DCT - sixteen registers the kernel does not have)
    uint32_t lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const uint32_t rr = lane >> 4, q = lane & 15, p = rr >> 1, k = rr & 1;
    const bool is_row = e >= 2, from_far = (e & 1u) != 0;  // wave-uniform
    const uint32_t limit = is_row ? H : W, len = is_row ? W : H;
    const int32_t left = (int32_t)len - 4 * (int32_t)q;
    const uint32_t nv = (uint32_t)(left < 0 ? 0 : left > 4 ? 4 : left);  // this lane's pixels 4 q .. 4 q + nv - 1 of the strip
    uint32_t n0 = 0, n1 = 0;
    bool live0 = true, live1 = true;
    while (live0 || live1) {
        const uint32_t s = min((p ? n1 : n0) + k, limit - 1);  // (a strip past the end is judged and never looked at)
        const uint32_t idx = from_far ? limit - 1 - s : s;
        const uint8_t *f = probe + p * kProbeBytes;
        uint32_t d;
        if (is_row) {
            d = *reinterpret_cast<const uint32_t *>(f + idx * kPitch + 4u * min(q, (len - 1) >> 2));
            const uint32_t keep = nv >= 4 ? 0xFFFFFFFFu : (1u << (8u * nv)) - 1u;  // bytes past the row's end repeat its first byte of this lane
            d = (d & keep) | (((d & 255u) * 0x01010101u) & ~keep);
        } else {
            const uint8_t *c = f + idx;
            d = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) d |= (uint32_t)c[min(4u * q + j, len - 1) * kPitch] << (8u * j);  // clamped rows repeat the last pixel
        }
        // accept: max - min <= tol
        const u16x2 ev = __builtin_bit_cast(u16x2, d & 0x00FF00FFu), od = __builtin_bit_cast(u16x2, (d >> 8) & 0x00FF00FFu);
        const u16x2 mx2 = __builtin_elementwise_max(ev, od), mn2 = __builtin_elementwise_min(ev, od);
        uint32_t mx = max((uint32_t)mx2.x, (uint32_t)mx2.y), mn = min((uint32_t)mn2.x, (uint32_t)mn2.y);
        if (nv == 0) { mx = 0; mn = 255; }
        u16x2 mm = {(unsigned short)mx, (unsigned short)(255u - mn)};
        mm = __builtin_elementwise_max(mm, __builtin_bit_cast(u16x2, row_ror<8>(__builtin_bit_cast(uint32_t, mm))));
        mm = __builtin_elementwise_max(mm, __builtin_bit_cast(u16x2, row_ror<4>(__builtin_bit_cast(uint32_t, mm))));
        mm = __builtin_elementwise_max(mm, __builtin_bit_cast(u16x2, row_ror<2>(__builtin_bit_cast(uint32_t, mm))));
        mm = __builtin_elementwise_max(mm, __builtin_bit_cast(u16x2, row_ror<1>(__builtin_bit_cast(uint32_t, mm))));
        const bool accept = (uint32_t)mm.x + (uint32_t)mm.y <= 255u + tol;
        // reject: eight bins of 32 values, byte counters (a strip has at most 64 pixels); best adjacent pair
        uint32_t clo = 0, chi = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t v = (d >> (8u * j)) & 255u, b = v >> 5, inc = j < nv ? 1u << (8u * (b & 3u)) : 0u;
            clo += (b & 4u) ? 0u : inc;
            chi += (b & 4u) ? inc : 0u;
        }
        clo += row_ror<8>(clo); chi += row_ror<8>(chi);
        clo += row_ror<4>(clo); chi += row_ror<4>(chi);
        clo += row_ror<2>(clo); chi += row_ror<2>(chi);
        clo += row_ror<1>(clo); chi += row_ror<1>(chi);
        const uint32_t pl = clo + (clo >> 8), ph = chi + (chi >> 8), mid = (clo >> 24) + (chi & 255u);  // sums <= 128: no carries between bytes
        const uint32_t best = max(max(max(pl & 255u, (pl >> 8) & 255u), max((pl >> 16) & 255u, mid)),
                                  max(max(ph & 255u, (ph >> 8) & 255u), (ph >> 16) & 255u));
        const bool reject = !accept && !(10u * best > 9u * len) && tol <= 16u;
        const uint64_t acc_m = __builtin_amdgcn_ballot_w64(accept), rej_m = __builtin_amdgcn_ballot_w64(reject);
        // the four verdicts in walking order
#pragma unroll
        for (uint32_t pp = 0; pp < 2; pp++) {
            uint32_t &n = pp ? n1 : n0;
            bool &live = pp ? live1 : live0;
#pragma unroll
            for (uint32_t kk = 0; kk < 2; kk++) {
                if (live) {
                    if (n >= limit) {
                        live = false;
                    } else {
                        const uint32_t bit = 16u * (2u * pp + kk);
                        bool ok;
                        if ((acc_m >> bit) & 1ull) ok = true;
                        else if ((rej_m >> bit) & 1ull) ok = false;
                        else {
                            const uint32_t at = from_far ? limit - 1 - n : n;
                            ok = strip_exact(probe + pp * kProbeBytes + (is_row ? at * kPitch : at), is_row ? 1u : kPitch, len, tol, hist);
                        }
                        if (ok) n++;
                        else live = false;
                    }
                }
            }
            if (n >= limit) live = false;
        }
    }
    n0_out = n0;
    n1_out = n1;
}
}  // namespace lbs

__global__ __launch_bounds__(256) void letterbox_resize_dct_hash_small_kernel(
    const uint8_t *__restrict__ frames, uint32_t W, uint32_t H, size_t frame_stride, size_t clip_stride,
    const uint8_t *__restrict__ box_tables, const double *__restrict__ cos_table, uint64_t *__restrict__ out_hashes,
    uint32_t *__restrict__ out_dontcare, uint32_t *__restrict__ out_crops, uint32_t n_clips)
{
    [[maybe_unused]] constexpr uint32_t tol = 16;  // LetterboxColour::AnyColour(16): video_frames_gray.rs:205
    __shared__ DctShared sh;
    __shared__ lbs::Shared lb;
    const uint32_t tid = threadIdx.x, lane = tid & 63, g = lane >> 4, r16 = lane & 15;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    if (tid < 32) sh.words[tid] = 0u;
    const v4i x80 = {(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};

    // probes: thread tid moves the 16 bytes at (row tid / 4, column 16 (tid % 4)) of frames 0 and 8
    v4i pr[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    // (addresses as a uniform 64-bit base plus a 32-bit lane offset - the global_load saddr form - and the lane offsets re-derived at
    // every use behind an opaque asm: as loop invariants they were the registers that no longer fitted three workgroups per CU)
    auto opaque_tid = [&]() { uint32_t t = threadIdx.x; asm volatile("" : "+v"(t)); return t; };
    auto issue_probe = [&](uint32_t clip) {
        const uint32_t t = opaque_tid(), prow = t >> 2, pcol = 16u * (t & 3u), p_off = prow * W + pcol;
        const bool p_ok = prow < H && pcol < W;
        const uint8_t *cb = frames + (size_t)clip * clip_stride;
        if (p_ok) {
            pr[0] = load_pixels16<false>(cb + p_off, nullptr);
            pr[1] = load_pixels16<false>(cb + 8 * frame_stride + p_off, nullptr);
        }
    };
    auto store_probe = [&]() {
        const uint32_t t = opaque_tid(), prow = t >> 2, pcol = 16u * (t & 3u);
        const bool p_ok = prow < H && pcol < W;
        if (p_ok) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                uint32_t *d = reinterpret_cast<uint32_t *>(lb.probe + k * lbs::kProbeBytes + prow * lbs::kPitch + pcol);
#pragma unroll
                for (int j = 0; j < 4; j++) d[j] = (uint32_t)pr[k][j];
            }
        }
    };
    // box of the clip whose probes are in LDS (all threads; one barrier inside); writes it to out_crops
    uint32_t bx0 = 0, by0 = 0, bw = W, bh = H;
    auto detect = [&](uint32_t clip) {
        uint32_t n0 = 0, n1 = 0;
#ifndef VDF_ABL_NO_WALK
        lbs::walk_edge(lb.probe, wave, W, H, tol, lb.hist[wave], n0, n1);
#endif
        if (lane == 0) { lb.edge[0][wave] = n0; lb.edge[1][wave] = n1; }
        __syncthreads();
        uint32_t c[4];
        {
            const uint32_t l0 = lb.edge[0][0], r0 = lb.edge[0][1], t0 = lb.edge[0][2], b0 = lb.edge[0][3];
            const uint32_t l1 = lb.edge[1][0], r1 = lb.edge[1][1], t1 = lb.edge[1][2], b1 = lb.edge[1][3];
            // video_frames_gray.rs:119-127: converging edges (a uniform frame) mean "no crop" for that frame; crop.rs:53-68: per-edge minimum
            const bool ok0 = (int32_t)W - (int32_t)l0 - (int32_t)r0 >= 1 && (int32_t)H - (int32_t)t0 - (int32_t)b0 >= 1;
            const bool ok1 = (int32_t)W - (int32_t)l1 - (int32_t)r1 >= 1 && (int32_t)H - (int32_t)t1 - (int32_t)b1 >= 1;
            c[0] = min(ok0 ? l0 : 0u, ok1 ? l1 : 0u);
            c[1] = min(ok0 ? r0 : 0u, ok1 ? r1 : 0u);
            c[2] = min(ok0 ? t0 : 0u, ok1 ? t1 : 0u);
            c[3] = min(ok0 ? b0 : 0u, ok1 ? b1 : 0u);
        }
        if (tid < 4) (out_crops + (size_t)clip * 4)[tid] = tid == 0 ? c[0] : tid == 1 ? c[1] : tid == 2 ? c[2] : c[3];
        bx0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c[0]);
        by0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c[2]);
        bw = W - bx0 - (uint32_t)__builtin_amdgcn_readfirstlane((int)c[1]);
        bh = H - by0 - (uint32_t)__builtin_amdgcn_readfirstlane((int)c[3]);
#ifdef VDF_ABL_IGNORE_BOX  // timing ablation (hashes wrong): the detect runs and reports, the resize takes the whole frame
        bx0 = by0 = 0; bw = W; bh = H;
#endif
    };
    // The tables of the current box (bw, bh) -> lb.tab[buf] / lb.tail[buf] by LDS-DMA: no registers, no wait here.  Wave 0 / 1 bring the
    // horizontal table's hi / lo half, wave 2 / 3 the vertical's; the first eight lanes of waves 0 and 2 the 128-byte tail (bias, precision).
    // The reader (the resize of the clip after the current one) sits behind an `s_waitcnt vmcnt(0)` of every wave and a barrier.
    const __amdgpu_buffer_rsrc_t tab_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(box_tables), 0, (W + H + 2u) * lbs::kTableStride, 0x00020000);
    auto dma_tables = [&](uint32_t buf) {
        const uint32_t at = (wave < 2 ? bw : W + 1u + bh) * lbs::kTableStride;  // wave-uniform
        const uint32_t ln = opaque_tid() & 63u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(tab_rs, (__attribute__((address_space(3))) void *)&lb.tab[buf][64u * wave], 16,
                                                 (int)(at + 1024u * (wave & 1u) + 16u * ln), 0, 0, 0);
        if ((wave & 1u) == 0 && ln < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tab_rs, (__attribute__((address_space(3))) void *)&lb.tail[buf][wave >> 1][0], 16,
                                                     (int)(at + 2048u + 16u * ln), 0, 0, 0);
    };
    // the box's pixels of this wave's four frames (4 wave .. 4 wave + 3): quartet q = frame 4 wave + q, four 16-row blocks
    v4i px[4][4];
    auto issue_pixels_q = [&](uint32_t clip, int q) {
        const uint32_t t = opaque_tid(), ln = t & 63u, l_off = (ln & 15u) * W + (ln & 48u);  // row r16, column 16 g
        const uint8_t *cb = frames + (size_t)clip * clip_stride + (size_t)(4 * wave + q) * frame_stride + (size_t)by0 * W + bx0;
        const bool col_ok = (ln & 48u) < bw;
        // Non-temporal: the pixel stream is read once and must not push the PROBE frames out of L2 before this very loop reads them again as
        // frames 0 and 8 (96 workgroups per XCD x 72 KB per clip is more than its 4 MB).  Measured, 20 000 clips of 64 x 64: no bars
        // 0.252 -> 0.240 ms, side bars 0.286 -> 0.279; the stream alone (no walk, no DCT) 0.233 -> 0.203 (gpurun_out/r6e).  -DVDF_LBS_NT=0: plain loads.
#ifndef VDF_LBS_NT
#define VDF_LBS_NT 1
#endif
#pragma unroll
        for (int m = 0; m < 4; m++) {
            px[q][m] = (v4i){0, 0, 0, 0};
            if (16u * m + (ln & 15u) < bh && col_ok) px[q][m] = load_pixels16<false, VDF_LBS_NT != 0>(cb + (size_t)(16 * m) * W + l_off, nullptr);
        }
    };

    // Per workgroup, clip by clip:   [probes of clip + 1 -> LDS, detect, its tables by DMA]  [resize of clip; every frame quartet's registers
    // are refilled with clip + 1's pixels as soon as they are consumed]  [probe loads of clip + 2]  [DCT of clip]
    // so the pixel stream of clip + 1 is in flight under the rest of the resize and the whole DCT, as in resize_dct_hash_persistent_kernel,
    // and the detect works at LDS latency on probes that arrived a DCT ago.
    uint32_t clip = blockIdx.x, cur = 0;
    if (clip < n_clips) {
        issue_probe(clip);
        store_probe();
        __syncthreads();
        detect(clip);
        dma_tables(0);
#pragma unroll
        for (int q = 0; q < 4; q++) issue_pixels_q(clip, q);
        if (clip + gridDim.x < n_clips) issue_probe(clip + gridDim.x);
    }
    while (clip < n_clips) {
        const uint32_t next = clip + gridDim.x;
        const bool more = next < n_clips;  // workgroup-uniform
        // everything this wave asked for has landed: the probes of `next`, its share of tab[cur], the pixels of `clip`
        // Raised priority from here to the last refill: detect + resize are what stands between this workgroup and its next clip's loads,
        // and they would otherwise queue behind the other two workgroups' DCTs (measured, 20 000 clips: no bars 0.234 -> 0.226 ms,
        // top / bottom bars 0.268 -> 0.255, side bars 0.278 -> 0.268; gpurun_out/r6h)
        __builtin_amdgcn_s_setprio(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (more) store_probe();
        __syncthreads();  // probes and tab[cur] complete (the previous detect's reads of lb.probe and lb.edge are a DCT ago)
        if (more) {
            detect(next);
            dma_tables(cur ^ 1u);  // (last read: the resize of the clip before `clip`)
        }
        const v4i t_bh = lb.tab[cur][lane], t_bl = lb.tab[cur][64 + lane], t_avh = lb.tab[cur][128 + lane], t_avl = lb.tab[cur][192 + lane];
        const int32_t bias_h = reinterpret_cast<const int32_t *>(lb.tail[cur][0])[r16];
        const v4i bias_v = lb.tail[cur][1][g];
        const int32_t prec_h = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int32_t *>(lb.tail[cur][0])[16]);
        const int32_t prec_v = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int32_t *>(lb.tail[cur][1])[16]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v4i b;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const v4i a = px[q][m] ^ x80;
                v4i ah = {0, 0, 0, 0}, al = {bias_h, bias_h, bias_h, bias_h};
                ah = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, t_bh, ah, 0, 0, 0);
                al = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, t_bl, al, 0, 0, 0);
                b[m] = (int)finalize4(ah, al, prec_h);
            }
            v4i vh = {0, 0, 0, 0}, vl = bias_v;
            vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(t_avh, b, vh, 0, 0, 0);
            vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(t_avl, b, vl, 0, 0, 0);
            sh.cube[(4 * wave + q) * 64 + g * 16 + r16] = finalize4(vh, vl, prec_v);
            if (more) issue_pixels_q(next, q);  // in flight during the rest of the resize and the whole DCT below
        }
#ifndef VDF_ABL_NO_PREFETCH
        if (next + gridDim.x < n_clips) issue_probe(next + gridDim.x);
#endif
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();  // the cube is complete
#ifdef VDF_ABL_NO_DCT2
        if (tid < 16) out_hashes[(size_t)clip * 16 + tid] = sh.cube[tid * 64];
        __syncthreads();
#else
        dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, clip, out_hashes, out_dontcare);
#endif
        clip = next;
        cur ^= 1u;
    }
}

hipError_t launch_letterbox_hash_small(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                       const void *box_tables, const double *cos_table, uint64_t *out_hashes, uint32_t *out_dontcare,
                                       uint32_t *out_crops, int wgs_per_cu, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    if (w > 64 || h > 64 || n_clips > 0xFFFFFFFFull) return hipErrorInvalidValue;
    int dev = 0, cus = 256, per_cu = 3;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, letterbox_resize_dct_hash_small_kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 3;
    if (wgs_per_cu > 0) per_cu = std::min(per_cu, wgs_per_cu);
    const uint32_t grid = (uint32_t)std::min<size_t>(n_clips, (size_t)cus * (size_t)per_cu);
    hipLaunchKernelGGL(letterbox_resize_dct_hash_small_kernel, dim3(grid), dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride,
                       reinterpret_cast<const uint8_t *>(box_tables), cos_table, out_hashes, out_dontcare, out_crops, (uint32_t)n_clips);
    return hipGetLastError();
}

hipError_t launch_resize_mfma_cropped(const uint8_t *frames, size_t n_clips, uint32_t pitch, size_t frame_stride,
                                      size_t clip_stride, const uint8_t *buf_end, const CropClipDesc *desc,
                                      const CropTableEntry *tables, uint8_t *small, bool wide, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    if (wide)
        hipLaunchKernelGGL(resize_mfma_cropped_kernel<true>, dim3((uint32_t)(n_clips * 16)), dim3(256), 0, stream, frames,
                           pitch, frame_stride, clip_stride, buf_end, desc, tables, small);
    else
        hipLaunchKernelGGL(resize_mfma_cropped_kernel<false>, dim3((uint32_t)(n_clips * 16)), dim3(256), 0, stream, frames,
                           pitch, frame_stride, clip_stride, buf_end, desc, tables, small);
    return hipGetLastError();
}

static MfmaResizeTables make_tables(const MfmaResizeArgs &a)
{
    MfmaResizeTables T;
    T.bh = reinterpret_cast<const v4i *>(a.bh);
    T.av = reinterpret_cast<const v4i *>(a.av);
    T.bias_h = a.bias_h;
    T.bias_v = a.bias_v;
    T.prec_h = a.prec_h;
    T.prec_v = a.prec_v;
    T.n_kt = a.n_kt;
    T.n_rg = a.n_rg;
    T.band_meta = a.band_meta;
    T.band_stride = a.band_stride;
    return T;
}

hipError_t launch_resize_dct_fused(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                   size_t clip_stride, const uint8_t *buf_end, const MfmaResizeArgs &a,
                                   const double *cos_table, uint64_t *out_hashes, uint32_t *out_dontcare,
                                   hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    // The persistent kernels load 16 bytes at a time without looking at the buffer's end.  With W % 16 == 0 no load crosses a row's end; with any other
    // width a row's last load runs up to 15 bytes into what follows - the next row, frame or clip, all inside the buffer, at zero coefficients -
    // except behind the LAST clip: that one goes to the one-workgroup-per-clip kernel and its careful loader (round 5).
    const bool persistent_ok = n_clips <= 0xFFFFFFFFull && !a.no_persistent && a.n_kt <= 4 && a.n_rg <= 4;
    // (clips that overlap or repeat - clip_stride below 16, e.g. 0: one clip hashed n times - end within 15 bytes of the buffer's end more than
    // once: then no clip may take the unchecked loads, and all of them go to the one-workgroup-per-clip kernel below)
    const bool tail = persistent_ok && w % 16 != 0 && n_clips >= 2 && clip_stride >= 16;  // the last clip apart
    const bool inside = w % 16 == 0 || tail;
    const size_t n_all = n_clips;
    if (tail) n_clips -= 1;
    const auto last_clip = [&]() -> hipError_t {
        if (!tail) return hipGetLastError();
        const uint8_t *f = frames + (n_all - 1) * clip_stride;
        uint64_t *oh = out_hashes + (n_all - 1) * 16;
        uint32_t *od = out_dontcare ? out_dontcare + (n_all - 1) : nullptr;
        if (a.n_kt == 1 && a.n_rg == 1)
            hipLaunchKernelGGL(resize_dct_hash_fused_kernel<true>, dim3(1), dim3(256), 0, stream, f, w, h, frame_stride, clip_stride, buf_end,
                               make_tables(a), cos_table, oh, od);
        else
            hipLaunchKernelGGL(resize_dct_hash_fused_kernel<false>, dim3(1), dim3(256), 0, stream, f, w, h, frame_stride, clip_stride, buf_end,
                               make_tables(a), cos_table, oh, od);
        return hipGetLastError();
    };
    if (a.n_kt == 1 && a.n_rg == 1 && inside && n_clips <= 0xFFFFFFFFull && !a.no_persistent) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint32_t grid = (uint32_t)std::min<size_t>(n_clips, (size_t)cus * a.persistent_wgs_per_cu);
        if (w == 64 && h == 64)
            hipLaunchKernelGGL(resize_dct_hash_persistent_kernel<true>, dim3(grid), dim3(256), 0, stream, frames, w, h,
                               frame_stride, clip_stride, make_tables(a), cos_table, out_hashes, out_dontcare,
                               (uint32_t)n_clips);
        else
            hipLaunchKernelGGL(resize_dct_hash_persistent_kernel<false>, dim3(grid), dim3(256), 0, stream, frames, w, h,
                               frame_stride, clip_stride, make_tables(a), cos_table, out_hashes, out_dontcare,
                               (uint32_t)n_clips);
    } else if (a.n_kt <= 4 && a.n_rg <= 4 && inside && n_clips <= 0xFFFFFFFFull && !a.no_persistent) {
        // up to 128 x 128: units of eight loads per lane in flight, persistent
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
#define VDF_TILED(KERNEL)                                                                                                       \
    do {                                                                                                                         \
        int per_cu = 3;                                                                                                          \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 256, 0) != hipSuccess || per_cu < 1) per_cu = 3;       \
        const uint32_t grid = (uint32_t)std::min<size_t>(n_clips, (size_t)cus * (size_t)per_cu);                                 \
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, stream, frames, w, h, frame_stride, clip_stride, make_tables(a),    \
                           cos_table, out_hashes, out_dontcare, (uint32_t)n_clips);                                              \
    } while (0)
        if (a.n_rg > 2) {  // 129 ... 256 rows: four row groups (the fourth may be empty)
            if (a.n_kt == 4) VDF_TILED((resize_dct_hash_tiled_kernel<4, 4, 2>));
            else if (a.n_kt == 3) VDF_TILED((resize_dct_hash_tiled_kernel<3, 4, 3>));
            else if (a.n_kt == 2) VDF_TILED((resize_dct_hash_tiled_kernel<2, 4, 2>));
            else VDF_TILED((resize_dct_hash_tiled_kernel<1, 4, 3>));
        } else if (a.n_kt == 4 && a.n_rg == 2) VDF_TILED((resize_dct_hash_tiled_kernel<4, 2, 2>));
        else if (a.n_kt == 4) VDF_TILED((resize_dct_hash_tiled_kernel<4, 1, 2>));
        else if (a.n_kt == 3 && a.n_rg == 2) VDF_TILED((resize_dct_hash_tiled_kernel<3, 2, 3>));
        else if (a.n_kt == 3) VDF_TILED((resize_dct_hash_tiled_kernel<3, 1, 3>));
        else if (a.n_kt == 2 && a.n_rg == 2) VDF_TILED((resize_dct_hash_tiled_kernel<2, 2, 3>));
        else if (a.n_kt == 2) VDF_TILED((resize_dct_hash_tiled_kernel<2, 1, 1>));
        else VDF_TILED((resize_dct_hash_tiled_kernel<1, 2, 1>));
#undef VDF_TILED
    } else if (a.n_kt == 1 && a.n_rg == 1)
        hipLaunchKernelGGL(resize_dct_hash_fused_kernel<true>, dim3((uint32_t)n_clips), dim3(256), 0, stream, frames, w,
                           h, frame_stride, clip_stride, buf_end, make_tables(a), cos_table, out_hashes, out_dontcare);
    else
        hipLaunchKernelGGL(resize_dct_hash_fused_kernel<false>, dim3((uint32_t)n_clips), dim3(256), 0, stream, frames,
                           w, h, frame_stride, clip_stride, buf_end, make_tables(a), cos_table, out_hashes,
                           out_dontcare);
    return last_clip();
}

hipError_t launch_resize_mfma_frames(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h,
                                     size_t frame_stride, size_t clip_stride, const uint8_t *buf_end,
                                     const MfmaResizeArgs &a, uint8_t *small, bool wide, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    if (!wide) return hipErrorInvalidValue;  // round 3: the per-frame 16 x 64 B kernel is gone (dominated at every size, profiles/r03_resize_sweep.txt)
    // a.av is in kMfmaLayoutVerticalWide order
    hipLaunchKernelGGL(resize_mfma_frame_wide_kernel, dim3((uint32_t)(n_clips * 16)), dim3(256), 0, stream, frames, w,
                       h, frame_stride, clip_stride, buf_end, make_tables(a), small);
    return hipGetLastError();
}

}  // namespace vdf
