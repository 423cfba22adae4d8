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

// 16-point unnormalised DCT-II pruned to outputs 0..9:  X[k] = sum_n v[n] cos(pi k (n + 1/2) / 16).
// One even/odd split: cos(pi k (15-n+1/2)/16) = (-1)^k cos(pi k (n+1/2)/16).
__device__ __forceinline__ void dct16_pruned(const double (&v)[16], double (&out)[10], const_f64_ptr cosv)
{
    double u[8], d[8];
#pragma unroll
    for (int n = 0; n < 8; n++) { u[n] = v[n] + v[15 - n]; d[n] = v[n] - v[15 - n]; }
#pragma unroll
    for (int k = 0; k < 10; k++) {
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < 8; n++) acc = fma((k & 1) ? d[n] : u[n], cosv[k * 16 + n], acc);
        out[k] = acc;
    }
}

constexpr int kPadY = 17;  // [t][kx][y] rows padded to 17 doubles

struct DctShared {
    double b[16 * 10 * kPadY];  // pass-x output  [t][kx][y]
    double c[16 * 100];         // pass-y output  [t][kx][ky]
    double cosv[10 * 16];       // cos[kt][t] for the per-lane kt of the last pass
    uint32_t dc[4];
    __attribute__((aligned(16))) uint8_t cube[16 * 256];  // resized clip [t][y][x] u8
};

// 256 threads: 3-D DCT-II of the u8 cube in sh.cube (pix - 128, f64), sign test, ballot pack.
// Caller has filled sh.cube and sh.cosv and synchronised.
__device__ __forceinline__ void dct_hash_block(DctShared &sh, const_f64_ptr cosv, size_t clip,
                                               uint64_t *__restrict__ out_hashes, uint32_t *__restrict__ out_dontcare)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pass x: thread (t, y) owns one 16-pixel row (16 contiguous bytes)
    {
        const uint32_t t = tid >> 4, y = tid & 15;
        const uint4 px = *reinterpret_cast<const uint4 *>(sh.cube + tid * 16);
        const uint32_t wsrc[4] = {px.x, px.y, px.z, px.w};
        double v[16], o[10];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = (double)(int32_t)((wsrc[i >> 2] >> ((i & 3) * 8)) & 255u) - 128.0;
        dct16_pruned(v, o, cosv);
#pragma unroll
        for (int kx = 0; kx < 10; kx++) sh.b[(t * 10 + kx) * kPadY + y] = o[kx];
    }
    __syncthreads();
    // pass y: thread (t, kx), 160 lines
    if (tid < 160) {
        double v[16], o[10];
#pragma unroll
        for (int y = 0; y < 16; y++) v[y] = sh.b[tid * kPadY + y];
        dct16_pruned(v, o, cosv);
#pragma unroll
        for (int ky = 0; ky < 10; ky++) sh.c[tid * 10 + ky] = o[ky];
    }
    __syncthreads();
    // pass t + sign + pack: lane l of wave-word w computes bit i = 64 w + l = 100 kt + 10 kx + ky
    uint32_t dc = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t word = wave + 4 * q;
        const uint32_t i = word * 64 + lane;
        double coef = 0.0;
        if (i < 1000) {
            const uint32_t kt = i / 100, rem = i - kt * 100;  // rem = 10 kx + ky
#pragma unroll
            for (int t = 0; t < 16; t++) coef = fma(sh.c[t * 100 + rem], sh.cosv[kt * 16 + t], coef);
        }
        const unsigned long long bits = __builtin_amdgcn_ballot_w64(coef > 0.0);  // 0.0 and NaN -> 0
        const unsigned long long tiny = __builtin_amdgcn_ballot_w64(i < 1000 && fabs(coef) < 1e-6);
        dc += (uint32_t)__builtin_popcountll(tiny);
        if (lane == 0) out_hashes[clip * 16 + word] = bits;
    }
    if (out_dontcare) {
        if (lane == 0) sh.dc[wave] = dc;
        __syncthreads();
        if (tid == 0) out_dontcare[clip] = sh.dc[0] + sh.dc[1] + sh.dc[2] + sh.dc[3];
    }
}

__global__ __launch_bounds__(256) void dct_hash_kernel(const uint8_t *__restrict__ small, size_t clip_stride,
                                                       size_t frame_stride, const double *__restrict__ cos_table,
                                                       uint64_t *__restrict__ out_hashes,
                                                       uint32_t *__restrict__ out_dontcare)
{
    __shared__ DctShared sh;
    const size_t clip = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    if (tid < 160) sh.cosv[tid] = cos_table[tid];
    {
        const uint32_t t = tid >> 4, y = tid & 15;
        *reinterpret_cast<uint4 *>(sh.cube + tid * 16) =
            *reinterpret_cast<const uint4 *>(small + clip * clip_stride + (size_t)t * frame_stride + y * 16);
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
typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));

struct MfmaResizeTables {
    const v4i *bh;          // [kt][2][64]   horizontal B operand (hi, lo)
    const v4i *av;          // [rg][2][64]   vertical A operand (hi, lo)
    const int32_t *bias_h;  // [16]
    const int32_t *bias_v;  // [16]
    int32_t prec_h, prec_v, n_kt, n_rg;
};

__device__ __forceinline__ v4i load_pixels16(const uint8_t *p, const uint8_t *buf_end)
{
    if (p + 16 <= buf_end) {
        const u32x4_unaligned v = *reinterpret_cast<const u32x4_unaligned *>(p);
        v4i r = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        return r;
    }
    uint32_t w[4] = {0, 0, 0, 0};  // last bytes of the buffer: never read past the end
    for (int i = 0; i < 16; i++)
        if (p + i < buf_end) w[i >> 2] |= (uint32_t)p[i] << ((i & 3) * 8);
    v4i r = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    return r;
}

__device__ __forceinline__ uint32_t finalize4(v4i hi, v4i lo, int prec)
{  // four (256 hi + lo) >> prec, clamped to u8, packed little-endian, re-centred for the next i8 product
    uint32_t packed = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int32_t s = ((hi[r] << 8) + lo[r]) >> prec;
        s = min(max(s, 0), 255);
        packed |= (uint32_t)s << (8 * r);
    }
    return packed ^ 0x80808080u;
}

// Row groups rg_begin, rg_begin + rg_step, ... of one frame; accumulates the vertical partial sums.
__device__ __forceinline__ void resize_row_groups(const uint8_t *__restrict__ src, uint32_t W, uint32_t H,
                                                  const uint8_t *buf_end, const MfmaResizeTables &T, int rg_begin,
                                                  int rg_step, v4i &acc_vh, v4i &acc_vl)
{
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
    const int32_t bias_h = T.bias_h[r16];
    for (int rg = rg_begin; rg < T.n_rg; rg += rg_step) {
        v4i ah[4], al[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { ah[m] = (v4i){0, 0, 0, 0}; al[m] = (v4i){bias_h, bias_h, bias_h, bias_h}; }
        for (int kt = 0; kt < T.n_kt; kt++) {
            const v4i bh = T.bh[(kt * 2 + 0) * 64 + lane], bl = T.bh[(kt * 2 + 1) * 64 + lane];
            const uint32_t x = 64u * kt + 16u * g;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const uint32_t row = 64u * rg + 16u * m + r16;
                v4i a = {0, 0, 0, 0};
                if (row < H && x < W) a = load_pixels16(src + (size_t)row * W + x, buf_end);
                a ^= (v4i){(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};
                ah[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bh, ah[m], 0, 0, 0);
                al[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bl, al[m], 0, 0, 0);
            }
        }
        v4i b;
#pragma unroll
        for (int m = 0; m < 4; m++) b[m] = (int)finalize4(ah[m], al[m], T.prec_h);
        acc_vh = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 0) * 64 + lane], b, acc_vh, 0, 0, 0);
        acc_vl = __builtin_amdgcn_mfma_i32_16x16x64_i8(T.av[(rg * 2 + 1) * 64 + lane], b, acc_vl, 0, 0, 0);
    }
}

// Small frames: one workgroup per clip, one wave per frame (4 frames each), resized frames go to LDS and the
// DCT runs in the same kernel: HBM traffic = the frames once + 128 B of hash.
__global__ __launch_bounds__(256) void resize_dct_hash_fused_kernel(
    const uint8_t *__restrict__ frames, uint32_t W, uint32_t H, size_t frame_stride, size_t clip_stride,
    const uint8_t *buf_end, MfmaResizeTables T, const double *__restrict__ cos_table,
    uint64_t *__restrict__ out_hashes, uint32_t *__restrict__ out_dontcare)
{
    __shared__ DctShared sh;
    const size_t clip = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    if (tid < 160) sh.cosv[tid] = cos_table[tid];
    v4i bias_v;
#pragma unroll
    for (int r = 0; r < 4; r++) bias_v[r] = T.bias_v[4 * g + r];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t f = wave + 4 * q;
        v4i vh = {0, 0, 0, 0}, vl = bias_v;
        resize_row_groups(frames + clip * clip_stride + (size_t)f * frame_stride, W, H, buf_end, T, 0, 1, vh, vl);
        const uint32_t px = finalize4(vh, vl, T.prec_v) ^ 0x80808080u;  // back to plain u8: out[oy = 4 g + r][x = r16]
#pragma unroll
        for (int r = 0; r < 4; r++) sh.cube[f * 256 + (4 * g + r) * 16 + r16] = (uint8_t)(px >> (8 * r));
    }
    __syncthreads();
    dct_hash_block(sh, (const_f64_ptr)(uintptr_t)cos_table, clip, out_hashes, out_dontcare);
}

// Large frames: one workgroup per frame, the four waves take alternate 64-row groups and their vertical
// partial sums (exact i32) are added through LDS.  Writes the 16 x 16 u8 frame to `small`.
__global__ __launch_bounds__(256) void resize_mfma_frame_kernel(const uint8_t *__restrict__ frames, uint32_t W,
                                                                uint32_t H, size_t frame_stride, size_t clip_stride,
                                                                const uint8_t *buf_end, MfmaResizeTables T,
                                                                uint8_t *__restrict__ small)
{
    __shared__ int32_t s_part[3][2][64][4];
    const size_t clip = blockIdx.x >> 4;
    const uint32_t f = blockIdx.x & 15;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r16 = lane & 15;
    v4i vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
    resize_row_groups(frames + clip * clip_stride + (size_t)f * frame_stride, W, H, buf_end, T, (int)wave, 4, vh, vl);
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
    return T;
}

hipError_t launch_resize_dct_fused(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                                   size_t clip_stride, const uint8_t *buf_end, const MfmaResizeArgs &a,
                                   const double *cos_table, uint64_t *out_hashes, uint32_t *out_dontcare,
                                   hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    hipLaunchKernelGGL(resize_dct_hash_fused_kernel, dim3((uint32_t)n_clips), dim3(256), 0, stream, frames, w, h,
                       frame_stride, clip_stride, buf_end, make_tables(a), cos_table, out_hashes, out_dontcare);
    return hipGetLastError();
}

hipError_t launch_resize_mfma_frames(const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h,
                                     size_t frame_stride, size_t clip_stride, const uint8_t *buf_end,
                                     const MfmaResizeArgs &a, uint8_t *small, hipStream_t stream)
{
    if (n_clips == 0) return hipSuccess;
    hipLaunchKernelGGL(resize_mfma_frame_kernel, dim3((uint32_t)(n_clips * 16)), dim3(256), 0, stream, frames, w, h,
                       frame_stride, clip_stride, buf_end, make_tables(a), small);
    return hipGetLastError();
}

}  // namespace vdf
