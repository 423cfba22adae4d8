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

__global__ __launch_bounds__(256) void dct_hash_kernel(const uint8_t *__restrict__ small, size_t clip_stride,
                                                       size_t frame_stride, const double *__restrict__ cos_table,
                                                       uint64_t *__restrict__ out_hashes,
                                                       uint32_t *__restrict__ out_dontcare)
{
    __shared__ double s_b[16 * 10 * kPadY];  // pass-x output  [t][kx][y]
    __shared__ double s_c[16 * 100];         // pass-y output  [t][kx][ky]
    __shared__ double s_cos[10 * 16];        // cos[kt][t] for the per-lane kt of the last pass
    __shared__ uint32_t s_dc[4];
    const size_t clip = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const_f64_ptr cosv = (const_f64_ptr)(uintptr_t)cos_table;

    if (tid < 160) s_cos[tid] = cos_table[tid];

    // pass x: thread (t, y) owns one 16-pixel row (16 contiguous bytes)
    {
        const uint32_t t = tid >> 4, y = tid & 15;
        const uint4 px = *reinterpret_cast<const uint4 *>(small + clip * clip_stride + (size_t)t * frame_stride + y * 16);
        const uint32_t wsrc[4] = {px.x, px.y, px.z, px.w};
        double v[16], o[10];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = (double)(int32_t)((wsrc[i >> 2] >> ((i & 3) * 8)) & 255u) - 128.0;
        dct16_pruned(v, o, cosv);
#pragma unroll
        for (int kx = 0; kx < 10; kx++) s_b[(t * 10 + kx) * kPadY + y] = o[kx];
    }
    __syncthreads();
    // pass y: thread (t, kx), 160 lines
    if (tid < 160) {
        double v[16], o[10];
#pragma unroll
        for (int y = 0; y < 16; y++) v[y] = s_b[tid * kPadY + y];
        dct16_pruned(v, o, cosv);
#pragma unroll
        for (int ky = 0; ky < 10; ky++) s_c[tid * 10 + ky] = o[ky];
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
            for (int t = 0; t < 16; t++) coef = fma(s_c[t * 100 + rem], s_cos[kt * 16 + t], coef);
        }
        const unsigned long long bits = __builtin_amdgcn_ballot_w64(coef > 0.0);  // 0.0 and NaN -> 0
        const unsigned long long tiny = __builtin_amdgcn_ballot_w64(i < 1000 && fabs(coef) < 1e-6);
        dc += (uint32_t)__builtin_popcountll(tiny);
        if (lane == 0) out_hashes[clip * 16 + word] = bits;
    }
    if (out_dontcare) {
        if (lane == 0) s_dc[wave] = dc;
        __syncthreads();
        if (tid == 0) out_dontcare[clip] = s_dc[0] + s_dc[1] + s_dc[2] + s_dc[3];
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

}  // namespace vdf
