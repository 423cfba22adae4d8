// Sustained rate of the two fp4 MFMA shapes on the search kernel's per-wave work unit (64 target rows x 32 candidates
// x 1024 bits, targets in 128 VGPRs, a changing B fragment per step) with random +-1 e2m1 operands, runs of >= 100 ms:
//   shape 0: v_mfma_scale_f32_32x32x64_f8f6f4   2 row tiles x 16 k-steps           = 32 MFMAs of 32 cycles
//   shape 1: v_mfma_scale_f32_16x16x128_f8f6f4  4 row tiles x 2 col tiles x 8 steps = 64 MFMAs of 16 cycles
// Both are 1024 matrix-pipe cycles per unit; MI355X_MICROARCH.md (DVFS give-back, item 7) reports that the chip holds
// a higher clock on the 16x16 bf16 shape under load.  This measures whether that carries over to fp4.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_mfma_shapes.hip -o tools/ubench_mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop(const uint4 *__restrict__ data, uint32_t iters, float *out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i a[32], b[8];
    const uint4 *p = data + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 40;
#pragma unroll
    for (int i = 0; i < 32; i++) { const uint4 v = p[i]; a[i] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = p[32 + i]; b[i] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
    float m = -1e30f;
    for (uint32_t it = 0; it < iters; it++) {
        if constexpr (SHAPE == 0) {
            v16f acc0 = {}, acc1 = {};
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const v8i bb = {b[s & 7].x, b[s & 7].y, b[s & 7].z, b[s & 7].w, 0, 0, 0, 0};
                const v8i a0 = {a[s].x, a[s].y, a[s].z, a[s].w, 0, 0, 0, 0};
                const v8i a1 = {a[16 + s].x, a[16 + s].y, a[16 + s].z, a[16 + s].w, 0, 0, 0, 0};
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, bb, acc0, 4, 4, 0, 127, 0, 127);
                acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, bb, acc1, 4, 4, 0, 127, 0, 127);
            }
            m = fmaxf(m, fmaxf(acc0[0], acc1[5]));
        } else if constexpr (SHAPE == 2) {  // the same 32 MFMAs as two SEQUENTIAL dependent chains (one accumulator at a time)
            v16f acc0 = {}, acc1 = {};
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const v8i bb = {b[s & 7].x, b[s & 7].y, b[s & 7].z, b[s & 7].w, 0, 0, 0, 0};
                const v8i a0 = {a[s].x, a[s].y, a[s].z, a[s].w, 0, 0, 0, 0};
                acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, bb, acc0, 4, 4, 0, 127, 0, 127);
            }
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const v8i bb = {b[s & 7].x, b[s & 7].y, b[s & 7].z, b[s & 7].w, 0, 0, 0, 0};
                const v8i a1 = {a[16 + s].x, a[16 + s].y, a[16 + s].z, a[16 + s].w, 0, 0, 0, 0};
                acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, bb, acc1, 4, 4, 0, 127, 0, 127);
            }
            m = fmaxf(m, fmaxf(acc0[0], acc1[5]));
        } else {
            v4f acc[4][2] = {};
#pragma unroll
            for (int s = 0; s < 8; s++) {
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
                    const v4i bv = b[(2 * s + ct) & 7];
                    const v8i bb = {bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0};
#pragma unroll
                    for (int rt = 0; rt < 4; rt++) {
                        const v4i av = a[rt * 8 + s];
                        const v8i aa = {av.x, av.y, av.z, av.w, 0, 0, 0, 0};
                        acc[rt][ct] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(aa, bb, acc[rt][ct], 4, 4, 0, 127, 0, 127);
                    }
                }
            }
            float t = 0.f;
#pragma unroll
            for (int rt = 0; rt < 4; rt++) t += acc[rt][0][rt] + acc[rt][1][3 - rt];
            m = fmaxf(m, t);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) b[i].x = __builtin_amdgcn_alignbit(b[i].x, b[i].x, 4);
    }
    if (m == 12345.0f) out[threadIdx.x] = m;
}

int main(int argc, char **argv)
{
    const int wgs_per_cu = 2, n_cu = 256;
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000;
    const int n_wg = n_cu * wgs_per_cu;
    const size_t n_vec = (size_t)n_wg * 4 * 64 * 40;
    std::vector<uint32_t> h(n_vec * 4);
    uint64_t x = 88172645463325252ull;
    for (auto &w : h) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        w = ((uint32_t)x & 0x88888888u) | 0x22222222u;
    }
    uint4 *d; float *o;
    hipMalloc(&d, n_vec * 16); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), n_vec * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++)
        for (int shape = 0; shape < 3; shape++) {
            auto launch = [&](uint32_t n) {
                if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(n_wg), dim3(256), 0, 0, d, n, o);
                else if (shape == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(n_wg), dim3(256), 0, 0, d, n, o);
                else hipLaunchKernelGGL(mfma_loop<1>, dim3(n_wg), dim3(256), 0, 0, d, n, o);
            };
            launch(iters / 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch(iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double units_per_simd = 2.0 * iters;  // 2 waves per SIMD, one 64 x 32 x 1024 unit per iteration
            const double pairs = (double)n_cu * 4 * units_per_simd * 64.0 * 32.0 / (ms * 1e-3);
            printf("rep %d shape %s: %.1f ms  %.1f ns per 1024-cycle unit (%.3f GHz-equivalent)  -> %.3e Hamming pairs/s chip-wide\n",
                   rep, shape == 0 ? "32x32x64 interleaved" : shape == 2 ? "32x32x64 sequential " : "16x16x128           ", ms, ms * 1e6 / units_per_simd, 1024.0 / (ms * 1e6 / units_per_simd), pairs);
        }
    return 0;
}
