// Sustained issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4) with the search kernel's register shape and
// REALISTIC operands: 2 row tiles x 16 k-steps of +-1 e2m1 nibbles in VGPRs, a different B fragment per step, runs of
// >= 100 ms so that the clock settles (the chip lowers its clock under load, and toggling operands draw more power
// than the near-zero ones tools/ubench_mfma.hip uses).  This is the ceiling the MFMA search kernel is compared with.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_mfma_sustained.hip -o tools/ubench_mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void mfma_loop(const uint4 *__restrict__ data, uint32_t iters, float *out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i a[2][16], b[8];
    const uint4 *p = data + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 40;
#pragma unroll
    for (int i = 0; i < 32; i++) { const uint4 v = p[i]; a[i >> 4][i & 15] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = p[32 + i]; b[i] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
    float m = -1e30f;
    for (uint32_t it = 0; it < iters; it++) {
        v16f acc0 = {}, acc1 = {};
#pragma unroll
        for (int s = 0; s < 16; s++) {
            const v8i bb = {b[s & 7].x, b[s & 7].y, b[s & 7].z, b[s & 7].w, 0, 0, 0, 0};
            const v8i a0 = {a[0][s].x, a[0][s].y, a[0][s].z, a[0][s].w, 0, 0, 0, 0};
            const v8i a1 = {a[1][s].x, a[1][s].y, a[1][s].z, a[1][s].w, 0, 0, 0, 0};
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, bb, acc0, 4, 4, 0, 127, 0, 127);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, bb, acc1, 4, 4, 0, 127, 0, 127);
        }
        // keep the B fragments changing (as fresh LDS reads would) at negligible VALU cost: rotate one register
#pragma unroll
        for (int i = 0; i < 8; i++) b[i].x = __builtin_amdgcn_alignbit(b[i].x, b[i].x, 4);
        m = fmaxf(m, fmaxf(acc0[0], acc1[5]));
    }
    if (m == 12345.0f) out[threadIdx.x] = m;
}

int main(int argc, char **argv)
{
    const int wgs_per_cu = 2, n_cu = 256;
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000;
    const int n_wg = n_cu * wgs_per_cu;
    const size_t n_vec = (size_t)n_wg * 4 * 64 * 40;
    for (int mode = 0; mode < 3; mode++) {  // 0: zeros, 1: small ints (old ubench), 2: random +-1 nibbles (the real data)
        std::vector<uint32_t> h(n_vec * 4);
        uint64_t x = 88172645463325252ull;
        for (auto &w : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            if (mode == 0) w = 0;
            else if (mode == 1) w = (uint32_t)(x & 15);
            else { uint32_t bits = (uint32_t)x & 0x88888888u; w = bits | 0x22222222u; }
        }
        uint4 *d; float *o;
        hipMalloc(&d, n_vec * 16); hipMalloc(&o, 4096);
        hipMemcpy(d, h.data(), n_vec * 16, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(mfma_loop, dim3(n_wg), dim3(256), 0, 0, d, iters / 10, o);  // warm-up
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(n_wg), dim3(256), 0, 0, d, iters, o);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: 2 waves x iters x 32 MFMAs
        const double mfma_per_simd = 2.0 * iters * 32.0;
        const double ns = ms * 1e6 / mfma_per_simd;
        const double pairs = (double)n_cu * 4 * mfma_per_simd * 1024.0 / 16.0 / (ms * 1e-3);  // 32x32 pairs per 16 MFMAs
        printf("mode=%d (%s)  %.1f ms  %.2f ns/MFMA/SIMD (%.1f cyc @2.4GHz)  -> %.3e Hamming pairs/s chip-wide\n", mode,
               mode == 0 ? "zeros" : mode == 1 ? "small ints" : "random +-1 fp4", ms, ns, ns * 2.4, pairs);
        hipFree(d); hipFree(o);
    }
    return 0;
}
