// The MFMA search kernel runs at the chip's POWER cap, not at its issue rate (profiles/r02_pmc_kernel_compare.txt: a
// stream with 87 % matrix-pipe utilisation and one with 81 % take the same wall time, the chip just clocks lower), so what
// decides the pairs/s is the ENERGY per pair.  This microbenchmark compares, on the kernel's per-wave work unit (64 target
// rows x 32 candidates, targets in registers, changing B fragments, 2 waves per SIMD, >= 150 ms per mode so the clock
// settles), variants that do the same exact integer job with different energy:
//   0  v_mfma_scale_f32_32x32x64_f8f6f4, scales 127/127 (2^0), operands +-1 e2m1, 13 k-steps (832 bits)   <- the kernel today
//   1  the same through the UNSCALED opcode (scale operands 0: LLVM selects v_mfma_f32_32x32x64_f8f6f4), values checked == mode 0
//   2  mode 0 with operands {0, +1} instead of {-1, +1} (dot = popcount(a & b); distance = pa + pb - 2 dot)
//   3  v_mfma_scale_f32_16x16x128_f8f6f4, +-1, 7 k-steps of 128 whose last step carries zeros in its upper 64 positions
//      (832 real bits in 7 x 16-cycle instructions per 16 x 16 tile: more cycles than mode 0, but zeros are cheap)
//   4  mode 3 with all 8 x 128 bits real (1024 bits: what no early exit costs on that shape), for reference
//   5  mode 0 with 16 k-steps (1024 bits), for reference
//   6  mode 0 with A (targets) +-1 and B (candidates) {0, 1};  7  A {0, 1} and B +-1;  8  mode 2 through the unscaled opcode
//   9  mode 6 through the unscaled opcode
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_mfma_energy.hip -o tools/ubench_mfma_energy
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void mfma_loop(const uint4 *__restrict__ data, uint32_t iters, float *out, float *check)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i a[32], b[8];
    const uint4 *p = data + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 40;
#pragma unroll
    for (int i = 0; i < 32; i++) { const uint4 v = p[i]; a[i] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = p[32 + i]; b[i] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }
    constexpr bool kA01 = MODE == 2 || MODE == 7 || MODE == 8, kB01 = MODE == 2 || MODE == 6 || MODE == 8 || MODE == 9;
    if (kA01) {
#pragma unroll
        for (int i = 0; i < 32; i++) { a[i].x = (a[i].x >> 2) & 0x22222222; a[i].y = (a[i].y >> 2) & 0x22222222; a[i].z = (a[i].z >> 2) & 0x22222222; a[i].w = (a[i].w >> 2) & 0x22222222; }
    }
    if (kB01) {
#pragma unroll
        for (int i = 0; i < 8; i++) { b[i].x = (b[i].x >> 2) & 0x22222222; b[i].y = (b[i].y >> 2) & 0x22222222; b[i].z = (b[i].z >> 2) & 0x22222222; b[i].w = (b[i].w >> 2) & 0x22222222; }
    }
    if (false) {  // {0, +1}: clear the sign bits (0xA -> 0x2) and then drop half of the ones: 0x2 -> 0x0 where the sign was set
#pragma unroll
        for (int i = 0; i < 32; i++) { a[i].x = (a[i].x >> 2) & 0x22222222; a[i].y = (a[i].y >> 2) & 0x22222222; a[i].z = (a[i].z >> 2) & 0x22222222; a[i].w = (a[i].w >> 2) & 0x22222222; }
#pragma unroll
        for (int i = 0; i < 8; i++) { b[i].x = (b[i].x >> 2) & 0x22222222; b[i].y = (b[i].y >> 2) & 0x22222222; b[i].z = (b[i].z >> 2) & 0x22222222; b[i].w = (b[i].w >> 2) & 0x22222222; }
    }
    float m = -1e30f, sum = 0.f;
    for (uint32_t it = 0; it < iters; it++) {
        if constexpr (MODE == 0 || MODE == 1 || MODE == 2 || MODE == 5 || MODE >= 6) {
            constexpr int KS = MODE == 5 ? 16 : 13;
            v16f acc0 = {}, acc1 = {};
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const v8i bb = {b[s & 7].x, b[s & 7].y, b[s & 7].z, b[s & 7].w, 0, 0, 0, 0};
                const v8i a0 = {a[s].x, a[s].y, a[s].z, a[s].w, 0, 0, 0, 0};
                const v8i a1 = {a[16 + s].x, a[16 + s].y, a[16 + s].z, a[16 + s].w, 0, 0, 0, 0};
                if constexpr (MODE == 1 || MODE == 8 || MODE == 9) {
                    acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, bb, acc0, 4, 4, 0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, bb, acc1, 4, 4, 0, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, bb, acc0, 4, 4, 0, 127, 0, 127);
                    acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, bb, acc1, 4, 4, 0, 127, 0, 127);
                }
            }
            m = fmaxf(m, fmaxf(acc0[0], acc1[5]));
            if (it == 0) { sum = 0.f; for (int r = 0; r < 16; r++) sum += acc0[r] * (float)(r + 1) + acc1[r] * (float)(r + 17); }
        } else {
            constexpr int KS = MODE == 3 ? 7 : 8;
            v4f acc[4][2] = {};
#pragma unroll
            for (int s = 0; s < KS; s++) {
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
                    v4i bv = b[(2 * s + ct) & 7];
                    if (MODE == 3 && s == KS - 1 && lane >= 32) bv = (v4i){0, 0, 0, 0};  // upper 64 of the 128 positions: zeros
                    const v8i bb = {bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0};
#pragma unroll
                    for (int rt = 0; rt < 4; rt++) {
                        v4i av = a[rt * 8 + s];
                        if (MODE == 3 && s == KS - 1 && lane >= 32) av = (v4i){0, 0, 0, 0};
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
    if (check && blockIdx.x == 0) check[threadIdx.x] = sum;  // first iteration's accumulators, folded: modes 0 and 1 must agree
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
    uint4 *d; float *o, *chk;
    hipMalloc(&d, n_vec * 16); hipMalloc(&o, 4096); hipMalloc(&chk, 2 * 256 * 4);
    hipMemcpy(d, h.data(), n_vec * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[10] = {"32x32x64 scaled +-1, 832 bits      ", "32x32x64 UNSCALED +-1, 832 bits    ", "32x32x64 scaled {0,1}, 832 bits    ",
                            "16x16x128 +-1, 832 bits + 64 zeros ", "16x16x128 +-1, 1024 bits           ", "32x32x64 scaled +-1, 1024 bits     ",
                            "32x32x64 A +-1, B {0,1}, 832 bits  ", "32x32x64 A {0,1}, B +-1, 832 bits  ", "32x32x64 UNSCALED {0,1}, 832 bits  ",
                            "32x32x64 UNSCALED A+-1 B{0,1}, 832 "};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 10; mode++) {
            auto launch = [&](uint32_t n, float *c) {
                switch (mode) {
                case 0: hipLaunchKernelGGL(mfma_loop<0>, dim3(n_wg), dim3(256), 0, 0, d, n, o, c); break;
                case 1: hipLaunchKernelGGL(mfma_loop<1>, dim3(n_wg), dim3(256), 0, 0, d, n, o, c ? c + 256 : c); break;
                case 2: hipLaunchKernelGGL(mfma_loop<2>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                case 3: hipLaunchKernelGGL(mfma_loop<3>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                case 4: hipLaunchKernelGGL(mfma_loop<4>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                case 5: hipLaunchKernelGGL(mfma_loop<5>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                case 6: hipLaunchKernelGGL(mfma_loop<6>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                case 7: hipLaunchKernelGGL(mfma_loop<7>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                case 8: hipLaunchKernelGGL(mfma_loop<8>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                default: hipLaunchKernelGGL(mfma_loop<9>, dim3(n_wg), dim3(256), 0, 0, d, n, o, nullptr); break;
                }
            };
            launch(iters / 10, chk);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch(iters, nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double units_per_simd = 2.0 * iters;  // 2 waves per SIMD, one 64 x 32 unit per iteration
            printf("rep %d mode %d %s: %.1f ms  %.1f ns per 64x32 unit -> %.3e pairs/s chip-wide at this bit count\n", rep, mode, names[mode], ms,
                   ms * 1e6 / units_per_simd, (double)n_cu * 4 * units_per_simd * 64.0 * 32.0 / (ms * 1e-3));
        }
    std::vector<float> c(512);
    hipMemcpy(c.data(), chk, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    double s0 = 0;
    for (int i = 0; i < 256; i++) { bad += c[i] != c[256 + i]; s0 += c[i]; }
    printf("unscaled opcode vs scales 127/127: %d of 256 lanes differ (checksum %.1f)\n", bad, s0);
    return 0;
}
