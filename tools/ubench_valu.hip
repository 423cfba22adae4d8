// Instruction-rate microbenchmark for gfx950: cycles per wave64 VALU instruction per SIMD at 8 waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu ; run on the GPU box.
// Each kernel issues ITER x 32 independent instructions per wave from inline asm; rate is derived from
// HIP-event time and s_memtime (shader clock) deltas.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 4096

#define BODY32(INS)                                                                                          \
    INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13)    \
    INS(14) INS(15) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11)    \
    INS(12) INS(13) INS(14) INS(15)

#define DEF_KERNEL(NAME, ASM_STR, CONSTRAINT_T, INIT)                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, unsigned long long *cyc, uint32_t seed)          \
    {                                                                                                          \
        CONSTRAINT_T a[16];                                                                                    \
        for (int i = 0; i < 16; i++) a[i] = INIT;                                                              \
        uint32_t s = seed | 1u;                                                                                \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int it = 0; it < ITER; it++) {                                                                    \
            _Pragma("unroll") for (int r = 0; r < 2; r++) {                                                    \
                _Pragma("unroll") for (int i = 0; i < 16; i++) { asm volatile(ASM_STR : "+v"(a[i]) : "s"(s)); } \
            }                                                                                                  \
        }                                                                                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        uint32_t acc = 0;                                                                                      \
        for (int i = 0; i < 16; i++) acc ^= (uint32_t)a[i];                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                             \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                       \
    }

typedef uint32_t u32;
typedef float f32;
typedef double f64;
typedef unsigned long long u64;

DEF_KERNEL(k_xor, "v_xor_b32 %0, %1, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_add, "v_add_u32 %0, %1, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_and, "v_and_b32 %0, %1, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_min3, "v_min3_u32 %0, %1, %0, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_mad24, "v_mad_u32_u24 %0, %1, %0, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_mullo, "v_mul_lo_u32 %0, %1, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_dot4, "v_dot4_i32_i8 %0, %1, %0, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_dot2, "v_dot2_i32_i16 %0, %1, %0, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %1, %0, %0 bitop3:0x96", u32, threadIdx.x + i)
DEF_KERNEL(k_sad, "v_sad_u8 %0, %1, %0, %0", u32, threadIdx.x + i)
DEF_KERNEL(k_fma32, "v_fma_f32 %0, %1, %0, %0", f32, 1.0f + i)
DEF_KERNEL(k_cvtub, "v_cvt_f32_ubyte1 %0, %0", f32, 1.0f + i)
DEF_KERNEL(k_pkfma, "v_pk_fma_f32 %0, %0, %0, %0", u64, (u64)(threadIdx.x + i))
DEF_KERNEL(k_fma64, "v_fma_f64 %0, %0, %0, %0", f64, 1.0 + i)
DEF_KERNEL(k_add64, "v_add_f64 %0, %0, %0", f64, 1.0 + i)
DEF_KERNEL(k_mul64, "v_mul_f64 %0, %0, %0", f64, 1.0 + i)

template <class K> void run(const char *name, K kern, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;  // 4 waves per block -> waves_per_simd waves on each SIMD
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    const double insts_per_wave = (double)ITER * 32;
    // per SIMD: waves_per_simd waves, each insts_per_wave instructions, in `avg` shader cycles
    printf("%-8s waves/SIMD=%d  time %.3f ms  cycles/wave-instr/SIMD (s_memtime) = %.2f   (event-time @2.4GHz: %.2f)\n",
           name, waves_per_simd, ms, avg / (insts_per_wave * waves_per_simd),
           ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 8}) {
        run("xor", k_xor, w); run("bcnt", k_bcnt, w); run("add_u32", k_add, w); run("and", k_and, w);
        run("min3", k_min3, w); run("mad24", k_mad24, w); run("mul_lo", k_mullo, w); run("dot4_i8", k_dot4, w);
        run("dot2_i16", k_dot2, w); run("bitop3", k_bitop3, w); run("sad_u8", k_sad, w); run("fma_f32", k_fma32, w);
        run("cvt_ub", k_cvtub, w); run("pk_fma", k_pkfma, w); run("fma_f64", k_fma64, w); run("add_f64", k_add64, w);
        run("mul_f64", k_mul64, w);
        printf("\n");
    }
    return 0;
}
