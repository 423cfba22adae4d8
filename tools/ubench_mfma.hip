// MFMA issue-rate microbenchmark (gfx950): cycles per instruction per SIMD for the integer / fp4 forms that an
// exact +-1-encoded Hamming Gram matrix could use.  One wave per SIMD and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define ITER 2048

__global__ __launch_bounds__(256) void k_i8_16(v4i *out, int seed)
{
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed ^ 5, seed, 7, 9};
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < ITER; i++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}
__global__ __launch_bounds__(256) void k_i8_32(v16i *out, int seed)
{
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed ^ 5, seed, 7, 9};
    v16i c0 = {}, c1 = {};
    for (int i = 0; i < ITER; i++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1;
}
// f8f6f4 with both operands fp4 (cbsz = blgp = 4): 16x16x128 and 32x32x64, scale exponents 127 (= 1.0)
__global__ __launch_bounds__(256) void k_fp4_16(v4f *out, int seed)
{
    v8i a = {seed, seed + 1, seed + 2, seed + 3, 0, 0, 0, 0}, b = {seed ^ 5, seed, 7, 9, 0, 0, 0, 0};
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < ITER; i++) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 4, 4, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 4, 4, 0, 127, 0, 127);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3;
}
__global__ __launch_bounds__(256) void k_fp4_32(v16f *out, int seed)
{
    v8i a = {seed, seed + 1, seed + 2, seed + 3, 0, 0, 0, 0}, b = {seed ^ 5, seed, 7, 9, 0, 0, 0, 0};
    v16f c0 = {}, c1 = {};
    for (int i = 0; i < ITER; i++) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1;
}

template <class T, class K> void run(const char *name, K kern, int waves_per_simd, double macs_per_instr)
{
    const int blocks = 256 * waves_per_simd;
    T *out; hipMalloc(&out, (size_t)blocks * 256 * sizeof(T));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)ITER * 4 * waves_per_simd;
    const double ns = ms * 1e6 / instr_per_simd;
    // Hamming pairs: one instruction covers (M*N) pairs x K bit positions; a pair needs 1024 bit positions
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f ns/instr/SIMD (%.1f cyc @2.4GHz)  -> %.3g Hamming pairs/s chip-wide\n", name,
           waves_per_simd, ms, ns, ns * 2.4, macs_per_instr / 1024.0 / (ns * 1e-9) * 1024.0);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<v4i>("i8 16x16x64", k_i8_16, w, 16.0 * 16 * 64);
        run<v16i>("i8 32x32x32", k_i8_32, w, 32.0 * 32 * 32);
        run<v4f>("fp4 16x16x128", k_fp4_16, w, 16.0 * 16 * 128);
        run<v16f>("fp4 32x32x64", k_fp4_32, w, 32.0 * 32 * 64);
    }
    return 0;
}
