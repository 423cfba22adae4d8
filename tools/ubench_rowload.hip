// Load-only probe for the large-frame resize: does the MFMA A-operand access shape (a wave instruction = 16 rows x 64 B
// at the frame's row pitch) cap HBM bandwidth, compared with the same bytes read as 8 rows x 128 B or fully linear?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rowload.hip -o tools/ubench_rowload
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

// one workgroup per frame (W x H u8), 4 waves take 64-row groups round-robin, like resize_mfma_frame_kernel
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void probe(const uint8_t *__restrict__ frames, uint32_t W, uint32_t H, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *src = frames + (size_t)blockIdx.x * W * H;
    uint4 acc = {0, 0, 0, 0};
    if (MODE == 2) {  // linear: the workgroup streams the frame 4 KB at a time
        const size_t n16 = (size_t)W * H / 16;
        for (size_t i = threadIdx.x; i < n16; i += 256 * DEPTH) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) v[d] = i + 256 * d < n16 ? reinterpret_cast<const uint4 *>(src)[i + 256 * d] : uint4{0, 0, 0, 0};
#pragma unroll
            for (int d = 0; d < DEPTH; d++) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
        }
    } else {
        const int n_rg = (H + 63) / 64, n_kt = (W + 63) / 64;
        for (int rg = wave; rg < n_rg; rg += 4) {
            for (int kt = 0; kt < n_kt; kt += DEPTH) {
                uint4 v[DEPTH][4];
#pragma unroll
                for (int d = 0; d < DEPTH; d++)
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        uint32_t row, x;
                        if (MODE == 0) { row = 64u * rg + 16u * m + (lane & 15); x = 64u * (kt + d) + 16u * (lane >> 4); }   // 16 rows x 64 B
                        else { row = 64u * rg + 16u * m + 8u * (d & 1) + (lane >> 3); x = 128u * ((kt + d) >> 1) + 16u * (lane & 7); }  // 8 rows x 128 B
                        v[d][m] = (row < H && x < W && kt + d < n_kt) ? *reinterpret_cast<const uint4 *>(src + (size_t)row * W + x) : uint4{0, 0, 0, 0};
                    }
#pragma unroll
                for (int d = 0; d < DEPTH; d++)
#pragma unroll
                    for (int m = 0; m < 4; m++) { acc.x ^= v[d][m].x; acc.y ^= v[d][m].y; acc.z ^= v[d][m].z; acc.w ^= v[d][m].w; }
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[threadIdx.x] = acc.x;
}

template <int MODE, int DEPTH>
static void run(const uint8_t *d, uint32_t W, uint32_t H, uint32_t n_frames, uint32_t *o, const char *name)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(n_frames), dim3(256), 0, 0, d, W, H, o);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(n_frames), dim3(256), 0, 0, d, W, H, o);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%ux%u %-28s depth %d: %.3f ms  %.0f GB/s\n", W, H, name, DEPTH, ms, (double)n_frames * W * H / ms / 1e6);
}

int main()
{
    const uint32_t sizes[][3] = {{1920, 1080, 8000}, {480, 270, 80000}, {128, 128, 320000}};
    for (auto &sz : sizes) {
        const uint32_t W = sz[0], H = sz[1], n = sz[2];
        uint8_t *d; uint32_t *o;
        (void)hipMalloc(&d, (size_t)n * W * H + 256); (void)hipMalloc(&o, 4096);
        (void)hipMemset(d, 1, (size_t)n * W * H + 256);
        run<0, 1>(d, W, H, n, o, "16 rows x 64 B (MFMA shape)");
        run<0, 2>(d, W, H, n, o, "16 rows x 64 B (MFMA shape)");
        run<0, 4>(d, W, H, n, o, "16 rows x 64 B (MFMA shape)");
        run<1, 2>(d, W, H, n, o, "8 rows x 128 B");
        run<1, 4>(d, W, H, n, o, "8 rows x 128 B");
        run<2, 4>(d, W, H, n, o, "linear");
        run<2, 8>(d, W, H, n, o, "linear");
        (void)hipFree(d); (void)hipFree(o);
    }
    return 0;
}
