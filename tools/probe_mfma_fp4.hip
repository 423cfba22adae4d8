// Confirms v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands on gfx950: +-1 encoding (0x2 / 0xA),
// unit E8M0 scales (127), lane maps (A row / B col = lane & 31, k-group = lane >> 5, C row = (reg&3)+8(reg>>2)+4(lane>>5)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const v8i *a, const v8i *b, v16f *d)
{
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 4, 4, 0, 127, 0, 127);
    d[threadIdx.x] = acc;
}

int main()
{
    int A[32][64], B[64][32];  // bits
    srand(7);
    for (int i = 0; i < 32; i++) for (int kk = 0; kk < 64; kk++) A[i][kk] = rand() & 1;
    for (int kk = 0; kk < 64; kk++) for (int j = 0; j < 32; j++) B[kk][j] = rand() & 1;
    int ref[32][32];
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int kk = 0; kk < 64; kk++) s += (A[i][kk] == B[kk][j]) ? 1 : -1; ref[i][j] = s; }
    uint8_t pa[64][32], pb[64][32];
    memset(pa, 0, sizeof pa); memset(pb, 0, sizeof pb);
    for (int l = 0; l < 64; l++) {
        int g = l >> 5, r = l & 31;
        for (int q = 0; q < 32; q++) {  // nibble q of the lane's 16 bytes <-> k = 32 g + q
            int kk = 32 * g + q;
            uint8_t na = A[r][kk] ? 0xA : 0x2, nb = B[kk][r] ? 0xA : 0x2;
            pa[l][q >> 1] |= na << (4 * (q & 1));
            pb[l][q >> 1] |= nb << (4 * (q & 1));
        }
    }
    v8i *da, *db; v16f *dd; float hd[64][16];
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 4096);
    hipMemcpy(da, pa, 2048, hipMemcpyHostToDevice); hipMemcpy(db, pb, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        if (hd[l][r] != (float)ref[row][col]) { if (bad < 5) printf("lane %d reg %d got %g want %d\n", l, r, hd[l][r], ref[row][col]); bad++; }
    }
    printf("fp4 32x32x64 +-1 dot products: %d mismatches of 1024\n", bad);
    return 0;
}
