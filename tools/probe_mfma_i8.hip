// Confirms the A/B/C lane maps of v_mfma_i32_16x16x64_i8 on gfx950 with exact integer data
// (cdna_hip_programming.md: "check the map with exact integer data before relying on it").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k(const v4i *a, const v4i *b, v4i *d)
{
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    d[threadIdx.x] = acc;
}

int main()
{
    int8_t A[16][64], B[64][16];
    srand(1);
    for (int i = 0; i < 16; i++) for (int kk = 0; kk < 64; kk++) A[i][kk] = (int8_t)(rand() % 255 - 127);
    for (int kk = 0; kk < 64; kk++) for (int j = 0; j < 16; j++) B[kk][j] = (int8_t)(rand() % 255 - 127);
    int ref[16][16];
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int kk = 0; kk < 64; kk++) s += A[i][kk] * B[kk][j]; ref[i][j] = s; }
    for (int hyp = 0; hyp < 2; hyp++) {
        int8_t pa[64][16], pb[64][16];
        for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
            int g = l >> 4, r = l & 15;
            int kk = hyp == 0 ? 16 * g + j : (j < 8 ? 8 * g + j : 32 + 8 * g + (j - 8));
            pa[l][j] = A[r][kk];
            pb[l][j] = B[kk][r];
        }
        v4i *da, *db, *dd; int hd[64][4];
        hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
        hipMemcpy(da, pa, 1024, hipMemcpyHostToDevice); hipMemcpy(db, pb, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) if (hd[l][r] != ref[(l >> 4) * 4 + r][l & 15]) bad++;
        printf("hypothesis %d (%s): %d mismatches of 256 (C map: col=lane&15,row=4*(lane>>4)+reg)\n", hyp,
               hyp == 0 ? "k = 16*(lane>>4)+j" : "k = 8*(lane>>4)+j | 32+8*(lane>>4)+(j-8)", bad);
    }
    return 0;
}
