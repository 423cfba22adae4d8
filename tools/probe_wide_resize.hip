// Debug harness: run resize_mfma_frame_kernel and resize_mfma_frame_wide_kernel on the same frames and diff the 16x16 outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I include -I vid_dup_finder_lib_amd/csrc \
//         tools/probe_wide_resize.hip vid_dup_finder_lib_amd/csrc/resize_tables.cpp -o tools/probe_wide_resize
#define VDF_DEBUG_WIDE 1
#include "../vid_dup_finder_lib_amd/csrc/dct_hash.hip"
#include "../vid_dup_finder_lib_amd/csrc/resize_tables.h"
#include <cstdio>
#include <vector>

using namespace vdf;

static void *up(const void *h, size_t n) { void *d; (void)hipMalloc(&d, n); (void)hipMemcpy(d, h, n, hipMemcpyHostToDevice); return d; }

int main(int argc, char **argv)
{
    const uint32_t H = argc > 1 ? atoi(argv[1]) : 32, W = argc > 2 ? atoi(argv[2]) : 128;
    std::vector<uint8_t> fr((size_t)16 * W * H + 256);
    uint64_t x = 88172645463325252ull;
    for (auto &b : fr) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; b = (uint8_t)(x >> 24); }
    uint8_t *d_fr = (uint8_t *)up(fr.data(), fr.size());
    MfmaAxisTable th, tv, tw;
    build_mfma_axis_table(W, kMfmaLayoutHorizontal, th);
    build_mfma_axis_table(H, kMfmaLayoutVertical, tv);
    build_mfma_axis_table(H, kMfmaLayoutVerticalWide, tw);
    printf("H=%u W=%u ok=%d %d %d n_kt=%d n_rg=%d prec %d %d\n", H, W, th.ok, tv.ok, tw.ok, th.n_tiles, tv.n_tiles, th.precision, tv.precision);
    MfmaResizeArgs a{};
    a.bh = up(th.operand.data(), th.operand.size());
    a.bias_h = (const int32_t *)up(th.bias.data(), 64);
    a.bias_v = (const int32_t *)up(tv.bias.data(), 64);
    a.prec_h = th.precision; a.prec_v = tv.precision; a.n_kt = th.n_tiles; a.n_rg = tv.n_tiles;
    const void *av_old = up(tv.operand.data(), tv.operand.size()), *av_wide = up(tw.operand.data(), tw.operand.size());
    uint8_t *s0, *s1; (void)hipMalloc(&s0, 4096); (void)hipMalloc(&s1, 4096);
    const uint8_t *buf_end = d_fr + (size_t)16 * W * H;
    a.av = av_old;
    launch_resize_mfma_frames(d_fr, 1, W, H, (size_t)W * H, (size_t)16 * W * H, buf_end, a, s0, false, 0);
    a.av = av_wide;
    launch_resize_mfma_frames(d_fr, 1, W, H, (size_t)W * H, (size_t)16 * W * H, buf_end, a, s1, true, 0);
    (void)hipDeviceSynchronize();
    std::vector<uint8_t> h0(4096), h1(4096);
    (void)hipMemcpy(h0.data(), s0, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), s1, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4096; i++) bad += h0[i] != h1[i];
    printf("differing bytes: %d of 4096\n", bad);
    for (int f = 0; f < 1; f++) {
        printf("frame %d old:\n", f);
        for (int y = 0; y < 16; y++) { for (int xx = 0; xx < 16; xx++) printf("%4d", h0[f * 256 + y * 16 + xx]); printf("\n"); }
        printf("frame %d wide:\n", f);
        for (int y = 0; y < 16; y++) { for (int xx = 0; xx < 16; xx++) printf("%4d", h1[f * 256 + y * 16 + xx]); printf("\n"); }
    }
    {
        int hb[64 * 4];
        (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(vdf::g_dbg_b), sizeof hb);
        printf("vertical B operand of wave (frame 0), lanes with o = 0: bytes (xor 0x80 removed)\n");
        for (int G = 0; G < 4; G++) {
            const unsigned char *pb = (const unsigned char *)&hb[(16 * G + 0) * 4];
            printf("G=%d:", G);
            for (int j = 0; j < 16; j++) printf(" %3d", pb[j] ^ 0x80);
            printf("\n");
        }
    }
    {
        int ha[4 * 64 * 4];
        (void)hipMemcpyFromSymbol(ha, HIP_SYMBOL(vdf::g_dbg_acc), sizeof ha);
        const char *nm[4] = {"eh", "el", "oh", "ol"};
        for (int G = 0; G < 2; G++)
            for (int k = 0; k < 4; k++) {
                const int *pa = &ha[(k * 64 + 16 * G + 0) * 4];
                printf("octet 1 lane(o=0,G=%d) %s: %d %d %d %d\n", G, nm[k], pa[0], pa[1], pa[2], pa[3]);
            }
        printf("prec_h %d bias_h[0] %d\n", th.precision, th.bias[0]);
    }
    return 0;
}
