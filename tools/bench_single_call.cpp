#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../include/vdf.h"
int main(int argc, char **argv)
{
    const uint32_t w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    const int lb = std::atoi(argv[3]);
    const size_t clip = (size_t)w * h * 16;
    vdf_ctx *ctx = nullptr;
    if (vdf_ctx_create(0, &ctx)) return 1;
    std::vector<uint8_t> buf(clip);
    std::mt19937_64 rng(1);
    for (size_t i = 0; i < clip / 8; i++) reinterpret_cast<uint64_t *>(buf.data())[i] = rng();
    uint64_t out[16]; uint32_t crop[4];
    for (int k = 0; k < 5; k++) lb ? vdf_hash_frames_u8_letterbox(ctx, buf.data(), 1, 16, w, h, (size_t)w * h, clip, out, crop, nullptr) : vdf_hash_frames_u8(ctx, buf.data(), 1, 16, w, h, (size_t)w * h, clip, out, nullptr);
    const int reps = 200;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) lb ? vdf_hash_frames_u8_letterbox(ctx, buf.data(), 1, 16, w, h, (size_t)w * h, clip, out, crop, nullptr) : vdf_hash_frames_u8(ctx, buf.data(), 1, 16, w, h, (size_t)w * h, clip, out, nullptr);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("%ux%u letterbox=%d: one clip per call from pageable memory: %.3f ms per call = %.1f GB/s\n", w, h, lb, ms, clip / ms / 1e6);
    vdf_ctx_destroy(ctx);
    return 0;
}
