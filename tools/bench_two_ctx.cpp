// Two (or one) contexts on one GPU, each driven by its own host thread calling the batch hash entry point in a loop on its own
// page-locked buffer of n clips: what the batching queue's slots do, without the queue.  Prints per-call time and the aggregate rate.
// Usage: tools/bench_two_ctx <w> <h> <n_clips> <contexts> <letterbox> [seconds] [collect: 1 = n threads copy one clip each from pageable memory into the buffer before every call]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../include/vdf.h"

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const uint32_t w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    const size_t n = std::atoi(argv[3]);
    const int C = std::atoi(argv[4]), letterbox = std::atoi(argv[5]);
    const double seconds = argc > 6 ? std::atof(argv[6]) : 3.0;
    const int collect = argc > 7 ? std::atoi(argv[7]) : 0;
    const size_t clip = (size_t)w * h * 16;
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> calls{0};
    std::vector<double> sum_ms((size_t)C, 0.0), sum_collect((size_t)C, 0.0);
    auto worker = [&](int c) {
        vdf_ctx *ctx = nullptr;
        if (vdf_ctx_create(0, &ctx) != VDF_OK) return;
        uint8_t *buf = nullptr;
        if (hipHostMalloc((void **)&buf, n * clip, hipHostMallocDefault) != hipSuccess) return;
        std::mt19937_64 rng(5 + c);
        for (size_t i = 0; i < n * clip / 8; i++) reinterpret_cast<uint64_t *>(buf)[i] = rng();
        std::vector<uint64_t> out(n * VDF_HASH_WORDS);
        std::vector<uint32_t> crops(n * 4);
        std::vector<std::vector<uint8_t>> src;
        if (collect) {
            src.assign(n, std::vector<uint8_t>(clip));
            for (size_t i = 0; i < n; i++) std::memcpy(src[i].data(), buf + i * clip, clip);
        }
        uint64_t k = 0;
        while (!stop.load()) {
            const auto tc = std::chrono::steady_clock::now();
            if (collect) {
                std::vector<std::thread> cp;
                for (size_t i = 0; i < n; i++) cp.emplace_back([&, i] { std::memcpy(buf + i * clip, src[i].data(), clip); });
                for (auto &t : cp) t.join();
            }
            const auto t0 = std::chrono::steady_clock::now();
            if (k >= 3) sum_collect[c] += std::chrono::duration<double, std::milli>(t0 - tc).count();
            const int rc = letterbox ? vdf_hash_frames_u8_letterbox(ctx, buf, n, 16, w, h, (size_t)w * h, clip, out.data(), crops.data(), nullptr)
                                     : vdf_hash_frames_u8(ctx, buf, n, 16, w, h, (size_t)w * h, clip, out.data(), nullptr);
            if (rc != VDF_OK) break;
            if (k++ >= 3) { sum_ms[c] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); calls++; }
        }
        hipHostFree(buf);
        vdf_ctx_destroy(ctx);
    };
    std::vector<std::thread> th;
    for (int c = 0; c < C; c++) th.emplace_back(worker, c);
    std::this_thread::sleep_for(std::chrono::milliseconds(1500));
    const uint64_t c0 = calls.load();
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const uint64_t c1 = calls.load();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true;
    for (auto &x : th) x.join();
    double ms = 0, cms = 0;
    for (double v : sum_ms) ms += v;
    for (double v : sum_collect) cms += v;
    std::printf("collect %.2f ms per call | ", cms / std::max<uint64_t>(calls.load(), 1));
    std::printf("%ux%u n=%zu contexts=%d letterbox=%d: %.1f calls/s, %.2f ms per call on average, aggregate %.1f GB/s\n", w, h, n, C, letterbox,
                (c1 - c0) / dt, ms / std::max<uint64_t>(calls.load(), 1), (c1 - c0) / dt * n * clip / 1e9);
    return 0;
}
