// What the batching queue (csrc/hash_queue.cpp) delivers to T caller threads that each hold one decoded clip in ordinary
// (pageable) memory and call vdf_hash_queue_submit in a loop - the app's rayon workers (video_hash_filesystem_cache.rs:237-257) without
// their decoders.  Prints clips/s and the PCIe rate next to the batch call vdf_hash_frames_u8 on the same bytes.
// Build: g++ -O2 -std=c++17 -pthread -o tools/bench_hash_queue tools/bench_hash_queue.cpp -Lvid_dup_finder_lib_amd -lvdf_hip -Wl,-rpath,$PWD/vid_dup_finder_lib_amd
// Usage: tools/bench_hash_queue <w> <h> <threads> <max_batch> <max_wait_us> <letterbox 0|1> [seconds]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../include/vdf.h"

int main(int argc, char **argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: %s w h threads max_batch max_wait_us letterbox [seconds]\n", argv[0]); return 2; }
    const uint32_t w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    const int T = std::atoi(argv[3]);
    const uint32_t max_batch = std::atoi(argv[4]), wait_us = std::atoi(argv[5]);
    const int letterbox = std::atoi(argv[6]);
    const double seconds = argc > 7 ? std::atof(argv[7]) : 4.0;
    const size_t clip = (size_t)w * h * 16;
    vdf_ctx *ctx = nullptr;
    if (vdf_ctx_create(0, &ctx) != VDF_OK) { std::fprintf(stderr, "ctx: %s\n", vdf_last_error(nullptr)); return 1; }
    vdf_hash_queue *q = nullptr;
    if (vdf_hash_queue_create(ctx, w, h, max_batch, wait_us, letterbox, &q) != VDF_OK) { std::fprintf(stderr, "queue: %s\n", vdf_last_error(ctx)); return 1; }
    // one clip per thread, all different; the first thread's clip also goes through the batch call for the reference hash
    std::vector<std::vector<uint8_t>> clips((size_t)T, std::vector<uint8_t>(clip));
    for (int t = 0; t < T; t++) {
        std::mt19937_64 rng(1000 + t);
        uint64_t *p = reinterpret_cast<uint64_t *>(clips[t].data());
        for (size_t i = 0; i < clip / 8; i++) p[i] = rng();
        if (letterbox && t % 2 == 1 && h >= 16)  // every other caller's clip has top / bottom bars: the queue's batches mix box shapes
            for (int f = 0; f < 16; f++) {
                std::memset(clips[t].data() + (size_t)f * w * h, 16, (size_t)w * (h / 8));
                std::memset(clips[t].data() + (size_t)f * w * h + (size_t)w * (h - h / 9), 17, (size_t)w * (h / 9));
            }
    }
    std::vector<uint64_t> want((size_t)T * VDF_HASH_WORDS);
    std::vector<uint32_t> want_crop((size_t)T * 4, 0u);
    for (int t = 0; t < T; t++) {
        const int rc = letterbox ? vdf_hash_frames_u8_letterbox(ctx, clips[t].data(), 1, 16, w, h, (size_t)w * h, clip, &want[(size_t)t * VDF_HASH_WORDS],
                                                                &want_crop[(size_t)t * 4], nullptr)
                                 : vdf_hash_frames_u8(ctx, clips[t].data(), 1, 16, w, h, (size_t)w * h, clip, &want[(size_t)t * VDF_HASH_WORDS], nullptr);
        if (rc != VDF_OK) return 1;
    }
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> done{0}, wrong{0};
    auto worker = [&](int t) {
        uint64_t out[VDF_HASH_WORDS];
        uint32_t crop[4];
        while (!stop.load(std::memory_order_relaxed)) {
            if (vdf_hash_queue_submit(q, clips[t].data(), out, crop) != VDF_OK) { wrong++; break; }
            if (std::memcmp(out, &want[(size_t)t * VDF_HASH_WORDS], sizeof out) != 0 || std::memcmp(crop, &want_crop[(size_t)t * 4], sizeof crop) != 0) wrong++;
            done++;
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    std::this_thread::sleep_for(std::chrono::milliseconds(500));  // warm-up: staging pinned, tables built
    const uint64_t d0 = done.load();
    const auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const uint64_t d1 = done.load();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop = true;
    for (auto &x : th) x.join();
    uint64_t nb = 0, nc = 0;
    vdf_hash_queue_stats(q, &nb, &nc);
    uint32_t infl = 0;
    vdf_hash_queue_in_flight_max(q, &infl);
    // the batch call on the same number of bytes per call as a full queue batch, from pageable memory
    const size_t nbatch = std::max<size_t>(max_batch, 32);
    std::vector<uint8_t> big(nbatch * clip);
    for (size_t c = 0; c < nbatch; c++) std::memcpy(big.data() + c * clip, clips[c % (size_t)T].data(), clip);
    std::vector<uint64_t> hb(nbatch * VDF_HASH_WORDS);
    double best = 1e30;
    for (int rep = 0; rep < 4; rep++) {
        const auto b0 = std::chrono::steady_clock::now();
        const int rc = letterbox ? vdf_hash_frames_u8_letterbox(ctx, big.data(), nbatch, 16, w, h, (size_t)w * h, clip, hb.data(), nullptr, nullptr)
                                 : vdf_hash_frames_u8(ctx, big.data(), nbatch, 16, w, h, (size_t)w * h, clip, hb.data(), nullptr);
        if (rc != VDF_OK) return 1;
        best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - b0).count());
    }
    const double rate = (double)(d1 - d0) / dt;
    std::printf("%ux%u letterbox=%d threads=%d max_batch=%u wait=%uus: queue %.0f clips/s = %.1f GB/s (mean batch %.1f clips, %u batches in flight at most, %llu wrong) | "
                "batch call of %zu clips: %.0f clips/s = %.1f GB/s\n",
                w, h, letterbox, T, max_batch, wait_us, rate, rate * clip / 1e9, nb ? (double)nc / nb : 0.0, infl, (unsigned long long)wrong.load(), nbatch,
                nbatch / best, nbatch * clip / best / 1e9);
    vdf_hash_queue_destroy(q);
    vdf_ctx_destroy(ctx);
    return wrong.load() ? 1 : 0;
}
