// The batching queue's host logic (csrc/hash_queue.cpp: one mutex, a condition variable per kind of wait) under ThreadSanitizer, with the
// GPU behind it replaced by stand-ins: vdf_ctx_create hands out an empty context, vdf_hash_frames_u8[_letterbox] "hashes" a clip to a
// checksum of its bytes after a short sleep.  A lost wake-up shows as a hang (the test's timeout), a wrong hand-over as a wrong checksum,
// an unlocked access as a TSan report.  Built by tests/test_host_sanitizers.py from hash_queue.cpp itself - no GPU, no libvdf_hip.so.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../vid_dup_finder_lib_amd/csrc/vdf_ctx.h"

vdf_ctx::~vdf_ctx() {}  // (api.cpp's releases device objects; the stand-in contexts own none)

static std::atomic<int> g_calls{0}, g_concurrent{0}, g_concurrent_max{0};

static uint64_t checksum(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

extern "C" {
int vdf_ctx_create(int device_id, vdf_ctx **out) { *out = new vdf_ctx(); (*out)->device = device_id; return VDF_OK; }
void vdf_ctx_destroy(vdf_ctx *ctx) { delete ctx; }
int vdf_ctx_device_count(const vdf_ctx *) { return 1; }
int vdf_ctx_device_at(const vdf_ctx *, int) { return 0; }
static int fake_hash(const uint8_t *frames, size_t n, size_t clip_stride, uint64_t *out, uint32_t *crops)
{
    const int c = ++g_concurrent;
    int m = g_concurrent_max.load();
    while (c > m && !g_concurrent_max.compare_exchange_weak(m, c)) {}
    g_calls++;
    std::this_thread::sleep_for(std::chrono::microseconds(150 + 20 * n));
    for (size_t i = 0; i < n; i++) {
        for (int w = 0; w < VDF_HASH_WORDS; w++) out[i * VDF_HASH_WORDS + w] = checksum(frames + i * clip_stride, clip_stride) + (uint64_t)w;
        if (crops) for (int k = 0; k < 4; k++) crops[4 * i + k] = frames[i * clip_stride + (size_t)k];
    }
    --g_concurrent;
    return VDF_OK;
}
int vdf_hash_frames_u8(vdf_ctx *, const uint8_t *frames, size_t n, uint32_t, uint32_t, uint32_t, size_t, size_t clip_stride, uint64_t *out, uint32_t *)
{
    return fake_hash(frames, n, clip_stride, out, nullptr);
}
int vdf_hash_frames_u8_letterbox(vdf_ctx *, const uint8_t *frames, size_t n, uint32_t, uint32_t, uint32_t, size_t, size_t clip_stride, uint64_t *out,
                                 uint32_t *crops, uint32_t *)
{
    return fake_hash(frames, n, clip_stride, out, crops);
}
}

static int run(int threads, uint32_t max_batch, uint32_t wait_us, int letterbox, int per_thread, const char *slots)
{
    if (slots) setenv("VDF_QUEUE_SLOTS", slots, 1); else unsetenv("VDF_QUEUE_SLOTS");
    const uint32_t w = 8, h = 4;
    const size_t clip = (size_t)w * h * 16;
    vdf_ctx *ctx = nullptr;
    vdf_ctx_create(0, &ctx);
    vdf_hash_queue *q = nullptr;
    if (vdf_hash_queue_create(ctx, w, h, max_batch, wait_us, letterbox, &q) != VDF_OK) return 1;
    std::atomic<int> wrong{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::mt19937 rng(100 + t);
            std::vector<uint8_t> c(clip);
            for (int k = 0; k < per_thread; k++) {
                for (auto &b : c) b = (uint8_t)rng();
                uint64_t out[VDF_HASH_WORDS];
                uint32_t crop[4] = {9, 9, 9, 9};
                if (vdf_hash_queue_submit(q, c.data(), out, crop) != VDF_OK) { wrong++; return; }
                const uint64_t want = checksum(c.data(), clip);
                for (int i = 0; i < VDF_HASH_WORDS; i++) if (out[i] != want + (uint64_t)i) wrong++;
                for (int i = 0; i < 4; i++) if (crop[i] != (letterbox ? c[(size_t)i] : 0u)) wrong++;
                if (rng() % 7 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));
            }
        });
    for (auto &x : th) x.join();
    uint64_t nb = 0, nc = 0;
    vdf_hash_queue_stats(q, &nb, &nc);
    uint32_t infl = 0;
    vdf_hash_queue_in_flight_max(q, &infl);
    const uint32_t n_slots = slots ? (uint32_t)std::atoi(slots) : 2u;
    const bool ok = wrong == 0 && nc == (uint64_t)threads * per_thread && infl <= n_slots && (uint32_t)g_concurrent_max.load() <= n_slots;
    std::printf("threads %d max_batch %u wait %u us letterbox %d slots %s: %llu clips in %llu batches, at most %u in flight, %d wrong%s\n", threads, max_batch, wait_us,
                letterbox, slots ? slots : "default", (unsigned long long)nc, (unsigned long long)nb, infl, wrong.load(), ok ? "" : "  <-- FAILED");
    g_concurrent_max = 0;
    vdf_hash_queue_destroy(q);
    vdf_ctx_destroy(ctx);
    return ok ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad += run(48, 4, 200, 0, 60, nullptr);    // many more callers than two batches hold: most sleep for a free slot
    bad += run(48, 4, 0, 1, 60, "1");          // one slot, no waiting for company
    bad += run(16, 64, 300, 1, 80, "4");       // batches that never fill: every leader runs into its deadline
    bad += run(33, 8, 2000, 0, 50, "3");       // a caller count that is no multiple of the batch
    bad += run(2, 1, 0, 0, 200, nullptr);      // batches of one
    bad += run(1, 16, 50, 1, 50, nullptr);     // a single caller
    std::puts(bad ? "queue tsan FAILED" : "queue tsan ok");
    return bad ? 1 : 0;
}
