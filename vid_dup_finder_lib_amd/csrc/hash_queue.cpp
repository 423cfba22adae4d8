// Caller-side batching of per-clip hash requests.
//
// The app hashes one file per rayon worker (vid_dup_finder_app/src/video_hash_filesystem_cache/
// video_hash_filesystem_cache.rs:237-257 -> VideoHashBuilder::hash -> gen_hash, video_hash_builder.rs:214-223), which
// would hand the GPU one clip per call.  The queue collects the clips of concurrent callers into batched launches.
//
// Slots: the queue owns two slots per GPU of the context (a multi-GPU context spreads the slots over its devices).  A
// slot is one batch in the making: pinned staging for max_batch clips, a PRIVATE single-device context (own streams and
// scratch, so batches of different slots run concurrently), and the state COLLECTING -> RUNNING -> DRAINING.  Arrivals
// join the slot that is collecting; its first caller leads: it waits up to max_wait_us (from its own arrival) for others
// or until the slot is full, closes the slot, hashes the batch, publishes the results.  The moment a slot closes the next
// free slot starts collecting, so new arrivals copy their frames in - and may even start their own batch - while the
// previous batch is still on the GPU (at 1080p a clip is 33 MB: the staging copies are the long part).  All clips of a
// queue have the same frame size (one queue per resolution).
//
// Who wakes whom (round 6; one mutex, but every wait has its own condition variable): the first form had ONE condition variable
// for everything and every joiner notified all of it, so with T callers each batch cost ~max_batch wake-ups of ~T threads that
// re-took the mutex to find nothing had changed for them - 64 callers at 1080p got 26 GB/s where 16 got the link's 57, and
// 64 x 64 clips 27 k clips/s (tools/bench_hash_queue.cpp, profiles/r06_hash_queue.txt).  Now
//   * arrivals that find no collecting slot sleep on the queue's cv_free; a slot that reopens wakes at most max_batch of them;
//   * a slot's leader sleeps on the slot's cv_leader; a joiner wakes it only when its join fills the batch or its copy is the
//     last one the closed batch waits for;
//   * joiners sleep on the slot's cv_done; the leader wakes them once, when the results are in.
// VDF_QUEUE_SLOTS (read when a queue is made): slots per GPU, default 2; more slots keep more batches in flight when there are
// many more callers than max_batch (each slot pins max_batch clips of staging; beyond three slots per GPU the slots' streams and the
// parent's outnumber HIP's four default hardware queues again - GPU_MAX_HW_QUEUES raises those).
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>

#include "vdf_ctx.h"

namespace {

// The leader's deadline runs on the steady clock.  (-DVDF_QUEUE_SYSTEM_CLOCK: tests/test_host_sanitizers.py builds this file under
// ThreadSanitizer, whose runtime in gcc 11 does not know pthread_cond_clockwait - the steady-clock wait - and then misses the mutex
// hand-over inside it; the system-clock wait is one it knows.)
#ifdef VDF_QUEUE_SYSTEM_CLOCK
using QueueClock = std::chrono::system_clock;
#else
using QueueClock = std::chrono::steady_clock;
#endif

struct Slot {
    vdf_ctx *ctx = nullptr;      // private context of this slot
    uint8_t *staging = nullptr;  // pinned host memory, max_batch clips of 16 frames
    bool pinned = false;
    std::vector<uint64_t> hashes;
    std::vector<uint32_t> crops;
    enum { COLLECTING, RUNNING, DRAINING } state = COLLECTING;
    uint32_t count = 0, ready = 0, remaining = 0;
    uint64_t gen = 0, done_gen = ~0ull;
    int batch_rc = VDF_OK;
    std::condition_variable cv_leader;  // the leader: the batch is full / every joined copy has finished
    std::condition_variable cv_done;    // the joiners: the batch's results are in
};

}  // namespace

struct vdf_hash_queue {
    uint32_t w = 0, h = 0, max_batch = 0, max_wait_us = 0;
    int letterbox = 0;
    size_t clip_bytes = 0;
    std::vector<Slot> slots;
    size_t cur = 0;  // the slot new arrivals join
    std::mutex mu;
    std::condition_variable cv_free;  // arrivals: some slot collects again
    uint32_t waiting_free = 0;        // ... how many sleep there
    uint64_t n_batches = 0, n_clips = 0;
    uint32_t in_flight = 0, in_flight_max = 0;
};

extern "C" {

void vdf_hash_queue_destroy(vdf_hash_queue *q)
{
    if (!q) return;
    for (Slot &s : q->slots) {
        if (s.staging) { if (s.pinned) (void)hipHostFree(s.staging); else std::free(s.staging); }
        if (s.ctx) vdf_ctx_destroy(s.ctx);
    }
    delete q;
}

int vdf_hash_queue_create(vdf_ctx *ctx, uint32_t w, uint32_t h, uint32_t max_batch, uint32_t max_wait_us, int letterbox,
                          vdf_hash_queue **out)
{
    if (!ctx || !out || w == 0 || h == 0 || max_batch == 0) return VDF_E_INVAL;
    *out = nullptr;
    vdf_hash_queue *q = new (std::nothrow) vdf_hash_queue();
    if (!q) return VDF_E_OOM;
    q->w = w; q->h = h; q->max_batch = max_batch; q->max_wait_us = max_wait_us; q->letterbox = letterbox;
    q->clip_bytes = (size_t)w * h * VDF_DCT_SIZE;
    const int n_dev = vdf_ctx_device_count(ctx);
    int per_gpu = 2;
    if (const char *e = std::getenv("VDF_QUEUE_SLOTS")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 16) per_gpu = v;
    }
    q->slots = std::vector<Slot>((size_t)(per_gpu * n_dev));
    vdf_impl::DeviceGuard restore_device;  // the loop below visits every device of the context on the caller's thread
    for (size_t k = 0; k < q->slots.size(); k++) {
        Slot &s = q->slots[k];
        const int dev = vdf_ctx_device_at(ctx, (int)(k % (size_t)n_dev));
        int rc = vdf_ctx_create(dev, &s.ctx);
        if (rc) { vdf_hash_queue_destroy(q); return rc; }
        s.ctx->one_stream = true;  // one stream per slot: the slots' streams and the parent's fit HIP's four hardware queues (vdf_ctx.h)
        (void)hipSetDevice(dev);
        if (hipHostMalloc((void **)&s.staging, q->clip_bytes * max_batch, hipHostMallocDefault) == hipSuccess) {
            s.pinned = true;
        } else {
            (void)hipGetLastError();
            s.staging = (uint8_t *)std::malloc(q->clip_bytes * max_batch);
            if (!s.staging) { vdf_hash_queue_destroy(q); return VDF_E_OOM; }
        }
        s.hashes.resize((size_t)max_batch * VDF_HASH_WORDS);
        s.crops.resize((size_t)max_batch * 4);
    }
    *out = q;
    return VDF_OK;
}

// frames: 16 gray frames of w x h, tightly packed (the builder's output contract).  Blocks until hashed.
int vdf_hash_queue_submit(vdf_hash_queue *q, const uint8_t *frames, uint64_t *out_hash, uint32_t *out_crop)
{
    if (!q || !frames || !out_hash) return VDF_E_INVAL;
    std::unique_lock<std::mutex> lk(q->mu);
    // join the collecting slot; if it has closed (or is full), the next slot that is free to collect takes over
    auto collecting = [&]() -> Slot * {
        for (size_t i = 0; i < q->slots.size(); i++) {
            const size_t k = (q->cur + i) % q->slots.size();
            Slot &c = q->slots[k];
            if (c.state == Slot::COLLECTING && c.count < q->max_batch) { q->cur = k; return &c; }
        }
        return nullptr;
    };
    Slot *sp = collecting();
    while (!sp) {
        q->waiting_free++;
        q->cv_free.wait(lk);
        q->waiting_free--;
        sp = collecting();
    }
    Slot &s = *sp;
    const uint32_t my = s.count++;
    const uint64_t my_gen = s.gen;
    const auto deadline = QueueClock::now() + std::chrono::microseconds(q->max_wait_us);
    if (my != 0 && s.count == q->max_batch) s.cv_leader.notify_one();  // this join fills the batch: the leader need not wait for its deadline
    lk.unlock();
    std::memcpy(s.staging + (size_t)my * q->clip_bytes, frames, q->clip_bytes);  // outside the lock: callers copy in parallel
    lk.lock();
    s.ready++;
    if (my == 0) {
        // leader: give others until the deadline (counted from the first arrival) or until the batch is full
        while (s.count < q->max_batch && s.cv_leader.wait_until(lk, deadline) != std::cv_status::timeout) {}
        s.state = Slot::RUNNING;                                   // no more joins here; arrivals move on to the next slot
        while (s.ready < s.count) s.cv_leader.wait(lk);            // every joined caller has finished its copy
        const uint32_t n = s.count;
        q->in_flight++;
        if (q->in_flight > q->in_flight_max) q->in_flight_max = q->in_flight;
        lk.unlock();
        int rc;
        if (q->letterbox)
            rc = vdf_hash_frames_u8_letterbox(s.ctx, s.staging, n, VDF_DCT_SIZE, q->w, q->h, (size_t)q->w * q->h,
                                              q->clip_bytes, s.hashes.data(), s.crops.data(), nullptr);
        else
            rc = vdf_hash_frames_u8(s.ctx, s.staging, n, VDF_DCT_SIZE, q->w, q->h, (size_t)q->w * q->h, q->clip_bytes,
                                    s.hashes.data(), nullptr);
        lk.lock();
        q->in_flight--;
        if (!q->letterbox) std::memset(s.crops.data(), 0, (size_t)n * 16);
        s.batch_rc = rc;
        s.done_gen = my_gen;
        s.remaining = n;
        s.state = Slot::DRAINING;
        q->n_batches++;
        q->n_clips += n;
        s.cv_done.notify_all();
    } else {
        if (s.state == Slot::RUNNING && s.ready == s.count) s.cv_leader.notify_one();  // the closed batch was waiting for this copy
        while (!(s.done_gen == my_gen && s.state == Slot::DRAINING)) s.cv_done.wait(lk);
    }
    const int rc = s.batch_rc;
    if (rc == VDF_OK) {
        std::memcpy(out_hash, s.hashes.data() + (size_t)my * VDF_HASH_WORDS, VDF_HASH_WORDS * 8);
        if (out_crop) std::memcpy(out_crop, s.crops.data() + (size_t)my * 4, 16);
    }
    if (--s.remaining == 0) {  // last one out reopens the slot
        s.count = 0;
        s.ready = 0;
        s.gen++;
        s.state = Slot::COLLECTING;
        // as many sleepers as the slot has room for (each takes the mutex in turn; the rest sleep on)
        const uint32_t wake = q->waiting_free < q->max_batch ? q->waiting_free : q->max_batch;
        if (wake >= q->waiting_free) q->cv_free.notify_all();
        else
            for (uint32_t i = 0; i < wake; i++) q->cv_free.notify_one();
    }
    return rc;
}

int vdf_hash_queue_stats(vdf_hash_queue *q, uint64_t *n_batches, uint64_t *n_clips)
{
    if (!q) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(q->mu);
    if (n_batches) *n_batches = q->n_batches;
    if (n_clips) *n_clips = q->n_clips;
    return VDF_OK;
}

int vdf_hash_queue_in_flight_max(vdf_hash_queue *q, uint32_t *out)
{
    if (!q || !out) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(q->mu);
    *out = q->in_flight_max;
    return VDF_OK;
}

}  // extern "C"
