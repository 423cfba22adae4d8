// Caller-side batching of per-clip hash requests.
//
// The app hashes one file per rayon worker (vid_dup_finder_app/src/video_hash_filesystem_cache/
// video_hash_filesystem_cache.rs:237-257 -> VideoHashBuilder::hash -> gen_hash, video_hash_builder.rs:214-223), which
// would hand the GPU one clip per call.  A queue collects the clips of concurrent callers and runs ONE batched
// launch: the first caller of a batch leads (waits up to max_wait_us for others to join or for the batch to fill,
// runs the batch, publishes the results); the others block until their result is there.  Batches are serial:
// GPU time per batch (microseconds to a few ms) is nothing next to decoding, the point is one launch per batch
// instead of one per clip.  All clips of a queue have the same frame size (one queue per resolution).
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/vdf.h"

struct vdf_hash_queue {
    vdf_ctx *ctx = nullptr;
    uint32_t w = 0, h = 0, max_batch = 0, max_wait_us = 0;
    int letterbox = 0;
    size_t clip_bytes = 0;
    uint8_t *staging = nullptr;  // pinned host memory, max_batch clips of 16 frames
    bool pinned = false;
    std::vector<uint64_t> hashes;
    std::vector<uint32_t> crops;
    std::mutex mu;
    std::condition_variable cv;
    enum { COLLECTING, RUNNING, DRAINING } state = COLLECTING;
    uint32_t count = 0, ready = 0, remaining = 0;
    uint64_t gen = 0, done_gen = ~0ull;
    int batch_rc = VDF_OK;
    uint64_t n_batches = 0, n_clips = 0;
};

extern "C" {

int vdf_hash_queue_create(vdf_ctx *ctx, uint32_t w, uint32_t h, uint32_t max_batch, uint32_t max_wait_us, int letterbox,
                          vdf_hash_queue **out)
{
    if (!ctx || !out || w == 0 || h == 0 || max_batch == 0) return VDF_E_INVAL;
    vdf_hash_queue *q = new (std::nothrow) vdf_hash_queue();
    if (!q) return VDF_E_OOM;
    q->ctx = ctx; q->w = w; q->h = h; q->max_batch = max_batch; q->max_wait_us = max_wait_us; q->letterbox = letterbox;
    q->clip_bytes = (size_t)w * h * VDF_DCT_SIZE;
    (void)hipSetDevice(vdf_ctx_device(ctx));
    if (hipHostMalloc((void **)&q->staging, q->clip_bytes * max_batch, hipHostMallocDefault) == hipSuccess) {
        q->pinned = true;
    } else {
        (void)hipGetLastError();
        q->staging = (uint8_t *)std::malloc(q->clip_bytes * max_batch);
        if (!q->staging) { delete q; return VDF_E_OOM; }
    }
    q->hashes.resize((size_t)max_batch * VDF_HASH_WORDS);
    q->crops.resize((size_t)max_batch * 4);
    *out = q;
    return VDF_OK;
}

void vdf_hash_queue_destroy(vdf_hash_queue *q)
{
    if (!q) return;
    if (q->pinned) (void)hipHostFree(q->staging); else std::free(q->staging);
    delete q;
}

// frames: 16 gray frames of w x h, tightly packed (the builder's output contract).  Blocks until hashed.
int vdf_hash_queue_submit(vdf_hash_queue *q, const uint8_t *frames, uint64_t *out_hash, uint32_t *out_crop)
{
    if (!q || !frames || !out_hash) return VDF_E_INVAL;
    std::unique_lock<std::mutex> lk(q->mu);
    q->cv.wait(lk, [&] { return q->state == vdf_hash_queue::COLLECTING && q->count < q->max_batch; });
    const uint32_t my = q->count++;
    const uint64_t my_gen = q->gen;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(q->max_wait_us);
    lk.unlock();
    std::memcpy(q->staging + (size_t)my * q->clip_bytes, frames, q->clip_bytes);  // outside the lock: callers copy in parallel
    lk.lock();
    q->ready++;
    if (my == 0) {
        // leader: give others until the deadline (counted from the first arrival) or until the batch is full
        q->cv.wait_until(lk, deadline, [&] { return q->count == q->max_batch; });
        q->state = vdf_hash_queue::RUNNING;                        // no more joins
        q->cv.wait(lk, [&] { return q->ready == q->count; });      // every joined caller has finished its copy
        const uint32_t n = q->count;
        lk.unlock();
        int rc;
        if (q->letterbox)
            rc = vdf_hash_frames_u8_letterbox(q->ctx, q->staging, n, VDF_DCT_SIZE, q->w, q->h, (size_t)q->w * q->h,
                                              q->clip_bytes, q->hashes.data(), q->crops.data(), nullptr);
        else
            rc = vdf_hash_frames_u8(q->ctx, q->staging, n, VDF_DCT_SIZE, q->w, q->h, (size_t)q->w * q->h, q->clip_bytes,
                                    q->hashes.data(), nullptr);
        lk.lock();
        if (!q->letterbox) std::memset(q->crops.data(), 0, (size_t)n * 16);
        q->batch_rc = rc;
        q->done_gen = my_gen;
        q->remaining = n;
        q->state = vdf_hash_queue::DRAINING;
        q->n_batches++;
        q->n_clips += n;
        q->cv.notify_all();
    } else {
        q->cv.notify_all();                                        // the leader may be waiting for count / ready
        q->cv.wait(lk, [&] { return q->done_gen == my_gen && q->state == vdf_hash_queue::DRAINING; });
    }
    const int rc = q->batch_rc;
    if (rc == VDF_OK) {
        std::memcpy(out_hash, q->hashes.data() + (size_t)my * VDF_HASH_WORDS, VDF_HASH_WORDS * 8);
        if (out_crop) std::memcpy(out_crop, q->crops.data() + (size_t)my * 4, 16);
    }
    if (--q->remaining == 0) {  // last one out reopens the queue
        q->count = 0;
        q->ready = 0;
        q->gen++;
        q->state = vdf_hash_queue::COLLECTING;
        q->cv.notify_all();
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

}  // extern "C"
