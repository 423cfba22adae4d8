// Host frames -> hashes on one device (the body of vdf_hash_frames_u8 / vdf_hash_frames_u8_letterbox).
//
// The caller's frames are ordinary (pageable) memory, any strides; only the first 16 frames of a clip are wanted
// (video_hash.rs:53-61, dct_3d.rs:25).  The kernel side runs at several TB/s, so this path is the PCIe link's:
//   caller memory --(host threads gather 16 frames per clip, tightly packed)--> pinned chunk A / B
//                 --(hipMemcpyAsync on the copy stream)--> device batch buffer 0 / 1
//                 --(hash kernels on the compute stream)--> device hashes --> pinned results --> caller arrays
// Two pinned chunks: the DMA of chunk k runs while the threads fill chunk k + 1.  Two device batch buffers: the
// kernels of batch b run while batch b + 1 streams in.  Results are copied out of pinned memory one batch late, so
// nothing in the loop waits for the GPU except when a buffer is about to be reused.
#include <algorithm>
#include <cstring>

#include "vdf_ctx.h"

namespace vdf_impl {

// A few persistent host threads for the gather into pinned memory (one memcpy thread moves ~10 GB/s; the link
// takes ~56): created on first use, VDF_COPY_THREADS overrides the count.
struct CopyPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    const std::function<void(int, int)> *job = nullptr;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;

    explicit CopyPool(int n)
    {
        for (int t = 0; t < n; t++)
            threads.emplace_back([this, t, n] {
                uint64_t seen = 0;
                for (;;) {
                    const std::function<void(int, int)> *f;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_job.wait(lk, [&] { return quit || gen != seen; });
                        if (quit) return;
                        seen = gen;
                        f = job;
                    }
                    (*f)(t, n);
                    {
                        std::lock_guard<std::mutex> lk(m);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    void run(const std::function<void(int, int)> &f)
    {
        std::unique_lock<std::mutex> lk(m);
        job = &f;
        pending = (int)threads.size();
        gen++;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv_job.notify_all();
        for (auto &t : threads) t.join();
    }
};

void destroy_copy_pool(vdf_ctx *ctx)
{
    delete ctx->copy_pool;
    ctx->copy_pool = nullptr;
}

static CopyPool *pool_of(vdf_ctx *ctx)
{
    if (!ctx->copy_pool) {
        int n = (int)std::thread::hardware_concurrency() / 2;
        n = std::max(1, std::min(n, 8));
        if (ctx->copy_threads) n = ctx->copy_threads;  // VDF_COPY_THREADS
        ctx->copy_pool = new CopyPool(n);
    }
    return ctx->copy_pool;
}

constexpr size_t kBatchBytes = 256ull << 20;  // device batch buffer (x 2)

int hash_host_locked(vdf_ctx *ctx, const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                     size_t clip_stride, int letterbox, uint64_t *out_hashes, uint32_t *out_crops, uint32_t *out_dontcare)
{
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    const size_t fbytes = (size_t)w * h, cbytes = fbytes * VDF_DCT_SIZE;
    const size_t batch = std::max<size_t>(1, std::min<size_t>(n_clips, kBatchBytes / cbytes));
    const size_t kChunkBytes = ctx->host_chunk_bytes;  // pinned staging chunk (x 2); VDF_HOST_CHUNK_MB for experiments
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(batch, kChunkBytes / cbytes));  // clips per pinned chunk
    // Tightly packed input needs no gather: hipMemcpyAsync straight from the caller's pageable memory (the runtime stages
    // it through its own pinned buffers) measured 54.5 GB/s against 53.7 for the library's pinned chunks - both at what
    // the PCIe Gen5 x16 link delivers (tools/sweep_host_path.sh).  VDF_HOST_DIRECT=0 forces the library's staging;
    // strided input (more than 16 frames per clip, padded frames) always takes it: only the wanted bytes cross the link.
    const bool direct_env = ctx->host_direct;
    const bool packed = frame_stride == fbytes && clip_stride == cbytes;
    DevBuf *d_frames[2] = {&ctx->frames, &ctx->frames2}, *d_hash[2] = {&ctx->out_hashes, &ctx->out_hashes2},
           *d_dc[2] = {&ctx->out_dc, &ctx->out_dc2};
    const size_t n_batches = (n_clips + batch - 1) / batch;
    for (int i = 0; i < (n_batches > 1 ? 2 : 1); i++) {
        VDF_HIP(ctx, d_frames[i]->reserve(batch * cbytes));
        VDF_HIP(ctx, d_hash[i]->reserve(batch * VDF_HASH_WORDS * 8));
        VDF_HIP(ctx, d_dc[i]->reserve(batch * 4));
        if (!ctx->pin_out[i].reserve(batch * (VDF_HASH_WORDS * 8 + 4))) return fail(ctx, VDF_E_OOM, "pinned result buffer");
    }
    for (int i = 0; i < 2; i++)  // full-size chunks from the first call on: pinning memory is slow, do it once
        if (!ctx->pin[i].reserve(std::max(kChunkBytes, chunk * cbytes))) return fail(ctx, VDF_E_OOM, "pinned staging buffer");
    CopyPool *pool = pool_of(ctx);
    hipStream_t s = ctx->stream, cs = s;
    if (n_batches > 1 && !ctx->one_stream) {  // the next batch's frames cross the link under this batch's kernels: a stream of their own
        if (!ctx->copy_stream) VDF_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        cs = ctx->copy_stream;
    }

    auto copy_out = [&](size_t b) {  // results of batch b: pinned -> caller arrays (its ev_done has been waited for)
        const size_t c0 = b * batch, nb = std::min(batch, n_clips - c0);
        const uint8_t *src = ctx->pin_out[b & 1].as<uint8_t>();
        std::memcpy(out_hashes + c0 * VDF_HASH_WORDS, src, nb * VDF_HASH_WORDS * 8);
        if (out_dontcare) std::memcpy(out_dontcare + c0, src + batch * VDF_HASH_WORDS * 8, nb * 4);
    };
    size_t chunk_counter = 0;
    for (size_t b = 0; b < n_batches; b++) {
        const size_t c0 = b * batch, nb = std::min(batch, n_clips - c0);
        const int slot = (int)(b & 1);
        if (b >= 2) {  // the slot's previous batch (b - 2): kernels done, results landed in pinned memory
            if (int rcw = wait_event(ctx, ctx->ev_done[slot])) return rcw;
            copy_out(b - 2);
        }
        // stage the batch chunk by chunk: threads fill pinned chunk p while the DMA of the other chunk is in flight
        // Turns on the link: contexts that feed one GPU from the host at the same time (the batching queue's slots) had their transfers
        // interleaved piece by piece by the copy engine - both batches arrive late and together, their callers then refill the staging
        // together with the link idle, and the pattern holds itself: 38 - 42 GB/s where batches taking turns reach the link's 55 - 57
        // (tools/bench_two_ctx.cpp, tools/bench_hash_queue.cpp).  A bulk transfer therefore holds the device's link mutex until its
        // frames are over; a batch below 1 MB is not worth a turn.
        std::unique_lock<std::mutex> turn(link_mutex(ctx->device), std::defer_lock);
        if (direct_env && packed) {
            if (!ctx->no_link_turns && nb * cbytes >= (1u << 20)) turn.lock();
            VDF_HIP(ctx, hipMemcpyAsync(d_frames[slot]->p, frames + c0 * clip_stride, nb * cbytes, hipMemcpyHostToDevice, cs));
            VDF_HIP(ctx, hipEventRecord(ctx->ev_copy[0], cs));
            chunk_counter = 1;
        } else
        for (size_t q0 = 0; q0 < nb; q0 += chunk, chunk_counter++) {
            const size_t nq = std::min(chunk, nb - q0);
            const int p = (int)(chunk_counter & 1);
            if (chunk_counter >= 2)  // the DMA that last read this chunk
                if (int rcw = wait_event(ctx, ctx->ev_copy[p])) return rcw;
            uint8_t *dst = ctx->pin[p].as<uint8_t>();
            const uint8_t *src = frames + (c0 + q0) * clip_stride;
            if (packed) {
                const size_t total = nq * cbytes;
                pool->run([&](int t, int n) {
                    const size_t per = ((total + (size_t)n - 1) / (size_t)n + 4095) & ~(size_t)4095;
                    const size_t lo = std::min(total, per * (size_t)t), hi = std::min(total, lo + per);
                    if (hi > lo) std::memcpy(dst + lo, src + lo, hi - lo);
                });
            } else {
                pool->run([&](int t, int n) {
                    for (size_t c = (size_t)t; c < nq; c += (size_t)n)
                        for (int f = 0; f < VDF_DCT_SIZE; f++)
                            std::memcpy(dst + c * cbytes + (size_t)f * fbytes, src + c * clip_stride + (size_t)f * frame_stride, fbytes);
                });
            }
            VDF_HIP(ctx, hipMemcpyAsync(d_frames[slot]->as<uint8_t>() + q0 * cbytes, dst, nq * cbytes, hipMemcpyHostToDevice, cs));
            VDF_HIP(ctx, hipEventRecord(ctx->ev_copy[p], cs));
        }
        // kernels of this batch wait for its last chunk only (the copy stream is in order)
        VDF_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_copy[(chunk_counter - 1) & 1], 0));
        uint32_t *dc = out_dontcare ? d_dc[slot]->as<uint32_t>() : nullptr;
        int rc;
        auto end_turn = [&]() -> int {  // the frames are over: the link is the next context's
            if (!turn.owns_lock()) return VDF_OK;
            const int rcw = wait_event(ctx, ctx->ev_copy[0]);
            turn.unlock();
            return rcw;
        };
        if (letterbox && (rc = end_turn())) return rc;  // (the letterbox call waits for its detect pass, i.e. for the frames, anyway)
        if (letterbox)
            rc = letterbox_hash_device_locked(ctx, d_frames[slot]->as<uint8_t>(), nb, VDF_DCT_SIZE, w, h, fbytes, cbytes,
                                              d_hash[slot]->as<uint64_t>(), dc, out_crops ? out_crops + 4 * c0 : nullptr, s);
        else
            rc = hash_device_locked(ctx, d_frames[slot]->as<uint8_t>(), nb, VDF_DCT_SIZE, w, h, fbytes, cbytes,
                                    d_hash[slot]->as<uint64_t>(), dc, s);
        if (rc) return rc;
        uint8_t *po = ctx->pin_out[slot].as<uint8_t>();
        VDF_HIP(ctx, hipMemcpyAsync(po, d_hash[slot]->p, nb * VDF_HASH_WORDS * 8, hipMemcpyDeviceToHost, s));
        if (dc) VDF_HIP(ctx, hipMemcpyAsync(po + batch * VDF_HASH_WORDS * 8, dc, nb * 4, hipMemcpyDeviceToHost, s));
        VDF_HIP(ctx, hipEventRecord(ctx->ev_done[slot], s));
        if ((rc = end_turn())) return rc;  // (behind the kernels' launches: they start the moment the frames are in)
        // the next batch's DMA into the other device buffer must not overtake the kernels that still read it
        if (b + 1 < n_batches && b >= 1) VDF_HIP(ctx, hipStreamWaitEvent(cs, ctx->ev_done[(b + 1) & 1], 0));
    }
    for (size_t b = n_batches >= 2 ? n_batches - 2 : 0; b < n_batches; b++) {
        if (int rcw = wait_event(ctx, ctx->ev_done[b & 1])) return rcw;
        copy_out(b);
    }
    return VDF_OK;
}

}  // namespace vdf_impl
