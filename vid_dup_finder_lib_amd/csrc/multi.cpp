// Multi-GPU context behind the C ABI: ONE process, one host thread + one stream per device inside the library.
//
// The crate's search() / search_with_references() are one call in one process
// (vid_dup_finder_lib/src/video_hashing/video_dup_finder.rs:7-13,19-46, called once from
// vid_dup_finder_app/src/app/app_fns.rs:478-482); a Rust caller that holds a multi-GPU context gets all listed GPUs from
// that one call.  How the path shards (DESIGN.md "Multi-GPU"):
//   * hashing: clips are independent -> split contiguously over the devices, no communication;
//   * search(): the sorted database is replicated on every device - straight from the caller's host arrays (no
//     collective needed), or, when the shards are already resident in HBM (hashes just produced on the GPUs), by ONE
//     RCCL all-gather over xGMI (hashes + durations); row tiles of the triangle are dealt round-robin, every device
//     emits the thresholded pairs of its tiles, the host merges and replays the greedy consumption once;
//   * search_with_references(): candidates replicated the same way, references split contiguously.
// RCCL is loaded lazily (dlopen) the first time a collective is needed, so single-GPU users never pay for it.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <set>

#include <rccl/rccl.h>

#include "vdf_ctx.h"

namespace vdf_impl {

struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    const std::function<int(int, vdf_ctx *)> *job = nullptr;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
};

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

struct RcclState {
    RcclApi api;
    std::vector<ncclComm_t> comms;  // one per device of the context, from ncclCommInitAll
};

static int load_rccl(vdf_ctx *ctx, RcclApi &a)
{
    if (a.lib) return VDF_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a.lib) break;
    }
    if (!a.lib) return fail(ctx, VDF_E_RCCL, std::string("cannot load librccl: ") + dlerror());
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(a.lib, name);
        if (!p) ok = false;
        return p;
    };
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(sym("ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
    a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(sym("ncclBroadcast"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        dlclose(a.lib);
        a = RcclApi{};
        return fail(ctx, VDF_E_RCCL, "librccl lacks a required symbol");
    }
    return VDF_OK;
}

#define VDF_NCCL(ctx, st, call)                                                                                  \
    do {                                                                                                         \
        ncclResult_t r__ = (call);                                                                               \
        if (r__ != ncclSuccess) return fail((ctx), VDF_E_RCCL, std::string(#call) + ": " + (st)->api.GetErrorString(r__)); \
    } while (0)

static bool devices_distinct(const vdf_ctx *ctx)
{
    std::set<int> s;
    for (const vdf_ctx *d : ctx->subs) s.insert(d->device);
    return s.size() == ctx->subs.size();
}

// Communicators for the context's device list (once).  RCCL wants every rank on its own device.
static int ensure_comms(vdf_ctx *ctx)
{
    if (!ctx->rccl) ctx->rccl = new RcclState();
    int rc = load_rccl(ctx, ctx->rccl->api);
    if (rc) return rc;
    if (!ctx->rccl->comms.empty()) return VDF_OK;
    std::vector<int> devs;
    for (const vdf_ctx *d : ctx->subs) devs.push_back(d->device);
    ctx->rccl->comms.assign(devs.size(), nullptr);
    ncclResult_t r = ctx->rccl->api.CommInitAll(ctx->rccl->comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) {
        ctx->rccl->comms.clear();
        return fail(ctx, VDF_E_RCCL, std::string("ncclCommInitAll: ") + ctx->rccl->api.GetErrorString(r));
    }
    return VDF_OK;
}

int for_each_device(vdf_ctx *ctx, const std::function<int(int, vdf_ctx *)> &f)
{
    if (ctx->subs.empty()) return f(0, ctx);
    const size_t G = ctx->subs.size();
    for (size_t k = 0; k < G; k++) {
        Worker *w = ctx->workers[k];
        std::lock_guard<std::mutex> lk(w->m);
        w->job = &f;
        w->has_job = true;
        w->done = false;
        w->cv.notify_all();
    }
    // The status reported is the ROOT CAUSE where there is one: a slot that only left because another slot failed (err_secondary, set by the
    // exchange's waiters) yields to the slot that failed on its own, whatever their order in the device list.
    int rc = VDF_OK;
    bool have_primary = false;
    for (size_t k = 0; k < G; k++) {
        Worker *w = ctx->workers[k];
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return w->done; });
        vdf_ctx *d = ctx->subs[k];
        if (w->rc != VDF_OK && (rc == VDF_OK || (!have_primary && !d->err_secondary))) {
            rc = w->rc;
            have_primary = !d->err_secondary;
            ctx->err = "device " + std::to_string(d->device) + " (slot " + std::to_string(k) + "): " + d->err;
        }
        d->err_secondary = false;
    }
    return rc;
}

static void worker_main(vdf_ctx *parent, int k)
{
    Worker *w = parent->workers[(size_t)k];
    vdf_ctx *d = parent->subs[(size_t)k];
    (void)hipSetDevice(d->device);  // the thread stays bound to its device
    for (;;) {
        const std::function<int(int, vdf_ctx *)> *f;
        {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->quit || w->has_job; });
            if (w->quit) return;
            f = w->job;
            w->has_job = false;
        }
        const int rc = (*f)(k, d);
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->rc = rc;
            w->done = true;
        }
        w->cv.notify_all();
    }
}

void destroy_multi(vdf_ctx *ctx)
{
    DeviceGuard restore_device;
    for (Worker *w : ctx->workers) {
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    ctx->workers.clear();
    if (ctx->rccl) {
        for (ncclComm_t c : ctx->rccl->comms)
            if (c) (void)ctx->rccl->api.CommDestroy(c);
        // librccl stays loaded: unloading a library that owns device state at exit is not worth the risk
        delete ctx->rccl;
        ctx->rccl = nullptr;
    }
    for (vdf_ctx *d : ctx->subs) {
        (void)hipSetDevice(d->device);
        (void)hipDeviceSynchronize();
        delete d;
    }
    ctx->subs.clear();
}

// Replicate per-device shards (shard k resident on device k) into every device's `full` buffer, rank order.
//   distinct devices: ONE RCCL collective per array over xGMI - ncclAllGather when the shards are equal, otherwise the
//   same exchange as a group of ncclBroadcasts (one per shard, root = its owner);
//   repeated devices (a device list like {0, 0}: tests on one GPU) or one device: plain device-to-device copies.
// elem_bytes must be a multiple of 4.  Each device's copy is ordered on that device's stream.
// rccl_optional (the replay filter's bitmaps - an optimisation must not make librccl a dependency of host-level search()): when the
// communicators cannot be had (no librccl, ncclCommInitAll fails) the data travels as plain device-to-device copies instead.
// *used_plain (nullable) reports which way it went: after plain copies the sources must stay untouched until every device's copies ran.
static int replicate(vdf_ctx *ctx, const void *const *shards, const size_t *shard_n, size_t elem_bytes,
                     const std::function<void *(vdf_ctx *)> &full, bool rccl_optional = false, bool *used_plain = nullptr)
{
    const size_t G = ctx->subs.size();
    std::vector<size_t> off(G + 1, 0);
    for (size_t k = 0; k < G; k++) off[k + 1] = off[k] + shard_n[k];
    const bool force = ctx->force_rccl;  // VDF_FORCE_RCCL, read when the context was made
    bool use_rccl = (G > 1 || force) && devices_distinct(ctx);
    if (use_rccl) {
        int rc = ensure_comms(ctx);
        if (rc && !rccl_optional) return rc;
        if (rc) { ctx->err.clear(); use_rccl = false; }
    }
    if (used_plain) *used_plain = !use_rccl;
    if (use_rccl) {
        RcclState *st = ctx->rccl;
        bool equal = true;
        for (size_t k = 1; k < G; k++) equal = equal && shard_n[k] == shard_n[0];
        VDF_NCCL(ctx, st, st->api.GroupStart());
        // A failing call must not leave the group open on this thread (some ranks queued, others not): remember the
        // first failure, stop queueing, ALWAYS close the group, then report.
        ncclResult_t first = ncclSuccess;
        const char *what = "";
        for (size_t k = 0; k < G && first == ncclSuccess; k++) {
            vdf_ctx *d = ctx->subs[k];
            char *dst = static_cast<char *>(full(d));
            if (equal) {  // X1 / X2 of SURVEY.md 2b: sendcount = (n / G) * 16 x u64, resp. n / G x u32, as 32-bit words
                first = st->api.AllGather(shards[k], dst, shard_n[k] * elem_bytes / 4, ncclUint32, st->comms[k], d->stream);
                what = "ncclAllGather";
            } else {
                for (size_t r = 0; r < G && first == ncclSuccess; r++)
                    if (shard_n[r]) {
                        first = st->api.Broadcast(shards[r], dst + off[r] * elem_bytes, shard_n[r] * elem_bytes / 4, ncclUint32,
                                                  (int)r, st->comms[k], d->stream);
                        what = "ncclBroadcast";
                    }
            }
        }
        const ncclResult_t ended = st->api.GroupEnd();
        if (first == ncclSuccess && ended != ncclSuccess) { first = ended; what = "ncclGroupEnd"; }
        if (first != ncclSuccess) {
            const std::string msg = std::string(what) + ": " + st->api.GetErrorString(first);
            for (ncclComm_t c : st->comms)  // the communicators' state is unknown now: the next call builds new ones
                if (c) (void)st->api.CommDestroy(c);
            st->comms.clear();
            return fail(ctx, VDF_E_RCCL, msg);
        }
        return VDF_OK;
    }
    for (size_t k = 0; k < G; k++) {
        vdf_ctx *d = ctx->subs[k];
        VDF_HIP(ctx, hipSetDevice(d->device));
        char *dst = static_cast<char *>(full(d));
        for (size_t r = 0; r < G; r++)
            if (shard_n[r])
                VDF_HIP(ctx, hipMemcpyAsync(dst + off[r] * elem_bytes, shards[r], shard_n[r] * elem_bytes, hipMemcpyDefault, d->stream));
    }
    return VDF_OK;
}

// ---- the replay filter's meeting points for the worker threads of ONE multi-GPU context ---------------------------------------
// A reusable barrier whose last arriver runs a step for everybody (sums the hit counts; queues the all-gather of the shards' bitmaps
// on every device's stream - the same replicate() as the database: ONE grouped RCCL collective over xGMI on distinct devices, plain
// device copies when the device list repeats a GPU).  A shard that fails calls abort(): the others return instead of waiting.
struct LocalExchange final : ShardExchange {
    vdf_ctx *parent;
    size_t G;
    std::mutex m;
    std::condition_variable cv;
    size_t arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    int step_rc = VDF_OK;
    int first_rc = VDF_OK;   // the status and message of the shard that failed first (abort): what the waiters report
    std::string first_msg;
    bool exchange_off = false;  // the exchange step itself could not be carried out: every shard goes on WITHOUT the filter
    bool used_plain = false;    // the last bitmap exchange went as plain copies (repeated devices, or RCCL not to be had)
    bool acc_complete = true, res_complete = true;
    uint64_t acc_total = 0, res_total = 0;
    std::vector<const void *> ptrs;
    std::vector<size_t> counts;

    explicit LocalExchange(vdf_ctx *p) : parent(p), G(p->subs.size()), ptrs(p->subs.size(), nullptr), counts(p->subs.size(), 0) {}

    // returns when all G shards have arrived (and the last one has run `last_step`), or the exchange broke
    int meet(vdf_ctx *d, const std::function<int()> &last_step)
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return secondary(d);
        if (++arrived == G) {
            arrived = 0;
            step_rc = last_step ? last_step() : (int)VDF_OK;
            if (step_rc) broken = true;
            generation++;
            cv.notify_all();
            if (step_rc) d->err = parent->err;
            return step_rc;
        }
        const uint64_t gen = generation;
        cv.wait(lk, [&] { return generation != gen || broken; });
        if (broken) return secondary(d);
        return VDF_OK;
    }

    // what a shard that leaves only because ANOTHER one failed reports: the first failure's status and message (m is held)
    int secondary(vdf_ctx *d)
    {
        d->err_secondary = true;
        const int rc = first_rc ? first_rc : step_rc ? step_rc : (int)VDF_E_HIP;
        return fail(d, rc, "another device of the sharded launch failed" + (first_msg.empty() ? std::string() : ": " + first_msg));
    }

    int agree(uint32_t, vdf_ctx *d, bool *all_complete, uint64_t *total_hits) override
    {
        {
            std::lock_guard<std::mutex> lk(m);
            acc_complete = acc_complete && *all_complete;
            acc_total += *total_hits;
        }
        const int rc = meet(d, [&] {
            res_complete = acc_complete; res_total = acc_total;
            acc_complete = true; acc_total = 0;
            return (int)VDF_OK;
        });
        if (rc) return rc;
        *all_complete = res_complete;  // stable until every shard has passed the next meeting point
        *total_hits = res_total;
        return VDF_OK;
    }

    int or_bitmap(uint32_t shard, vdf_ctx *d, uint32_t *d_bitmap, size_t n_words, hipStream_t stream) override
    {
        VDF_HIP(d, d->bitmap_gather.reserve(G * n_words * 4 + 16));
        // peers read this bitmap (device copies on THEIR streams in the repeated-device form): the marking kernels must be done
        VDF_HIP(d, hipStreamSynchronize(stream));
        {
            std::lock_guard<std::mutex> lk(m);
            ptrs[shard] = d_bitmap;
            counts[shard] = n_words;
        }
        int rc = meet(d, [&] {  // one thread queues the exchange for all devices, as for the database
            DeviceGuard restore_device;
            bool plain = false;
            int r = parent->test_exchange_fail ? fail(parent, VDF_E_RCCL, "VDF_TEST_EXCHANGE_FAIL")
                                               : replicate(parent, ptrs.data(), counts.data(), 4, [](vdf_ctx *q) { return q->bitmap_gather.p; },
                                                           /*rccl_optional=*/true, &plain);
            used_plain = plain;
            if (r) {  // the filter is an optimisation: a failed exchange turns it off for this launch, it does not fail the search
                exchange_off = true;
                parent->err.clear();
            }
            return (int)VDF_OK;
        });
        if (rc) return rc;
        if (exchange_off) return kExchangeOff;  // every shard reads the same flag: all go on with their unfiltered lists
        if (used_plain) {
            // plain copies: nobody may change its bitmap before every device's copies of it have finished
            VDF_HIP(d, hipSetDevice(d->device));
            VDF_HIP(d, hipStreamSynchronize(d->stream));
            rc = meet(d, nullptr);
            if (rc) return rc;
        }
        VDF_HIP(d, hipSetDevice(d->device));
        VDF_HIP(d, vdf::launch_bitmap_or(d_bitmap, d->bitmap_gather.as<uint32_t>(), n_words, (uint32_t)G, stream));
        return VDF_OK;
    }

    void abort(int rc, const std::string &msg) override
    {
        std::lock_guard<std::mutex> lk(m);
        if (!broken && first_rc == VDF_OK) { first_rc = rc; first_msg = msg; }
        broken = true;
        cv.notify_all();
    }
};

ShardExchange *make_local_exchange(vdf_ctx *ctx) { return ctx->subs.size() > 1 ? new LocalExchange(ctx) : nullptr; }

}  // namespace vdf_impl

using namespace vdf_impl;

extern "C" {

int vdf_ctx_create_multi(const int *device_ids, int n_devices, vdf_ctx **out)
{
    if (!out) return VDF_E_INVAL;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return VDF_E_INVAL;
    vdf_ctx *parent = new (std::nothrow) vdf_ctx();
    if (!parent) return VDF_E_OOM;
    DeviceGuard restore_device;  // create_single binds the calling thread to each listed device in turn
    parent->device = device_ids[0];
    std::string err;
    for (int k = 0; k < n_devices; k++) {
        vdf_ctx *d = nullptr;
        int rc = create_single(device_ids[k], &d, &err);
        if (rc) {
            set_create_error("device list entry " + std::to_string(k) + ": " + err);  // vdf_last_error(NULL), like vdf_ctx_create
            delete parent;
            return rc;
        }
        parent->subs.push_back(d);
    }
    for (int k = 0; k < n_devices; k++) parent->workers.push_back(new Worker());
    for (int k = 0; k < n_devices; k++) parent->workers[(size_t)k]->th = std::thread(worker_main, parent, k);
    parent->hit_capacity = parent->subs[0]->hit_capacity;
    parent->force_rccl = std::getenv("VDF_FORCE_RCCL") != nullptr;
    parent->test_exchange_fail = std::getenv("VDF_TEST_EXCHANGE_FAIL") != nullptr;
    *out = parent;
    return VDF_OK;
}

int vdf_ctx_device_count(const vdf_ctx *ctx) { return ctx ? device_count(ctx) : 0; }

int vdf_ctx_device_at(const vdf_ctx *ctx, int k)
{
    if (!ctx || k < 0 || k >= device_count(ctx)) return -1;
    return ctx->subs.empty() ? ctx->device : ctx->subs[(size_t)k]->device;
}

int vdf_ctx_device_search_stats(const vdf_ctx *ctx, int k, vdf_search_stats *out)
{
    if (!ctx || !out || k < 0 || k >= device_count(ctx)) return VDF_E_INVAL;
    if (ctx->subs.empty()) { *out = ctx->stats; return VDF_OK; }
    if ((size_t)k >= ctx->dev_stats.size()) { *out = vdf_search_stats{}; return VDF_OK; }
    *out = ctx->dev_stats[(size_t)k];
    return VDF_OK;
}

int vdf_ctx_rccl_ranks(const vdf_ctx *ctx)
{
    if (!ctx || !ctx->rccl) return 0;
    int n = 0;
    for (ncclComm_t c : ctx->rccl->comms) n += c != nullptr;
    return n;
}

int vdf_ctx_device_search_timing(const vdf_ctx *ctx, int k, vdf_search_timing *out)
{
    if (!ctx || !out || k < 0 || k >= device_count(ctx)) return VDF_E_INVAL;
    if (ctx->subs.empty()) { *out = ctx->timing; return VDF_OK; }
    if ((size_t)k >= ctx->dev_timing.size()) { *out = vdf_search_timing{}; return VDF_OK; }
    *out = ctx->dev_timing[(size_t)k];
    return VDF_OK;
}

int vdf_search_self_shards(vdf_ctx *ctx, const uint64_t *const *d_hash_shards, const uint32_t *const *d_dur_shards,
                           const size_t *shard_n, uint32_t tol_int, vdf_groups *out)
{
    if (!ctx || !out) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::memset(out, 0, sizeof *out);
    ctx->stats = vdf_search_stats{};
    if (ctx->subs.empty()) return fail(ctx, VDF_E_INVAL, "the *_shards calls take a context from vdf_ctx_create_multi");
    if (!d_hash_shards || !d_dur_shards || !shard_n) return fail(ctx, VDF_E_INVAL, "null pointer");
    const size_t G = ctx->subs.size();
    size_t n = 0;
    for (size_t k = 0; k < G; k++) {
        if (shard_n[k] && (!d_hash_shards[k] || !d_dur_shards[k])) return fail(ctx, VDF_E_INVAL, "null shard pointer");
        n += shard_n[k];
    }
    if (n == 0) return vdf_groups_finish_self(out);
    if (n >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    DeviceGuard restore_device;
    for (vdf_ctx *d : ctx->subs) {
        VDF_HIP(ctx, hipSetDevice(d->device));
        VDF_HIP(ctx, d->up_hashes.reserve(n * VDF_HASH_WORDS * 8));
        VDF_HIP(ctx, d->up_dur.reserve(std::max<size_t>(n * 4, 16)));
    }
    int rc = replicate(ctx, reinterpret_cast<const void *const *>(d_hash_shards), shard_n, VDF_HASH_WORDS * 8,
                       [](vdf_ctx *d) { return d->up_hashes.p; });
    if (rc == VDF_OK)
        rc = replicate(ctx, reinterpret_cast<const void *const *>(d_dur_shards), shard_n, 4, [](vdf_ctx *d) { return d->up_dur.p; });
    if (rc) return rc;
    return search_self_resident(ctx, n, tol_int, out);  // kernels run on the streams the replication was queued on
}

int vdf_search_refs_shards(vdf_ctx *ctx, const uint64_t *const *d_cand_hash_shards, const uint32_t *const *d_cand_dur_shards,
                           const size_t *cand_shard_n, const uint64_t *const *d_ref_hash_shards,
                           const uint32_t *const *d_ref_dur_shards, const size_t *ref_shard_n, uint32_t tol_int,
                           vdf_groups *out)
{
    if (!ctx || !out) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::memset(out, 0, sizeof *out);
    ctx->stats = vdf_search_stats{};
    if (ctx->subs.empty()) return fail(ctx, VDF_E_INVAL, "the *_shards calls take a context from vdf_ctx_create_multi");
    if (!d_cand_hash_shards || !d_cand_dur_shards || !cand_shard_n || !d_ref_hash_shards || !d_ref_dur_shards || !ref_shard_n)
        return fail(ctx, VDF_E_INVAL, "null pointer");
    const size_t G = ctx->subs.size();
    size_t n_cand = 0, n_ref = 0;
    std::vector<size_t> cnt(G), base(G);
    for (size_t k = 0; k < G; k++) {
        if (cand_shard_n[k] && (!d_cand_hash_shards[k] || !d_cand_dur_shards[k])) return fail(ctx, VDF_E_INVAL, "null shard pointer");
        if (ref_shard_n[k] && (!d_ref_hash_shards[k] || !d_ref_dur_shards[k])) return fail(ctx, VDF_E_INVAL, "null shard pointer");
        base[k] = n_ref;
        cnt[k] = ref_shard_n[k];
        n_cand += cand_shard_n[k];
        n_ref += ref_shard_n[k];
    }
    if (n_cand == 0 || n_ref == 0) return vdf_groups_from_ref_hits(nullptr, 0, out);
    if (n_cand >= 0xFFFFFFFFull || n_ref >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    DeviceGuard restore_device;
    for (size_t k = 0; k < G; k++) {
        vdf_ctx *d = ctx->subs[k];
        VDF_HIP(ctx, hipSetDevice(d->device));
        VDF_HIP(ctx, d->up_hashes.reserve(n_cand * VDF_HASH_WORDS * 8));
        VDF_HIP(ctx, d->up_dur.reserve(std::max<size_t>(n_cand * 4, 16)));
        // the device's own reference slice is used in place: copy it next to the library's other operands (cheap, keeps
        // search_refs_resident uniform with the host-array entry point)
        VDF_HIP(ctx, d->up_ref_hashes.reserve(std::max<size_t>(cnt[k] * VDF_HASH_WORDS * 8, 16)));
        VDF_HIP(ctx, d->up_ref_dur.reserve(std::max<size_t>(cnt[k] * 4, 16)));
        if (cnt[k]) {
            VDF_HIP(ctx, hipMemcpyAsync(d->up_ref_hashes.p, d_ref_hash_shards[k], cnt[k] * VDF_HASH_WORDS * 8, hipMemcpyDeviceToDevice, d->stream));
            VDF_HIP(ctx, hipMemcpyAsync(d->up_ref_dur.p, d_ref_dur_shards[k], cnt[k] * 4, hipMemcpyDeviceToDevice, d->stream));
        }
    }
    int rc = replicate(ctx, reinterpret_cast<const void *const *>(d_cand_hash_shards), cand_shard_n, VDF_HASH_WORDS * 8,
                       [](vdf_ctx *d) { return d->up_hashes.p; });
    if (rc == VDF_OK)
        rc = replicate(ctx, reinterpret_cast<const void *const *>(d_cand_dur_shards), cand_shard_n, 4, [](vdf_ctx *d) { return d->up_dur.p; });
    if (rc) return rc;
    return search_refs_resident(ctx, n_cand, cnt, base, tol_int, out);
}

int vdf_hash_frames_u8_shards(vdf_ctx *ctx, const uint8_t *const *d_frames, const size_t *n_clips, uint32_t frames_per_clip,
                              uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *const *d_out_hashes,
                              uint32_t *const *d_out_dontcare)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->subs.empty()) return fail(ctx, VDF_E_INVAL, "the *_shards calls take a context from vdf_ctx_create_multi");
    if (!d_frames || !n_clips || !d_out_hashes) return fail(ctx, VDF_E_INVAL, "null pointer");
    return for_each_device(ctx, [&](int k, vdf_ctx *d) {
        if (n_clips[k] == 0) return (int)VDF_OK;
        int rc = hash_device_locked(d, d_frames[k], n_clips[k], frames_per_clip, w, h, frame_stride, clip_stride, d_out_hashes[k],
                                    d_out_dontcare ? d_out_dontcare[k] : nullptr, d->stream);
        if (rc) return rc;
        VDF_HIP(d, hipStreamSynchronize(d->stream));  // the hashes are complete when the call returns
        return (int)VDF_OK;
    });
}

}  // extern "C"
