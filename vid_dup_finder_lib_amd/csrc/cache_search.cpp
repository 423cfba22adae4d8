// vdf_search_cache_entries: from the SoA arrays of a decoded hash cache to MatchGroups in one call (SURVEY.md section 8f,
// row N1 -> a7 -> a8 / a9).  What the app does between loading its cache and printing groups
// (vid_dup_finder_app/src/app/app_fns.rs:428-482: all_cached_paths -> the --files / --with-refs filters -> cache.fetch per path ->
// search() or search_with_references(), whose Search::new sorts by (duration, src_path): search_algorithm.rs:31-34,55-61) without
// one host object per entry: the only per-entry host work is vdf_path_ranks (multi-threaded, allocation-free) and two 4-byte
// gathers; hashes go up once and are put into Search::sort order on the device.
#include <chrono>
#include <cstring>
#include <new>

#include "vdf_ctx.h"

using namespace vdf_impl;

namespace {

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Scratch {  // device buffers of one call, released on every exit path
    DevBuf all_h, sel, sel_h, sel_d, sel_rank, perm, blob, off, facts;
    ~Scratch()
    {
        for (DevBuf *b : {&all_h, &sel, &sel_h, &sel_d, &sel_rank, &perm, &blob, &off, &facts}) b->release();
    }
};

}  // namespace

static int search_cache_entries_impl(vdf_ctx *ctx, const uint64_t *hashes, const uint32_t *durations, const uint64_t *path_offsets,
                                     const char *paths, size_t n, const uint64_t *cand_idx, size_t n_cand, const uint64_t *ref_idx,
                                     size_t n_ref, uint32_t tol_int, vdf_groups *out, vdf_cache_search_timing *timing);

extern "C" int vdf_search_cache_entries(vdf_ctx *ctx, const uint64_t *hashes, const uint32_t *durations, const uint64_t *path_offsets,
                                        const char *paths, size_t n, const uint64_t *cand_idx, size_t n_cand, const uint64_t *ref_idx,
                                        size_t n_ref, uint32_t tol_int, vdf_groups *out, vdf_cache_search_timing *timing)
{
    if (!ctx || !out) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    try {
        return search_cache_entries_impl(ctx, hashes, durations, path_offsets, paths, n, cand_idx, n_cand, ref_idx, n_ref, tol_int, out, timing);
    } catch (const std::bad_alloc &) {  // the host staging vectors: nothing may be thrown through the C ABI
        vdf_groups_free(out);
        return fail(ctx, VDF_E_OOM, "host staging");
    }
}

static int search_cache_entries_impl(vdf_ctx *ctx, const uint64_t *hashes, const uint32_t *durations, const uint64_t *path_offsets,
                                     const char *paths, size_t n, const uint64_t *cand_idx, size_t n_cand, const uint64_t *ref_idx,
                                     size_t n_ref, uint32_t tol_int, vdf_groups *out, vdf_cache_search_timing *timing)
{
    std::memset(out, 0, sizeof *out);
    if (timing) std::memset(timing, 0, sizeof *timing);
    ctx->stats = vdf_search_stats{};
    const double t0 = now_ms();
    const size_t nc = cand_idx ? n_cand : n;
    const bool refs = n_ref != 0;
    if (nc == 0 || n == 0) return refs ? vdf_groups_from_ref_hits(nullptr, 0, out) : vdf_groups_finish_self(out);
    if (!hashes || !durations || !path_offsets || !paths || (refs && !ref_idx)) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n >= 0xFFFFFFFFull || n_ref >= 0xFFFFFFFFull || nc >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    if (cand_idx)
        for (size_t i = 0; i < nc; i++)
            if (cand_idx[i] >= n) return fail(ctx, VDF_E_INVAL, "candidate index out of range");
    for (size_t i = 0; i < n_ref; i++)
        if (ref_idx[i] >= n) return fail(ctx, VDF_E_INVAL, "reference index out of range");

    // Host staging first, device scratch second, the stream drain last: whatever path leaves this function, the streams are drained
    // BEFORE the buffers an asynchronous copy may still be reading or writing go away (destruction runs in reverse order).
    std::vector<uint32_t> rank, sel32, sel_dur, sel_rank, perm, ref_d;
    std::vector<uint64_t> ref_h;
    Scratch sc;
    struct Drain {
        vdf_ctx *ctx;
        ~Drain()
        {
            for (int k = 0; k < device_count(ctx); k++) {
                vdf_ctx *d = device_ctx(ctx, k);
                if (hipSetDevice(d->device) == hipSuccess) (void)hipStreamSynchronize(d->stream);
            }
            (void)hipGetLastError();
        }
    };
    DeviceGuard restore_device;  // (destroyed after `drain`, which moves the calling thread over the devices)
    Drain drain{ctx};

    // ---- Search::sort order of the candidates: (duration, PathBuf order of the path)  (search_algorithm.rs:55-61).
    // Round 6: on the DEVICE when the paths are plain (sort_order.hip: an LSD radix sort over 8-byte words of the paths, then the stable
    // duration sort) - the blob goes up once (0.55 GB for 10 M paths) instead of 0.2 - 0.3 s of sample sort on 32 host threads.  Paths
    // that are not plain (an empty, "." or ".." component somewhere: nothing a directory walk produces) or longer than kMaxDevicePath
    // bytes take the host's component comparator (vdf_path_ranks) and the rank-keyed device sort, as before.
    constexpr uint32_t kMaxDevicePath = 1024;
    const int G = device_count(ctx);
    vdf_ctx *d0 = device_ctx(ctx, 0);
    VDF_HIP(ctx, hipSetDevice(d0->device));
    hipStream_t s = d0->stream;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        int r = upload(d0, b, src, bytes, s);
        if (r && d0 != ctx) ctx->err = d0->err;
        return r;
    };
    for (size_t i = 0; i < n; i++)
        if (path_offsets[i + 1] < path_offsets[i]) return fail(ctx, VDF_E_INVAL, "path offsets are not ascending");
    int rc = VDF_OK;
    if (cand_idx) {
        sel32.resize(nc); sel_dur.resize(nc);
        for (size_t i = 0; i < nc; i++) {
            const uint32_t e = (uint32_t)cand_idx[i];
            sel32[i] = e; sel_dur[i] = durations[e];
        }
        if ((rc = up(sc.sel, sel32.data(), nc * 4))) return rc;
    }
    if ((rc = up(sc.sel_d, cand_idx ? sel_dur.data() : durations, nc * 4))) return rc;
    bool on_device = !ctx->no_device_path_order;
    uint32_t facts[4] = {0, 0, 0, 0};  // not_plain, max_len, shared, pad
    if (on_device) {
        const size_t blob_bytes = (size_t)(path_offsets[n] - path_offsets[0]);
        if ((rc = up(sc.blob, paths + path_offsets[0], std::max<size_t>(blob_bytes, 1)))) return rc;
        if ((rc = up(sc.off, path_offsets, (n + 1) * 8))) return rc;
        VDF_HIP(ctx, sc.facts.reserve(16));
        // (the offsets are relative to `paths`; the uploaded blob starts at paths + path_offsets[0])
        const char *d_blob = sc.blob.as<char>() - path_offsets[0];
        VDF_HIP(ctx, vdf::launch_path_facts(d_blob, sc.off.as<unsigned long long>(), cand_idx ? sc.sel.as<uint32_t>() : nullptr, (uint32_t)nc,
                                            sc.facts.p, s));
        VDF_HIP(ctx, hipMemcpyAsync(facts, sc.facts.p, 16, hipMemcpyDeviceToHost, s));
        VDF_HIP(ctx, hipStreamSynchronize(s));
        on_device = facts[0] == 0 && facts[1] <= kMaxDevicePath;
    }
    if (!on_device) {
        // PathBuf ranks over all n entries order any subset of them too (the cost is that of all n whatever the selection)
        rank.resize(n);
        rc = vdf_path_ranks(paths, path_offsets, n, rank.data(), 0);
        if (rc) return fail(ctx, rc, "path offsets are not ascending");
        if (cand_idx) {
            sel_rank.resize(nc);
            for (size_t i = 0; i < nc; i++) sel_rank[i] = rank[sel32[i]];
        }
    }
    const double t_rank = now_ms();  // (device route: the uploads of blob, offsets and durations and the facts pass; host route: the ranks)

    // ---- hashes up; the order
    if ((rc = up(sc.all_h, hashes, n * VDF_HASH_WORDS * 8))) return rc;
    const uint64_t *d_src_h = sc.all_h.as<uint64_t>();
    if (cand_idx) {
        VDF_HIP(ctx, sc.sel_h.reserve(nc * VDF_HASH_WORDS * 8));
        VDF_HIP(ctx, vdf::launch_gather_hashes(sc.all_h.as<uint64_t>(), nullptr, sc.sel.as<uint32_t>(), (uint32_t)nc, sc.sel_h.as<uint64_t>(),
                                               nullptr, s));
        d_src_h = sc.sel_h.as<uint64_t>();
    }
    if (!on_device && (rc = up(sc.sel_rank, cand_idx ? sel_rank.data() : rank.data(), nc * 4))) return rc;
    VDF_HIP(ctx, hipStreamSynchronize(s));  // (the host vectors may go; and the phase is timed)
    const double t_up = now_ms();
    VDF_HIP(ctx, sc.perm.reserve(nc * 4));
    if (on_device) {
        const char *d_blob = sc.blob.as<char>() - path_offsets[0];
        VDF_HIP(ctx, d0->sort_scratch.reserve(vdf::path_order_scratch_bytes((uint32_t)nc)));
        VDF_HIP(ctx, vdf::launch_path_duration_order(d_blob, sc.off.as<unsigned long long>(), cand_idx ? sc.sel.as<uint32_t>() : nullptr,
                                                     sc.sel_d.as<uint32_t>(), (uint32_t)nc, facts[2] / 8, (facts[1] + 7) / 8, sc.perm.as<uint32_t>(),
                                                     d0->sort_scratch.p, d0->sort_scratch.cap, s));
    } else {
        VDF_HIP(ctx, d0->sort_scratch.reserve(vdf::sort_order_scratch_bytes((uint32_t)nc, true)));
        VDF_HIP(ctx, vdf::launch_sort_order(sc.sel_d.as<uint32_t>(), sc.sel_rank.as<uint32_t>(), (uint32_t)nc, sc.perm.as<uint32_t>(),
                                            d0->sort_scratch.p, d0->sort_scratch.cap, s));
    }
    VDF_HIP(ctx, d0->up_hashes.reserve(nc * VDF_HASH_WORDS * 8));
    VDF_HIP(ctx, d0->up_dur.reserve(std::max<size_t>(nc * 4, 16)));
    VDF_HIP(ctx, vdf::launch_gather_hashes(d_src_h, sc.sel_d.as<uint32_t>(), sc.perm.as<uint32_t>(), (uint32_t)nc, d0->up_hashes.as<uint64_t>(),
                                           d0->up_dur.as<uint32_t>(), s));
    perm.resize(nc);
    VDF_HIP(ctx, hipMemcpyAsync(perm.data(), sc.perm.p, nc * 4, hipMemcpyDeviceToHost, s));
    // the other devices of a multi-GPU context receive the sorted database from the first (device-to-device, on its stream)
    for (int k = 1; k < G; k++) {
        vdf_ctx *d = device_ctx(ctx, k);
        VDF_HIP(ctx, hipSetDevice(d->device));
        VDF_HIP(ctx, d->up_hashes.reserve(nc * VDF_HASH_WORDS * 8));
        VDF_HIP(ctx, d->up_dur.reserve(std::max<size_t>(nc * 4, 16)));
        VDF_HIP(ctx, hipSetDevice(d0->device));
        VDF_HIP(ctx, hipMemcpyAsync(d->up_hashes.p, d0->up_hashes.p, nc * VDF_HASH_WORDS * 8, hipMemcpyDefault, s));
        VDF_HIP(ctx, hipMemcpyAsync(d->up_dur.p, d0->up_dur.p, nc * 4, hipMemcpyDefault, s));
    }
    // references: gathered on the host in ref_idx order, split contiguously over the devices (as vdf_search_refs does)
    std::vector<size_t> cnt((size_t)G, 0), base((size_t)G, 0);
    if (refs) {
        ref_h.resize(n_ref * VDF_HASH_WORDS);
        ref_d.resize(n_ref);
        for (size_t i = 0; i < n_ref; i++) {
            std::memcpy(&ref_h[i * VDF_HASH_WORDS], hashes + ref_idx[i] * VDF_HASH_WORDS, VDF_HASH_WORDS * 8);
            ref_d[i] = durations[ref_idx[i]];
        }
        for (int k = 0; k < G; k++) {
            const size_t b = n_ref / (size_t)G, rem = n_ref % (size_t)G;
            base[(size_t)k] = (size_t)k * b + std::min<size_t>((size_t)k, rem);
            cnt[(size_t)k] = b + ((size_t)k < rem ? 1 : 0);
            vdf_ctx *d = device_ctx(ctx, k);
            VDF_HIP(ctx, hipSetDevice(d->device));
            int r = upload(d, d->up_ref_hashes, ref_h.data() + base[(size_t)k] * VDF_HASH_WORDS, cnt[(size_t)k] * VDF_HASH_WORDS * 8, d->stream);
            if (r == VDF_OK) r = upload(d, d->up_ref_dur, ref_d.data() + base[(size_t)k], cnt[(size_t)k] * 4, d->stream);
            if (r) { if (d != ctx) ctx->err = d->err; return r; }
        }
    }
    for (int k = 0; k < G; k++) {  // everything the search reads is in place before the devices' own streams start
        vdf_ctx *d = device_ctx(ctx, k);
        VDF_HIP(ctx, hipSetDevice(d->device));
        VDF_HIP(ctx, hipStreamSynchronize(d->stream));
    }
    const double t_sort = now_ms();

    rc = refs ? search_refs_resident(ctx, nc, cnt, base, tol_int, out) : search_self_resident(ctx, nc, tol_int, out);
    if (rc) { vdf_groups_free(out); return rc; }
    const double t_search = now_ms();

    // ---- members: sorted position -> the caller's entry
    const uint64_t n_members = out->n_groups ? out->offsets[out->n_groups] : 0;
    for (uint64_t i = 0; i < n_members; i++) {
        const uint32_t at = perm[out->members[i]];
        out->members[i] = cand_idx ? cand_idx[at] : at;
    }
    if (refs)
        for (uint64_t g = 0; g < out->n_groups; g++) out->ref_index[g] = (int64_t)ref_idx[out->ref_index[g]];
    const double t_end = now_ms();
    if (timing) {
        timing->rank_ms = (float)(t_rank - t0);
        timing->upload_ms = (float)(t_up - t_rank);
        timing->sort_ms = (float)(t_sort - t_up);
        timing->search_ms = (float)(t_search - t_sort);
        timing->map_ms = (float)(t_end - t_search);
        timing->total_ms = (float)(t_end - t0);
    }
    return VDF_OK;
}

// Search::sort's order of n entries from their durations and paths (host arrays in, host order out): the path half on the device when the
// paths are plain, through the host's component comparator otherwise.  out_order[k] = index of the entry at position k.
extern "C" int vdf_sort_order_paths(vdf_ctx *ctx, const uint32_t *durations, const uint64_t *path_offsets, const char *paths, size_t n,
                                    uint32_t *out_order, int *used_device)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (used_device) *used_device = 0;
    if (n == 0) return VDF_OK;
    if (!durations || !path_offsets || !paths || !out_order) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 entries");
    for (size_t i = 0; i < n; i++)
        if (path_offsets[i + 1] < path_offsets[i]) return fail(ctx, VDF_E_INVAL, "path offsets are not ascending");
    try {
        std::vector<uint32_t> rank;
        Scratch sc;
        DeviceGuard restore_device;
        vdf_ctx *d0 = device_ctx(ctx, 0);
        VDF_HIP(ctx, hipSetDevice(d0->device));
        hipStream_t s = d0->stream;
        struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); (void)hipGetLastError(); } } drain{s};
        auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
            int r = upload(d0, b, src, bytes, s);
            if (r && d0 != ctx) ctx->err = d0->err;
            return r;
        };
        int rc;
        if ((rc = up(sc.sel_d, durations, n * 4))) return rc;
        VDF_HIP(ctx, sc.perm.reserve(n * 4));
        bool on_device = !ctx->no_device_path_order;
        uint32_t facts[4] = {0, 0, 0, 0};
        const char *d_blob = nullptr;
        if (on_device) {
            const size_t blob_bytes = (size_t)(path_offsets[n] - path_offsets[0]);
            if ((rc = up(sc.blob, paths + path_offsets[0], std::max<size_t>(blob_bytes, 1)))) return rc;
            if ((rc = up(sc.off, path_offsets, (n + 1) * 8))) return rc;
            VDF_HIP(ctx, sc.facts.reserve(16));
            d_blob = sc.blob.as<char>() - path_offsets[0];
            VDF_HIP(ctx, vdf::launch_path_facts(d_blob, sc.off.as<unsigned long long>(), nullptr, (uint32_t)n, sc.facts.p, s));
            VDF_HIP(ctx, hipMemcpyAsync(facts, sc.facts.p, 16, hipMemcpyDeviceToHost, s));
            VDF_HIP(ctx, hipStreamSynchronize(s));
            on_device = facts[0] == 0 && facts[1] <= 1024;
        }
        if (on_device) {
            VDF_HIP(ctx, d0->sort_scratch.reserve(vdf::path_order_scratch_bytes((uint32_t)n)));
            VDF_HIP(ctx, vdf::launch_path_duration_order(d_blob, sc.off.as<unsigned long long>(), nullptr, sc.sel_d.as<uint32_t>(), (uint32_t)n,
                                                         facts[2] / 8, (facts[1] + 7) / 8, sc.perm.as<uint32_t>(), d0->sort_scratch.p,
                                                         d0->sort_scratch.cap, s));
        } else {
            rank.resize(n);
            if ((rc = vdf_path_ranks(paths, path_offsets, n, rank.data(), 0))) return fail(ctx, rc, "path offsets are not ascending");
            if ((rc = up(sc.sel_rank, rank.data(), n * 4))) return rc;
            VDF_HIP(ctx, d0->sort_scratch.reserve(vdf::sort_order_scratch_bytes((uint32_t)n, true)));
            VDF_HIP(ctx, vdf::launch_sort_order(sc.sel_d.as<uint32_t>(), sc.sel_rank.as<uint32_t>(), (uint32_t)n, sc.perm.as<uint32_t>(),
                                                d0->sort_scratch.p, d0->sort_scratch.cap, s));
        }
        VDF_HIP(ctx, hipMemcpyAsync(out_order, sc.perm.p, n * 4, hipMemcpyDeviceToHost, s));
        VDF_HIP(ctx, hipStreamSynchronize(s));
        if (used_device) *used_device = on_device ? 1 : 0;
        return VDF_OK;
    } catch (const std::bad_alloc &) {
        return fail(ctx, VDF_E_OOM, "host staging");
    }
}
