// C-ABI layer (include/vdf.h): context, device scratch, launch orchestration, the hit-buffer
// overflow protocol and the host-level convenience calls.  Compiled with hipcc; the kernels live
// in hamming.hip and dct_hash.hip.  There is deliberately NO CPU fallback for the compute path:
// without a usable GPU every compute entry point fails with VDF_E_HIP.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>

#include "host_sort.h"
#include "vdf_ctx.h"

namespace {
thread_local std::string g_create_error;
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

vdf_ctx::~vdf_ctx()
{
    if (!subs.empty() || !workers.empty()) vdf_impl::destroy_multi(this);
    if (copy_pool) vdf_impl::destroy_copy_pool(this);
    for (auto &kv : axis_tables) {
        kv.second->start.release(); kv.second->size.release(); kv.second->w.release();
        delete kv.second;
    }
    for (auto &kv : mfma_tables) {
        kv.second->operand.release(); kv.second->bias.release(); kv.second->meta.release();
        delete kv.second;
    }
    for (auto &kv : box_tables) {
        kv.second->blob.release(); kv.second->entries.release();
        delete kv.second;
    }
    DevBuf *all[] = {&row_lo, &row_hi, &tile_lo, &tile_hi, &tile_first, &tile_count, &tile_offset, &counters,
                     &hits, &perm, &matched, &exp_cols, &exp_rows, &pop_cols, &pop_rows, &cand, &group_cmin, &group_offset, &group_blocks, &up_hashes,
                     &up_dur, &up_ref_hashes, &up_ref_dur, &small, &frames, &frames2, &out_hashes, &out_hashes2, &out_dc,
                     &out_dc2, &cos_table, &crops, &crop_desc, &crop_tables, &crop_desc2, &crop_tables2, &crop_work, &sort_scratch, &sort_scratch_pub, &hits2, &hit_bitmaps, &bitmap_gather};
    for (DevBuf *b : all) b->release();
    for (PinBuf &b : pin) b.release();
    for (PinBuf &b : pin_out) b.release();
    pin_small.release();
    pin_ctrl.release();
    pin_crops.release();
    pin_desc.release();
    host_hits.buf.release();
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (ev_mid) (void)hipEventDestroy(ev_mid);
    if (ev_wait) (void)hipEventDestroy(ev_wait);
    for (hipEvent_t e : ev_copy) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_done) if (e) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (stream) (void)hipStreamDestroy(stream);
}

namespace vdf_impl {

void set_create_error(const std::string &msg) { g_create_error = msg; }

int fail(vdf_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

int fail_hip(vdf_ctx *ctx, hipError_t e, const char *what)
{
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return fail(ctx, e == hipErrorOutOfMemory ? VDF_E_OOM : VDF_E_HIP, msg);
}

bool is_sorted_u32(const uint32_t *d, size_t n)
{
    for (size_t i = 1; i < n; i++)
        if (d[i] < d[i - 1]) return false;
    return true;
}

// Shared core of both searches: windows + tiles, distance kernel, hit download (sorted by (row, col)).
constexpr size_t kPinSmallBytes = 4u << 20;
constexpr uint64_t kDeviceSortHits = 1u << 17;
constexpr uint32_t kRefsMinWorkgroups = 2048;  // a reference search with fewer (tile, chunk) workgroups than this narrows its chunks
constexpr uint64_t kSpecSortHits = 1u << 14;    // hit lists expected to be at least this long are sorted on the device speculatively
constexpr uint64_t kFilterHits = 1u << 16;      // from this many hits on, a replay-only launch drops the rows that cannot become targets  // hit lists from this length on are sorted on the device

int search_core(vdf_ctx *ctx, int mode, const uint64_t *d_col_hashes, const uint32_t *d_col_dur, size_t n_cols,
                const uint64_t *d_row_hashes, const uint32_t *d_row_dur, const uint32_t *d_row_perm, size_t n_rows,
                uint32_t tol_int, uint32_t shard_index, uint32_t shard_count, uint32_t row_begin, uint32_t row_end,
                const uint32_t *d_matched, uint32_t row_index_base, vdf_hit *hits, uint64_t capacity,
                uint64_t *n_hits_out, uint32_t *overflow_row_out, hipStream_t stream, bool replay_only, vdf_ctx::HostHits *staging,
                ShardExchange *fx)
{
    *n_hits_out = 0;
    *overflow_row_out = 0xFFFFFFFFu;
    if (n_rows == 0 || n_cols == 0) return VDF_OK;
    if (n_rows >= 0xFFFFFFFFull || n_cols >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    if (shard_count == 0 || shard_index >= shard_count) return fail(ctx, VDF_E_INVAL, "bad shard index/count");
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    const double t_enter = now_ms();

    vdf::SearchLaunch L{};
    const bool mfma = ctx->search_backend == 1;
    // MFMA kernel: 512-row workgroups for search(); reference searches may take 256-row ones (a tile's candidate range is the
    // union of its rows' duration windows: fewer rows, narrower union)
    L.tile_rows = mfma ? (mode == 1 ? ctx->mfma_refs_rows : ctx->mfma_self_rows) : ctx->tile_rows;
    L.chunk_cols = mfma ? ctx->mfma_chunk_cols : ctx->chunk_cols;
    L.n_row_tiles = (uint32_t)((n_rows + L.tile_rows - 1) / L.tile_rows);
    if (mfma && L.chunk_cols == 0) {
        // Long chunks amortise a workgroup's target loads (65536 columns measured best at 1 M hashes), but a small search
        // must still be cut into enough workgroups to fill and balance 256 CUs: halve (down to 4096) until there are >= 8192 of them
        // (swept at 10 k .. 1 M hashes, tools/sweep_sizes.py).
        L.chunk_cols = 65536;
        while (L.chunk_cols > 4096 && (uint64_t)L.n_row_tiles * ((n_cols + L.chunk_cols - 1) / L.chunk_cols) < ctx->mfma_min_wgs) L.chunk_cols /= 2;
    }
    // (row tile, chunk) workgroups are numbered with 32 bits: widen the chunks rather than refuse very large inputs
    while ((uint64_t)L.n_row_tiles * ((n_cols + L.chunk_cols - 1) / L.chunk_cols) >= 0x40000000ull && L.chunk_cols < (1u << 22))
        L.chunk_cols *= 2;
    const size_t padded_rows = (size_t)L.n_row_tiles * L.tile_rows;
    VDF_HIP(ctx, ctx->row_lo.reserve(padded_rows * 4));
    VDF_HIP(ctx, ctx->row_hi.reserve(padded_rows * 4));
    VDF_HIP(ctx, ctx->tile_lo.reserve((size_t)L.n_row_tiles * 4));
    VDF_HIP(ctx, ctx->tile_hi.reserve((size_t)L.n_row_tiles * 4));
    VDF_HIP(ctx, ctx->tile_first.reserve((size_t)L.n_row_tiles * 4));
    VDF_HIP(ctx, ctx->tile_count.reserve((size_t)L.n_row_tiles * 4));
    VDF_HIP(ctx, ctx->tile_offset.reserve(((size_t)L.n_row_tiles + 1) * 4));
    VDF_HIP(ctx, ctx->counters.reserve(64));
    const uint64_t dev_cap = std::max<uint64_t>(capacity, 1);
    VDF_HIP(ctx, ctx->hits.reserve(dev_cap * sizeof(vdf_hit)));

    L.row_hashes = reinterpret_cast<const uint32_t *>(d_row_hashes);
    L.row_perm = d_row_perm;
    L.n_rows = (uint32_t)n_rows;
    L.row_index_base = row_index_base;
    L.col_hashes = reinterpret_cast<const uint32_t *>(d_col_hashes);
    L.n_cols = (uint32_t)n_cols;
    L.row_lo = ctx->row_lo.as<uint32_t>();
    L.row_hi = ctx->row_hi.as<uint32_t>();
    L.tile_lo = ctx->tile_lo.as<uint32_t>();
    L.tile_hi = ctx->tile_hi.as<uint32_t>();
    L.tile_first = ctx->tile_first.as<uint32_t>();
    L.tile_count = ctx->tile_count.as<uint32_t>();
    L.tile_offset = ctx->tile_offset.as<uint32_t>();
    L.tol = tol_int;
    L.matched = d_matched;
    L.self_mode = (mode == 0);
    L.shard_index = shard_index;
    L.shard_count = shard_count;
    L.hits = ctx->hits.as<vdf_hit>();
    L.capacity = capacity;
    L.counters = ctx->counters.as<unsigned long long>();
    L.overflow_row = reinterpret_cast<uint32_t *>(ctx->counters.as<unsigned long long>() + 4);

    // Early-exit step (both backends): the first 64-bit k-step after which unrelated hashes (partial distance bits / 2 +-
    // sqrt(bits) / 2) are, 4 sigma down, still more than tol apart - then nearly every block stops there.  Exact whatever is
    // picked.  Kernels are instantiated for steps 6, 8, 10, 11, 12, 13, 14 (MFMA) / 6, 10, 12 (VALU): round up; 16 = no test.
    // (tolerance 350 -> step 12 = 832 bits; up to 387 -> 13; up to 417 -> 14 = 960 bits; beyond that unrelated hashes are no longer
    // 4 sigma away from the tolerance after any prefix and the stream runs all 16 steps)
    L.prune_step = 16;
    if (ctx->mfma_prune_step >= 0) {
        L.prune_step = ctx->mfma_prune_step;
    } else {
        for (int st = 6; st <= 14; st++) {
            const double bits = 64.0 * (st + 1);
            if (bits / 2 - 2.0 * std::sqrt(bits) >= (double)tol_int + 1) { L.prune_step = st; break; }
        }
    }
    if (mfma) L.prune_step = L.prune_step <= 6 ? 6 : L.prune_step <= 8 ? 8 : L.prune_step <= 10 ? 10 : L.prune_step <= 14 ? L.prune_step : 16;
    else L.prune_step = L.prune_step <= 6 ? 6 : L.prune_step <= 10 ? 10 : L.prune_step <= 12 ? 12 : 16;
    if (mfma) {
        L.group_size = std::min<uint32_t>(ctx->mfma_group, L.n_row_tiles);
        L.n_groups = (L.n_row_tiles + L.group_size - 1) / L.group_size;
        if (L.n_groups > 1024) return fail(ctx, VDF_E_INVAL, "too many row-tile groups");
        VDF_HIP(ctx, ctx->group_cmin.reserve((size_t)L.n_groups * 4));
        VDF_HIP(ctx, ctx->group_offset.reserve(((size_t)L.n_groups + 1) * 4));
        VDF_HIP(ctx, ctx->group_blocks.reserve((size_t)L.n_groups * 4));
        L.group_blocks = ctx->group_blocks.as<uint32_t>();
        L.group_cmin = ctx->group_cmin.as<uint32_t>();
        L.group_offset = ctx->group_offset.as<uint32_t>();
        // expand both operands to {0, 1} fp4 nibbles (512 B per hash) plus popcount arrays; rows share the column copy in self mode
        const uint32_t k_steps = (uint32_t)(L.prune_step < 15 ? L.prune_step + 1 : 16);  // k-steps of the tested prefix
        const uint32_t col_pad = (uint32_t)((n_cols + vdf::kMfmaRowPad - 1) / vdf::kMfmaRowPad * vdf::kMfmaRowPad) + vdf::kMfmaColPad;
        VDF_HIP(ctx, ctx->exp_cols.reserve((size_t)col_pad * 512));
        VDF_HIP(ctx, ctx->pop_cols.reserve((size_t)col_pad * 12));
        // A database the caller has pinned (vdf_ctx_pin_database: "these bytes will not change") keeps its expansion between
        // searches: 0.15 ms per million hashes that a reference search of 0.9 ms need not pay again.
        const ExpOwner want_owner{d_col_hashes, n_cols, k_steps, ctx->exp_cols.p};
        const bool reuse = ctx->pinned_db == d_col_hashes && ctx->pinned_n == n_cols && ctx->exp_owner == want_owner;
        if (!reuse) {
            ctx->exp_owner = ExpOwner{};
            VDF_HIP(ctx, vdf::launch_expand_fp4(L.col_hashes, (uint32_t)n_cols, col_pad, ctx->exp_cols.p, k_steps,
                                                ctx->pop_cols.as<float>(), stream));
            ctx->exp_owner = want_owner;
        }
        L.col_exp = ctx->exp_cols.p;
        L.col_pop3 = ctx->pop_cols.as<float>();
        L.col_pad = col_pad;
        if (d_row_hashes == d_col_hashes) {
            L.row_exp = ctx->exp_cols.p;
            L.row_pop3 = L.col_pop3;
            L.row_pad = col_pad;
        } else {
            const uint32_t row_pad = (uint32_t)((padded_rows + 127) / 128 * 128);
            VDF_HIP(ctx, ctx->exp_rows.reserve((size_t)row_pad * 512));
            VDF_HIP(ctx, ctx->pop_rows.reserve((size_t)row_pad * 12));
            VDF_HIP(ctx, vdf::launch_expand_fp4(L.row_hashes, (uint32_t)n_rows, row_pad, ctx->exp_rows.p, k_steps,
                                                ctx->pop_rows.as<float>(), stream));
            L.row_exp = ctx->exp_rows.p;
            L.row_pop3 = ctx->pop_rows.as<float>();
            L.row_pad = row_pad;
        }
    }
    // counters[0..2] = 0, overflow_row = UINT32_MAX
    const unsigned long long init[8] = {0, 0, 0, 0, 0xFFFFFFFFull, 0, 0, 0};
    // How much of the hit list is fetched together with the counters (one synchronisation instead of two): about what the
    // previous call produced.  From 16 k pairs on that head is also put into (row, col) order on the device BEFORE its
    // length is known: the slots are pre-filled with 0xFF (sorts last), so if the list fits the head the host receives it
    // sorted - the host radix sort of a 50 k-pair list cost 0.2 ms of a 1.5 ms reference search.
    const bool pin_ok = ctx->pin_small.reserve(kPinSmallBytes) && ctx->pin_small.pinned;
    uint64_t spec = std::min<uint64_t>(capacity, std::max<uint64_t>(ctx->hits_guess + ctx->hits_guess / 4 + 1024, 8192));
    if (!pin_ok || spec * sizeof(vdf_hit) > kPinSmallBytes) spec = 0;  // long lists go straight to the caller's buffer, once their length is known
    const bool spec_sort = spec && ctx->hits_guess >= kSpecSortHits;
    if (spec_sort) VDF_HIP(ctx, hipMemsetAsync(ctx->hits.p, 0xFF, (size_t)spec * sizeof(vdf_hit), stream));
    // A reference search's workgroups are (256-row tile, chunk of its rows' united +-5 % windows): how many there are is only known
    // once the windows are.  With the chunk width picked for a full rectangle the BASELINE configs[4] shape ends up with ~600
    // workgroups of very different lengths for 512 resident slots (kernel 0.88 ms); 8192-column chunks give ~2500 and 0.73 ms
    // (gpurun_out/r03m).  So: if the first pass yields too few, the chunks are halved until there would be enough and the windows /
    // tile kernels run once more (60 us); the width is remembered for the next search of the same shape.  search() keeps its long
    // chunks: its 512-row workgroups pay 208 KB of target loads each, and narrower chunks measured slower at every width.
    const bool adapt_refs = mode == 1 && mfma && ctx->mfma_chunk_cols == 0;
    if (adapt_refs && ctx->refs_hint_cols == n_cols && ctx->refs_hint_rows == n_rows && ctx->refs_hint_chunk) L.chunk_cols = ctx->refs_hint_chunk;
    if (!ctx->pin_ctrl.reserve(256)) return fail(ctx, VDF_E_OOM, "pinned staging");
    unsigned long long *pre = ctx->pin_ctrl.as<unsigned long long>() + 8;
    uint32_t total_tiles = 0;
    unsigned long long unsorted = 0, admitted = 0;
    for (int pass = 0;; pass++) {
        VDF_HIP(ctx, hipMemcpyAsync(ctx->counters.p, init, sizeof init, hipMemcpyHostToDevice, stream));
        VDF_HIP(ctx, vdf::launch_windows_tiles(mode, d_col_dur, (uint32_t)n_cols, d_row_dur, d_row_perm, (uint32_t)n_rows,
                                               row_begin, row_end, shard_index, shard_count, L, stream));
        // workgroup count, sortedness flag and admitted pairs come back through pinned memory (a pageable destination would
        // make each of the copies a blocking round trip of its own)
        VDF_HIP(ctx, hipMemcpyAsync(pre, L.counters, 64, hipMemcpyDeviceToHost, stream));
        VDF_HIP(ctx, hipMemcpyAsync(pre + 8, mfma ? L.group_offset + L.n_groups : L.tile_offset + L.n_row_tiles, 4,
                                    hipMemcpyDeviceToHost, stream));
        // (the grouped grid of the matrix-core backend counts every (tile, chunk) of a group's rectangle; the workgroups that
        // have work are the sum of the tiles' own chunk counts)
        VDF_HIP(ctx, hipMemcpyAsync(pre + 9, L.tile_offset + L.n_row_tiles, 4, hipMemcpyDeviceToHost, stream));
        VDF_HIP(ctx, hipStreamSynchronize(stream));
        total_tiles = *reinterpret_cast<const uint32_t *>(pre + 8);
        const uint32_t busy_tiles = *reinterpret_cast<const uint32_t *>(pre + 9);
        unsorted = pre[5];
        admitted = pre[2];
        if (!adapt_refs || pass > 0 || unsorted || busy_tiles == 0 || busy_tiles >= kRefsMinWorkgroups || L.chunk_cols <= 4096) break;
        uint64_t est = busy_tiles;
        while (L.chunk_cols > 4096 && est < kRefsMinWorkgroups) { L.chunk_cols /= 2; est *= 2; }
    }
    if (adapt_refs) { ctx->refs_hint_cols = n_cols; ctx->refs_hint_rows = n_rows; ctx->refs_hint_chunk = L.chunk_cols; }
    // the windows are binary searches over the candidate durations (search_algorithm.rs:93-117,173-185 rely on Search::sort)
    if (unsorted) return fail(ctx, VDF_E_INVAL, "durations are not ascending: pass the arrays in Search::sort order");
    if (total_tiles >= 0x7FFFFFFFu) return fail(ctx, VDF_E_INVAL, "tile count exceeds the grid limit");
    if (mfma) {
        // Queue of suspect pairs the stream cannot rule out (16-byte entries).  Unrelated hashes put ~2.3e-6 of the admitted
        // pairs there at the 4-sigma test point; slots are handed out in chunks of 8 per wave, so every wave may strand
        // a few.  Sized generously from the admitted pairs; if it still overflows, the hit-buffer overflow protocol takes
        // over (rows below the smallest row that lost a suspect are complete).
        const uint64_t want = ((uint64_t)((double)admitted * 6e-6) + (uint64_t)total_tiles * 8 * 8 + std::max<uint64_t>(capacity, 1ull << 20)) * ctx->cand_scale;
        L.cand_capacity = (uint32_t)std::min<uint64_t>(want, 0x40000000ull);
        if (ctx->cand_capacity_override)  // VDF_CAND_CAPACITY (tests: forces the overflow path)
            L.cand_capacity = (uint32_t)std::min<uint64_t>((uint64_t)ctx->cand_capacity_override * ctx->cand_scale, 0x40000000ull);
        const void *before = ctx->cand.p;
        VDF_HIP(ctx, ctx->cand.reserve((size_t)L.cand_capacity * 16));
        if (ctx->cand.p != before) ctx->cand_dirty = SIZE_MAX;  // fresh allocation: fill all of it
        // Only the slots earlier launches may have touched need the empty pattern (0xFF) again - ALL of them, whatever this
        // launch's capacity is: a launch with a smaller queue (after an overflow retry's x4, or a smaller search) must not
        // leave the slots beyond its own capacity stale for a later, larger launch to read as suspects.
        const size_t slots = ctx->cand.cap / 16;
        const size_t fill = (ctx->cand_dirty == SIZE_MAX ? slots : std::min<size_t>(slots, ctx->cand_dirty)) * 16;
        if (fill) VDF_HIP(ctx, hipMemsetAsync(ctx->cand.p, 0xFF, fill, stream));
        ctx->cand_dirty = L.cand_capacity;  // until the launch reports how far it got (an error return below keeps this bound)
        L.cand = ctx->cand.p;
        L.cand_head = ctx->counters.as<unsigned long long>() + 6;
    }

    ctx->timing.prep_ms += (float)(now_ms() - t_enter);
    VDF_HIP(ctx, hipEventRecord(ctx->ev0, stream));
    if (mfma) {
        VDF_HIP(ctx, vdf::launch_hamming_tiles_mfma2(L, total_tiles, stream));
        VDF_HIP(ctx, hipEventRecord(ctx->ev_mid, stream));
        if (total_tiles) VDF_HIP(ctx, vdf::launch_resolve_candidates(L, stream));
    } else {
        VDF_HIP(ctx, vdf::launch_hamming_tiles(L, total_tiles, stream));
        VDF_HIP(ctx, hipEventRecord(ctx->ev_mid, stream));
    }
    VDF_HIP(ctx, hipEventRecord(ctx->ev1, stream));
    // Counters and - speculatively - the head of the hit list come back behind ONE synchronisation: the list is usually
    // about as long as the previous call's, and a second round trip costs more than copying a few KB too many.
    if (!ctx->pin_ctrl.reserve(256)) return fail(ctx, VDF_E_OOM, "pinned staging");
    unsigned long long *fin = ctx->pin_ctrl.as<unsigned long long>();
    VDF_HIP(ctx, hipMemcpyAsync(fin, ctx->counters.p, 64, hipMemcpyDeviceToHost, stream));
    unsigned row_bits = 1;
    while (row_bits < 32 && ((uint64_t)row_index_base + n_rows) >> row_bits) row_bits++;
    unsigned col_bits = 1;  // columns are candidate indices below n_cols: the hit sort skips the bytes above them
    while (col_bits < 32 && ((uint64_t)n_cols >> col_bits)) col_bits++;
    if (spec_sort) {
        VDF_HIP(ctx, ctx->sort_scratch.reserve(vdf::sort_hits_scratch_bytes((size_t)spec)));
        VDF_HIP(ctx, vdf::launch_sort_hits(ctx->hits.as<vdf_hit>(), (size_t)spec, row_bits, ctx->sort_scratch.p, ctx->sort_scratch.cap, stream, col_bits));
    }
    if (spec) VDF_HIP(ctx, hipMemcpyAsync(ctx->pin_small.p, ctx->hits.p, (size_t)spec * sizeof(vdf_hit), hipMemcpyDeviceToHost, stream));
    VDF_HIP(ctx, hipStreamSynchronize(stream));
    const double t_synced = now_ms();
    float ms = 0.f, ms_stream = 0.f;
    VDF_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    VDF_HIP(ctx, hipEventElapsedTime(&ms_stream, ctx->ev0, ctx->ev_mid));
    ctx->timing.stream_ms += ms_stream;
    ctx->timing.resolve_ms += ms - ms_stream;
    if (mfma) { ctx->timing.suspects += fin[6]; ctx->timing.suspect_capacity = L.cand_capacity; }

    if (mfma) ctx->cand_dirty = (size_t)std::min<uint64_t>(fin[6], L.cand_capacity);  // slots this launch used
    const uint64_t produced = fin[0];
    uint64_t stored = std::min<uint64_t>(produced, capacity);
    ctx->hits_guess = stored;
    uint64_t n_out = produced;
    vdf_hit *d_list = ctx->hits.as<vdf_hit>();
    uint64_t have = std::min(stored, spec);
    // Dense near-duplicates: most of the thresholded pairs belong to rows the greedy replay never uses as targets
    // (hamming.hip: launch_filter_*).  With the COMPLETE hit set of the launch in hand they are dropped here, before the
    // sort, the download and the host replay - a cluster of s mutual duplicates sends down s - 1 pairs instead of
    // s (s - 1) / 2.  A sharded launch has the complete set spread over its shards: they agree that nobody overflowed, and OR
    // the two bitmaps (1 bit per entry: 125 KB per million) between the steps.  EVERY shard takes part, also one without hits.
    bool filter = false;
    if (replay_only && mode == 0 && !ctx->no_hit_filter) {
        bool complete = produced <= capacity && (uint32_t)fin[4] == 0xFFFFFFFFu;
        uint64_t total = produced;
        if (fx) {
            const int r = fx->agree(shard_index, ctx, &complete, &total);
            if (r) return r;
        } else if (shard_count != 1) {
            complete = false;
        }
        filter = complete && total >= kFilterHits;
    }
    if (filter) {
        const size_t words = ((size_t)n_cols + 31) / 32;
        VDF_HIP(ctx, ctx->hits2.reserve((size_t)std::max<uint64_t>(stored, 1) * sizeof(vdf_hit)));
        VDF_HIP(ctx, ctx->hit_bitmaps.reserve(2 * words * 4 + 16));
        uint32_t *bm = ctx->hit_bitmaps.as<uint32_t>();
        VDF_HIP(ctx, vdf::launch_filter_mark_incoming(d_list, stored, (uint32_t)n_cols, bm, stream));
        // (an exchange that could not be carried out says so on every shard alike - kExchangeOff: the launch goes on unfiltered)
        if (fx) { const int r = fx->or_bitmap(shard_index, ctx, bm, words, stream); if (r < 0) return r; if (r == vdf_impl::kExchangeOff) filter = false; }
        if (filter) {
            VDF_HIP(ctx, vdf::launch_filter_mark_covered(d_list, stored, (uint32_t)n_cols, bm, stream));
            if (fx) { const int r = fx->or_bitmap(shard_index, ctx, bm + words, words, stream); if (r < 0) return r; if (r == vdf_impl::kExchangeOff) filter = false; }
        }
    }
    if (filter) {
        uint32_t *bm = ctx->hit_bitmaps.as<uint32_t>();
        VDF_HIP(ctx, vdf::launch_filter_compact(d_list, stored, (uint32_t)n_cols, bm, ctx->hits2.as<vdf_hit>(), L.counters + 7, stream));
        VDF_HIP(ctx, hipMemcpyAsync(fin + 7, L.counters + 7, 8, hipMemcpyDeviceToHost, stream));
        VDF_HIP(ctx, hipStreamSynchronize(stream));
        d_list = ctx->hits2.as<vdf_hit>();
        stored = fin[7];
        n_out = stored;
        have = 0;  // the speculative copy held unfiltered pairs
        ctx->timing.hits_filtered += produced - stored;
    }
    if (stored) {
        if (staging) {  // the library's own staging grows to the list (host-level calls)
            const size_t need = std::max<size_t>((size_t)stored, 1u << 16);
            if (staging->size() < need && !staging->resize(need + need / 4)) return fail(ctx, VDF_E_OOM, "hit staging");
            hits = staging->data();
        }
        if (have) std::memcpy(hits, ctx->pin_small.p, (size_t)have * sizeof(vdf_hit));
        if (stored > have) {
            // Long list: it is put into (row, col) order on the device before it comes down - a host radix sort of 1e7
            // pairs costs about as much as the search kernel.
            const bool dev_sort = stored >= (d_list == ctx->hits.as<vdf_hit>() ? kDeviceSortHits : kDeviceSortHits / 16);  // after the filter nothing of the list is on the host yet
            if (dev_sort) {
                VDF_HIP(ctx, ctx->sort_scratch.reserve(vdf::sort_hits_scratch_bytes((size_t)stored)));
                VDF_HIP(ctx, vdf::launch_sort_hits(d_list, (size_t)stored, row_bits, ctx->sort_scratch.p, ctx->sort_scratch.cap, stream, col_bits));
                VDF_HIP(ctx, hipMemcpyAsync(hits, d_list, (size_t)stored * sizeof(vdf_hit), hipMemcpyDeviceToHost, stream));
            } else {
                VDF_HIP(ctx, hipMemcpyAsync(hits + have, d_list + have, (size_t)(stored - have) * sizeof(vdf_hit), hipMemcpyDeviceToHost, stream));
            }
            VDF_HIP(ctx, hipStreamSynchronize(stream));
            if (!dev_sort) sort_hits(hits, (size_t)stored);
        } else if (!spec_sort) {
            sort_hits(hits, (size_t)stored);
        }
    }
    ctx->timing.download_ms += (float)(now_ms() - t_synced);
    *n_hits_out = n_out;
    *overflow_row_out = (uint32_t)fin[4];
    ctx->stats.pairs += fin[2];
    ctx->stats.pairs_computed += fin[1];
    ctx->stats.n_hits += produced;
    ctx->stats.n_tiles += total_tiles;
    ctx->stats.n_launches += 1;
    ctx->stats.kernel_ms += ms;
    ctx->stats.pairs_early_exit += fin[3];
    ctx->stats.early_exit_bits = L.prune_step < 15 ? 64u * (uint32_t)(L.prune_step + 1) : 0u;
    return VDF_OK;
}

int upload(vdf_ctx *ctx, DevBuf &buf, const void *src, size_t bytes, hipStream_t stream)
{
    VDF_HIP(ctx, buf.reserve(std::max<size_t>(bytes, 16)));
    if (bytes) VDF_HIP(ctx, hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, stream));
    return VDF_OK;
}

std::mutex &link_mutex(int device)
{
    static std::mutex m[64];
    return m[(unsigned)device % 64u];
}

int wait_event(vdf_ctx *ctx, hipEvent_t ev)
{
    if (!ctx->spin_wait) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(50);
        do {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) return VDF_OK;
            if (q != hipErrorNotReady) return fail_hip(ctx, q, "hipEventQuery");
        } while (std::chrono::steady_clock::now() < until);
    }
    VDF_HIP(ctx, hipEventSynchronize(ev));
    return VDF_OK;
}

DeviceAxisTable *axis_table(vdf_ctx *ctx, uint32_t in_size, hipStream_t stream, int *rc)
{
    auto it = ctx->axis_tables.find(in_size);
    if (it != ctx->axis_tables.end()) { *rc = VDF_OK; return it->second; }
    DeviceAxisTable *t = new DeviceAxisTable();
    if (!vdf::build_axis_table(in_size, VDF_DCT_SIZE, t->host)) {
        delete t;
        *rc = fail(ctx, VDF_E_BAD_DIMS, "cannot build resize table");
        return nullptr;
    }
    int r = upload(ctx, t->start, t->host.start.data(), t->host.start.size() * 4, stream);
    if (r == VDF_OK) r = upload(ctx, t->size, t->host.size.data(), t->host.size.size() * 4, stream);
    if (r == VDF_OK) r = upload(ctx, t->w, t->host.w.data(), t->host.w.size() * 2, stream);
    if (r == VDF_OK && hipStreamSynchronize(stream) != hipSuccess) r = fail(ctx, VDF_E_HIP, "table upload failed");
    if (r != VDF_OK) {
        t->start.release(); t->size.release(); t->w.release();
        delete t;
        *rc = r;
        return nullptr;
    }
    ctx->axis_tables[in_size] = t;
    *rc = VDF_OK;
    return t;
}

DeviceMfmaTable *mfma_table(vdf_ctx *ctx, uint32_t in_size, int layout, hipStream_t stream, int *rc)
{
    const uint64_t key = (uint64_t)in_size * 4 + (uint64_t)layout;
    auto it = ctx->mfma_tables.find(key);
    if (it != ctx->mfma_tables.end()) { *rc = VDF_OK; return it->second; }
    DeviceMfmaTable *t = new DeviceMfmaTable();
    if (!vdf::build_mfma_axis_table(in_size, layout, t->host)) {
        delete t;
        *rc = fail(ctx, VDF_E_BAD_DIMS, "cannot build resize table");
        return nullptr;
    }
    int r = VDF_OK;
    if (t->host.ok) {
        r = upload(ctx, t->operand, t->host.operand.data(), t->host.operand.size(), stream);
        if (r == VDF_OK) r = upload(ctx, t->bias, t->host.bias.data(), t->host.bias.size() * 4, stream);
        if (r == VDF_OK && !t->host.band_meta.empty())
            r = upload(ctx, t->meta, t->host.band_meta.data(), t->host.band_meta.size() * 4, stream);
        if (r == VDF_OK && hipStreamSynchronize(stream) != hipSuccess) r = fail(ctx, VDF_E_HIP, "table upload failed");
    }
    if (r != VDF_OK) {
        t->operand.release(); t->bias.release(); t->meta.release();
        delete t;
        *rc = r;
        return nullptr;
    }
    ctx->mfma_tables[key] = t;
    *rc = VDF_OK;
    return t;
}

// The tables of EVERY crop-box size of a w x h frame (vdf_ctx.h: BoxTableSet), built and uploaded once per context and frame size.
BoxTableSet *box_table_set(vdf_ctx *ctx, uint32_t w, uint32_t h, hipStream_t stream, int *rc)
{
    const uint64_t key = ((uint64_t)w << 32) | h;
    auto it = ctx->box_tables.find(key);
    if (it != ctx->box_tables.end()) { *rc = VDF_OK; return it->second; }
    BoxTableSet *set = new BoxTableSet();
    set->one_tile = w <= 64 && h <= 64;
    const size_t n_idx = (size_t)w + h + 2;
    std::vector<vdf::CropTableEntry> entries(n_idx, vdf::CropTableEntry{nullptr, nullptr, 0, 0});
    std::vector<size_t> offset(n_idx, 0);
    std::vector<uint8_t> blob;
    bool usable = true;
    for (size_t idx = 0; idx < n_idx && usable; idx++) {
        const bool vertical = idx > w;
        const uint32_t size = (uint32_t)(vertical ? idx - w - 1 : idx);
        offset[idx] = blob.size();
        if (size == 0) {  // no such box; the one-tile form keeps the stride
            if (set->one_tile) blob.resize(blob.size() + vdf::kSmallBoxTableStride, 0);
            continue;
        }
        vdf::MfmaAxisTable t;
        if (!vdf::build_mfma_axis_table(size, vertical ? vdf::kMfmaLayoutVertical : vdf::kMfmaLayoutHorizontal, t) || !t.ok) { usable = false; break; }
        const size_t at = blob.size();
        blob.resize(at + t.operand.size() + 128, 0);
        std::memcpy(blob.data() + at, t.operand.data(), t.operand.size());
        std::memcpy(blob.data() + at + t.operand.size(), t.bias.data(), 64);
        const int32_t tail[2] = {t.precision, t.n_tiles};
        std::memcpy(blob.data() + at + t.operand.size() + 64, tail, sizeof tail);
        entries[idx].n_tiles = t.n_tiles;
        entries[idx].precision = t.precision;
        if (set->one_tile && blob.size() - at != vdf::kSmallBoxTableStride) usable = false;  // (cannot happen: sizes <= 64 are one tile)
    }
    int r = VDF_OK;
    if (usable) {
        r = upload(ctx, set->blob, blob.data(), blob.size(), stream);
        if (r == VDF_OK) {
            for (size_t idx = 0; idx < n_idx; idx++) {
                if (!entries[idx].n_tiles) continue;
                entries[idx].operand = set->blob.as<uint8_t>() + offset[idx];
                entries[idx].bias = reinterpret_cast<const int32_t *>(set->blob.as<uint8_t>() + offset[idx] + (size_t)entries[idx].n_tiles * 2048);
            }
            r = upload(ctx, set->entries, entries.data(), entries.size() * sizeof(vdf::CropTableEntry), stream);
        }
        // (pageable sources: both copies have left the host buffers when hipMemcpyAsync returns, but the one-time wait keeps this like mfma_table)
        if (r == VDF_OK && hipStreamSynchronize(stream) != hipSuccess) r = fail(ctx, VDF_E_HIP, "table upload failed");
    }
    if (r != VDF_OK) {
        set->blob.release(); set->entries.release();
        delete set;
        *rc = r;
        return nullptr;
    }
    set->usable = usable;
    ctx->box_tables[key] = set;
    *rc = VDF_OK;
    return set;
}

vdf::ResizeAxisTable dev_view(const DeviceAxisTable *t)
{
    vdf::ResizeAxisTable v{};
    if (t) {
        v.start = t->start.as<int32_t>();
        v.size = t->size.as<int32_t>();
        v.w = t->w.as<int16_t>();
        v.window = t->host.window;
        v.precision = t->host.precision;
    }
    return v;
}

int ensure_cos_table(vdf_ctx *ctx, hipStream_t stream)
{
    if (ctx->cos_table.p) return VDF_OK;
    // [k][n] cosine matrix (kept for reference kernels), then the constants of the split-radix DCT-16 in the order
    // dct_hash.hip reads them: 4 + 2 + 1 twiddles (cos, sin) and sqrt(1/2).  Computed exactly as rustdct does
    // (twiddles::single_twiddle(i, len).conj(): angle = (-2 pi / len) * i) so that they equal the oracle's bit for bit.
    double tab[16 * 16 + 17];
    for (int k = 0; k < 16; k++)
        for (int n = 0; n < 16; n++) tab[k * 16 + n] = std::cos(M_PI * (double)k * ((double)n + 0.5) / 16.0);
    int at = 256;
    auto twiddle = [&](int i, int fft_len) {
        const double angle_constant = M_PI * -2.0 / (double)fft_len;
        const double angle = angle_constant * (double)i;
        tab[at++] = std::cos(angle);
        tab[at++] = -std::sin(angle);
    };
    for (int i = 0; i < 4; i++) twiddle(2 * i + 1, 64);
    for (int i = 0; i < 2; i++) twiddle(2 * i + 1, 32);
    twiddle(1, 16);
    tab[at++] = M_SQRT1_2;
    tab[at++] = 0.0; tab[at++] = 0.0;
    int rc = upload(ctx, ctx->cos_table, tab, sizeof tab, stream);
    if (rc) return rc;
    VDF_HIP(ctx, hipStreamSynchronize(stream));
    return VDF_OK;
}

constexpr size_t kMaxClipsPerLaunch = 256 * 1024;  // x 16 frames x 256 threads stays under HIP's 2^32 work-item grid limit

int hash_device_locked(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w,
                       uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *d_out, uint32_t *d_dc,
                       hipStream_t stream)
{
    if (n_clips > kMaxClipsPerLaunch) {
        for (size_t c0 = 0; c0 < n_clips; c0 += kMaxClipsPerLaunch) {
            const size_t nb = std::min(kMaxClipsPerLaunch, n_clips - c0);
            int rc = hash_device_locked(ctx, d_frames + c0 * clip_stride, nb, frames_per_clip, w, h, frame_stride,
                                        clip_stride, d_out + c0 * VDF_HASH_WORDS, d_dc ? d_dc + c0 : nullptr, stream);
            if (rc) return rc;
        }
        return VDF_OK;
    }
    if (frames_per_clip < VDF_DCT_SIZE) return fail(ctx, VDF_E_NOT_ENOUGH_FRAMES, "fewer than 16 frames per clip");
    if (w == 0 || h == 0) return fail(ctx, VDF_E_BAD_DIMS, "zero frame dimension");
    if (frame_stride < (size_t)w * h) return fail(ctx, VDF_E_INVAL, "frame_stride smaller than a frame");
    if (n_clips == 0) return VDF_OK;
    if (n_clips > 0x7FFFFFFull) return fail(ctx, VDF_E_INVAL, "too many clips in one call");
    if (!d_frames || !d_out) return fail(ctx, VDF_E_INVAL, "null pointer");
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    {
        int rc0 = ensure_cos_table(ctx, stream);
        if (rc0) return rc0;
    }
    const bool direct = (w == VDF_DCT_SIZE && h == VDF_DCT_SIZE && ((uintptr_t)d_frames % 16) == 0 &&
                         frame_stride % 16 == 0 && clip_stride % 16 == 0);
    if (direct) {
        VDF_HIP(ctx, vdf::launch_dct_hash(d_frames, clip_stride, frame_stride, n_clips, ctx->cos_table.as<double>(),
                                          d_out, d_dc, stream));
        return VDF_OK;
    }
    const int need_h = (w != VDF_DCT_SIZE), need_v = (h != VDF_DCT_SIZE);
    int rc = VDF_OK;
    const uint8_t *buf_end = d_frames + (n_clips - 1) * clip_stride + (VDF_DCT_SIZE - 1) * frame_stride + (size_t)w * h;
    if (ctx->resize_mode != 1) {
        // Resize on the matrix cores (exact i8 x i8 -> i32): small frames fuse the DCT into the same kernel.
        // frames taller than two 64-row groups go to the per-frame kernel; its whole-line form is the default
        // (round 5: except the wide, short ones that stream faster - resize_short_prefers_stream - where they are eligible to)
        // (and the narrow ones of up to 256 rows that the tiled persistent kernel serves better than the stream kernels - resize_tall_prefers_tiled)
        const bool fused = ctx->resize_mode == 3 ||
                           (ctx->resize_mode == 0 && (h + 63) / 64 <= 2 &&
                            !(vdf::resize_short_prefers_stream(w, h) &&
                              vdf::resize_stream_eligible(d_frames, w, h, frame_stride, clip_stride, ctx->wavestream_knob))) ||
                           (ctx->resize_mode == 0 && !ctx->hash_no_persistent && vdf::resize_tall_prefers_tiled(w, h));
        // tightly packed frames stream linearly through LDS where that is the faster form (resize_stream_eligible)
        bool streamed = !fused && (ctx->resize_mode == 0 || ctx->resize_mode == 5) &&
                        vdf::resize_stream_eligible(d_frames, w, h, frame_stride, clip_stride, ctx->wavestream_knob);
        DeviceMfmaTable *mh = nullptr;
        if (streamed && vdf::resize_stream_wants_band(w, ctx->wavestream_knob)) {  // wide frames: the horizontal table in band form
            mh = mfma_table(ctx, w, vdf::kMfmaLayoutHorizontalBand, stream, &rc);
            if (rc) return rc;
            if (!mh->host.ok) { streamed = false; mh = nullptr; }
        }
        // frames wider than the per-wave buffers (1920 columns): the K-split form, table in registers (measured against the
        // whole-line kernel: 3840 wide 5.6 -> 6.7 TB/s, 2560 5.5 -> 6.2, 2000 3.6 -> 5.8, 2048 level)
        const bool ksplit = !fused && ((ctx->resize_mode == 0 && !streamed && w > 1920) || ctx->resize_mode == 6) &&
                            vdf::resize_ksplit_eligible(d_frames, w, h, frame_stride, clip_stride);
        if (ksplit) { streamed = false; mh = nullptr; }
        const bool wide = !fused && !streamed && !ksplit;
        if (!mh) mh = mfma_table(ctx, w, vdf::kMfmaLayoutHorizontal, stream, &rc);
        if (rc) return rc;
        DeviceMfmaTable *mv = mfma_table(ctx, h, wide ? vdf::kMfmaLayoutVerticalWide : vdf::kMfmaLayoutVertical, stream, &rc);
        if (rc) return rc;
        if (mh->host.ok && mv->host.ok) {
            vdf::MfmaResizeArgs a{};
            a.bh = mh->operand.p;
            a.av = mv->operand.p;
            a.bias_h = mh->bias.as<int32_t>();
            a.bias_v = mv->bias.as<int32_t>();
            a.prec_h = mh->host.precision;
            a.prec_v = mv->host.precision;
            a.n_kt = mh->host.n_tiles;
            a.n_rg = mv->host.n_tiles;
            if (!mh->host.band_meta.empty()) {
                a.band_meta = mh->meta.as<int32_t>();
                a.band_stride = mh->host.band_stride;
            }
            a.no_persistent = ctx->hash_no_persistent;
            a.persistent_wgs_per_cu = ctx->hash_wgs_per_cu;
            a.wavestream_knob = ctx->wavestream_knob;
            if (fused) {
                VDF_HIP(ctx, vdf::launch_resize_dct_fused(d_frames, n_clips, w, h, frame_stride, clip_stride, buf_end, a,
                                                          ctx->cos_table.as<double>(), d_out, d_dc, stream));
                return VDF_OK;
            }
            VDF_HIP(ctx, ctx->small.reserve(n_clips * 4096));
            if (ksplit) {
                VDF_HIP(ctx, vdf::launch_resize_mfma_frames_ksplit(d_frames, n_clips, w, h, frame_stride, clip_stride, a,
                                                                   ctx->small.as<uint8_t>(), stream));
            } else if (streamed) {
                VDF_HIP(ctx, vdf::launch_resize_mfma_frames_stream(d_frames, n_clips, w, h, frame_stride, clip_stride, a,
                                                                   ctx->small.as<uint8_t>(), stream));
            } else {
                VDF_HIP(ctx, vdf::launch_resize_mfma_frames(d_frames, n_clips, w, h, frame_stride, clip_stride, buf_end, a,
                                                            ctx->small.as<uint8_t>(), wide, stream));
            }
            VDF_HIP(ctx, vdf::launch_dct_hash(ctx->small.as<uint8_t>(), 4096, 256, n_clips,
                                              ctx->cos_table.as<double>(), d_out, d_dc, stream));
            return VDF_OK;
        }
        if (ctx->resize_mode != 0) return fail(ctx, VDF_E_BAD_DIMS, "coefficients do not fit the i8 split");
    }
    // scalar fixed-point fallback (any coefficient range)
    DeviceAxisTable *th = need_h ? axis_table(ctx, w, stream, &rc) : nullptr;
    if (rc) return rc;
    DeviceAxisTable *tv = need_v ? axis_table(ctx, h, stream, &rc) : nullptr;
    if (rc) return rc;
    int32_t y_first = 0, tmp_rows = VDF_DCT_SIZE;
    if (need_v) {
        y_first = tv->host.start[0];
        tmp_rows = tv->host.start[VDF_DCT_SIZE - 1] + tv->host.size[VDF_DCT_SIZE - 1] - y_first;
    }
    if ((size_t)tmp_rows * 16 > 64 * 1024) return fail(ctx, VDF_E_BAD_DIMS, "frame height above 4096 is not supported by the scalar resize kernel");
    VDF_HIP(ctx, ctx->small.reserve(n_clips * 4096));
    VDF_HIP(ctx, vdf::launch_resize_generic(d_frames, n_clips, w, h, frame_stride, clip_stride, dev_view(th),
                                            dev_view(tv), need_h, need_v, y_first, tmp_rows,
                                            ctx->small.as<uint8_t>(), stream));
    VDF_HIP(ctx, vdf::launch_dct_hash(ctx->small.as<uint8_t>(), 4096, 256, n_clips, ctx->cos_table.as<double>(),
                                      d_out, d_dc, stream));
    return VDF_OK;
}

// Hash clips whose crop boxes (HOST array [n_clips][4] = left, right, top, bottom; null = no crop) are read in place.
int hash_cropped_locked(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w,
                        uint32_t h, size_t frame_stride, size_t clip_stride, const uint32_t *crops, uint64_t *d_out,
                        uint32_t *d_dc, hipStream_t stream)
{
    if (n_clips > kMaxClipsPerLaunch) {
        for (size_t c0 = 0; c0 < n_clips; c0 += kMaxClipsPerLaunch) {
            const size_t nb = std::min(kMaxClipsPerLaunch, n_clips - c0);
            int rc = hash_cropped_locked(ctx, d_frames + c0 * clip_stride, nb, frames_per_clip, w, h, frame_stride,
                                         clip_stride, crops ? crops + 4 * c0 : nullptr, d_out + c0 * VDF_HASH_WORDS,
                                         d_dc ? d_dc + c0 : nullptr, stream);
            if (rc) return rc;
        }
        return VDF_OK;
    }
    bool any = false;
    if (crops)
        for (size_t i = 0; i < n_clips * 4 && !any; i++) any = crops[i] != 0;
    if (!any)  // the common case: the fused / persistent kernels
        return hash_device_locked(ctx, d_frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride, d_out, d_dc, stream);
    if (frames_per_clip < VDF_DCT_SIZE) return fail(ctx, VDF_E_NOT_ENOUGH_FRAMES, "fewer than 16 frames per clip");
    if (w == 0 || h == 0) return fail(ctx, VDF_E_BAD_DIMS, "zero frame dimension");
    if (frame_stride < (size_t)w * h) return fail(ctx, VDF_E_INVAL, "frame_stride smaller than a frame");
    if (!d_frames || !d_out) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n_clips > 0x7FFFFFFull) return fail(ctx, VDF_E_INVAL, "too many clips in one call");
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    {
        int rc0 = ensure_cos_table(ctx, stream);
        if (rc0) return rc0;
    }
    // Three kernels read crop boxes in place; a call is dealt out between the first two by box shape, clip by clip (each entry of a
    // launch names its clip of the batch: CropStreamClip::src_clip), and one dct_hash launch follows over the whole batch:
    //  * ROWCROP stream kernels - full-width boxes (top / bottom bars only: a 2.39 : 1 film in a 16 : 9 frame, the commonest letterbox;
    //    clips without bars among them are boxes of the whole frame).  The box is a contiguous range of rows at the frame's own pitch,
    //    so it streams like a shorter frame: the kernel the uncropped call would take at that width (per-wave, chunk or K-split form)
    //    with a per-clip first row, height and vertical table.
    //  * the cropped stream kernel - boxes with side bars (pillarboxed clips): rows x0 .. x0 + w of the box go through LDS by gather
    //    DMA.  Also full-width boxes where no ROWCROP kernel applies and the pitch is not a multiple of the 128-byte line.
    //    Measured, detect + crop + hash against the whole-line kernel below: 854x480 1.68 -> 1.24 ms per 1000 clips, 720x576 1.42 -> 1.14,
    //    426x240 x4000 1.85 -> 1.33, 1366x768 x500 2.19 -> 1.95; pillarboxed 1920x1080 x1000 crop + hash 7.7 -> 5.5 ms, 1280x720 x2000
    //    6.4 -> 3.9, 1536x864 4.5 -> 2.9, 640x360 x4000 3.4 -> 2.5; 1024 wide 3.5 -> 3.8: stays (gpurun_out/r03pb).
    //  * the whole-line cropped kernel (below) - everything else: misaligned buffers, frames that do not end on 16 bytes, short frames,
    //    boxes whose tables do not fit the i8 split, VDF_RESIZE_MODE=4.
    const bool tall = (h + 63) / 64 > 2, ends16 = ((uint64_t)w * h) % 16 == 0 && (uint64_t)w * h < (1ull << 31);
    // small frames (round 5): one workgroup per CLIP, a wave per four frames, resize + DCT (resize_dct_hash_cropped_small_kernel) - plain vertical layout
    const bool small_crop = !tall && w <= 256 && ctx->resize_mode == 0 && !ctx->no_smallcrop;
    if (small_crop) {
        // Small frames: ONE pass over the boxes - check, table entries by box size (arrays, not a map: 20 000 clips were 0.19 ms of host time in
        // the general plan below, with the GPU idle behind the wait for the boxes), descriptors written straight into the pinned staging -
        // then one upload and one launch that resizes, transforms and hashes (resize_dct_hash_cropped_small_kernel).
        const size_t nd = n_clips * sizeof(vdf::CropClipDesc), off = (nd + 63) & ~size_t(63);
        if (!ctx->pin_desc.reserve(off + ((size_t)w + h + 2) * sizeof(vdf::CropTableEntry))) return fail(ctx, VDF_E_OOM, "host staging for the crop descriptors");
        vdf::CropClipDesc *dsc = ctx->pin_desc.as<vdf::CropClipDesc>();
        vdf::CropTableEntry *ent = reinterpret_cast<vdf::CropTableEntry *>(ctx->pin_desc.as<char>() + off);
        std::vector<int32_t> at_h(w + 1, -1), at_v(h + 1, -1);
        uint32_t n_ent = 0;
        int rc = VDF_OK;
        auto entry = [&](uint32_t size, bool vertical) -> int32_t {
            DeviceMfmaTable *t = mfma_table(ctx, size, vertical ? vdf::kMfmaLayoutVertical : vdf::kMfmaLayoutHorizontal, stream, &rc);
            if (rc) return -1;
            if (!t->host.ok) { rc = fail(ctx, VDF_E_BAD_DIMS, "crop box size whose coefficients do not fit the i8 split"); return -1; }
            ent[n_ent] = vdf::CropTableEntry{t->operand.p, t->bias.as<int32_t>(), t->host.n_tiles, t->host.precision};
            return (int32_t)n_ent++;
        };
        for (size_t c = 0; c < n_clips; c++) {
            const uint32_t l = crops[4 * c], r = crops[4 * c + 1], t = crops[4 * c + 2], b = crops[4 * c + 3];
            if ((uint64_t)l + r >= w || (uint64_t)t + b >= h) return fail(ctx, VDF_E_INVAL, "crop box leaves no pixels");  // crop.rs:21-22
            const uint32_t bw = w - l - r, bh = h - t - b;
            if (at_h[bw] < 0 && (at_h[bw] = entry(bw, false)) < 0) return rc;
            if (at_v[bh] < 0 && (at_v[bh] = entry(bh, true)) < 0) return rc;
            dsc[c] = vdf::CropClipDesc{l, t, bw, bh, (uint32_t)at_h[bw], (uint32_t)at_v[bh], (uint32_t)c, 0u};
        }
        if ((rc = upload(ctx, ctx->crop_desc2, dsc, nd, stream))) return rc;
        if ((rc = upload(ctx, ctx->crop_tables2, ent, std::max<size_t>(n_ent, 1) * sizeof(vdf::CropTableEntry), stream))) return rc;
        VDF_HIP(ctx, hipEventRecord(ctx->ev_mid, stream));
        const uint8_t *buf_end = d_frames + (n_clips - 1) * clip_stride + (VDF_DCT_SIZE - 1) * frame_stride + (size_t)w * h;
        VDF_HIP(ctx, vdf::launch_resize_dct_cropped_small(d_frames, n_clips, w, frame_stride, clip_stride, buf_end, ctx->crop_desc2.as<vdf::CropClipDesc>(),
                                                          ctx->crop_tables2.as<vdf::CropTableEntry>(), ctx->cos_table.as<double>(), d_out, d_dc, stream));
        // the staging is read by the two copies: they must have run before the next call on this context rewrites it (the kernel stays queued)
        VDF_HIP(ctx, hipEventSynchronize(ctx->ev_mid));
        return VDF_OK;
    }
    std::vector<uint32_t> rows_clips, side_clips;  // by box shape
    for (size_t c = 0; c < n_clips; c++) {
        const uint32_t l = crops[4 * c], r = crops[4 * c + 1], t = crops[4 * c + 2], b = crops[4 * c + 3];
        if ((uint64_t)l + r >= w || (uint64_t)t + b >= h) return fail(ctx, VDF_E_INVAL, "crop box leaves no pixels");  // crop.rs:21-22
        (l == 0 && r == 0 ? rows_clips : side_clips).push_back((uint32_t)c);
    }
    // -- can the full-width boxes take a ROWCROP kernel, and which?
    bool row_stream = false, row_ksplit = false, row_band = false;
    DeviceMfmaTable *row_mh = nullptr;
    if (ctx->resize_mode == 0 && tall && !rows_clips.empty() && (vdf::resize_rowcrop_streams(w) || ctx->rowcrop_all) && !ctx->no_rowcrop) {
        row_stream = vdf::resize_stream_eligible(d_frames, w, h, frame_stride, clip_stride, ctx->wavestream_knob);
        row_band = row_stream && vdf::resize_stream_wants_band(w, ctx->wavestream_knob);
        row_ksplit = !row_stream && w > 1920 && vdf::resize_ksplit_eligible(d_frames, w, h, frame_stride, clip_stride);
        if (row_stream || row_ksplit) {
            int rc = VDF_OK;
            row_mh = mfma_table(ctx, w, row_band ? vdf::kMfmaLayoutHorizontalBand : vdf::kMfmaLayoutHorizontal, stream, &rc);
            if (rc) return rc;
            if (!row_mh->host.ok) row_mh = nullptr;
        }
    }
    // -- can boxes take the cropped stream kernel?
    int stream_cls = 0;
    const bool crop_stream_ok = tall && ends16 && (((uintptr_t)d_frames | frame_stride | clip_stride) & 3) == 0 &&
                                (ctx->resize_mode == 0 || ctx->resize_mode == 5) && vdf::resize_cropped_stream_class(w, &stream_cls);
    // descriptors of a ROWCROP launch over `ids`; false: some box's vertical table does not fit the i8 split
    std::vector<vdf::CropStreamClip> rsc, ssc;
    std::vector<vdf::CropStreamTable> rst, sst;
    std::map<uint32_t, uint32_t> vindex;  // box height -> entry of rst
    auto append_rows = [&](const std::vector<uint32_t> &ids, int *rc) -> bool {  // (appends: a call may hold several ROWCROP launches)
        const size_t first = rsc.size();
        rsc.resize(first + ids.size(), vdf::CropStreamClip{});
        for (size_t i = 0; i < ids.size(); i++) {
            const uint32_t c = ids[i], t = crops[4 * c + 2], b = crops[4 * c + 3];
            vdf::CropStreamClip &q = rsc[first + i];
            q.x0 = crops[4 * c]; q.y0 = t; q.w = w - crops[4 * c] - crops[4 * c + 1]; q.h = h - t - b; q.wp = w; q.src_clip = c;
            auto it = vindex.find(q.h);
            if (it == vindex.end()) {
                DeviceMfmaTable *tv = mfma_table(ctx, q.h, vdf::kMfmaLayoutVertical, stream, rc);
                if (*rc || !tv->host.ok) return false;
                rst.push_back(vdf::CropStreamTable{tv->operand.p, tv->bias.as<int32_t>(), nullptr, tv->host.n_tiles, tv->host.precision, 0, 0});
                it = vindex.emplace(q.h, (uint32_t)rst.size() - 1).first;
            }
            q.v_table = it->second;
        }
        return true;
    };
    // descriptors of a cropped-stream launch over `ids`; false: a box does not fit the kernel (tables, blocks per chunk)
    bool need_shift = (w & 3u) != 0;
    auto build_stream = [&](const std::vector<uint32_t> &ids, int *rc) -> bool {
        std::map<uint64_t, uint32_t> sindex;  // (size * 2 + vertical) -> entry
        ssc.assign(ids.size(), vdf::CropStreamClip{});
        sst.clear();
        auto stream_entry = [&](uint32_t size, bool vertical, uint32_t *at) -> bool {
            const uint64_t key = (uint64_t)size * 2 + (vertical ? 1 : 0);
            auto it = sindex.find(key);
            if (it != sindex.end()) { *at = it->second; return true; }
            DeviceMfmaTable *t = mfma_table(ctx, size, vertical ? vdf::kMfmaLayoutVertical : vdf::kMfmaLayoutHorizontalBand, stream, rc);
            if (*rc || !t->host.ok) return false;
            sst.push_back(vdf::CropStreamTable{t->operand.p, t->bias.as<int32_t>(), vertical ? nullptr : t->meta.as<int32_t>(), t->host.n_tiles,
                                               t->host.precision, t->host.band_stride, 0});
            *at = sindex[key] = (uint32_t)sst.size() - 1;
            return true;
        };
        for (size_t i = 0; i < ids.size(); i++) {
            const uint32_t c = ids[i], l = crops[4 * c], r = crops[4 * c + 1], t = crops[4 * c + 2], b = crops[4 * c + 3];
            vdf::CropStreamClip &q = ssc[i];
            q.x0 = l; q.y0 = t; q.w = w - l - r; q.h = h - t - b; q.src_clip = c;
            q.nb = vdf::resize_cropped_stream_blocks(q.w, q.x0, w, stream_cls, &q.wp);
            need_shift = need_shift || (q.x0 & 3u) != 0;
            if (q.nb == 0 || (q.nb < 2 && q.h > 16)) return false;  // one block per chunk would leave three of the four waves idle
            q.step_rows = 4096u / q.wp;
            q.step_x = 4096u - q.step_rows * q.wp;
            q.n_chunks = (q.h + 16 * q.nb - 1) / (16 * q.nb);
            if (!stream_entry(q.w, false, &q.h_table) || !stream_entry(q.h, true, &q.v_table)) return false;
        }
        return true;
    };
    auto upload_desc = [&](const std::vector<vdf::CropStreamClip> &sc, const std::vector<vdf::CropStreamTable> &st, DevBuf &bd, DevBuf &bt) -> int {
        int rc = upload(ctx, bd, sc.data(), sc.size() * sizeof(vdf::CropStreamClip), stream);
        if (rc == VDF_OK) rc = upload(ctx, bt, st.data(), st.size() * sizeof(vdf::CropStreamTable), stream);
        return rc;
    };
    auto launch_rows = [&](size_t first, size_t n_sub, DevBuf &bd, DevBuf &bt) -> int {
        vdf::MfmaResizeArgs a{};
        a.bh = row_mh->operand.p;
        a.bias_h = row_mh->bias.as<int32_t>();
        a.prec_h = row_mh->host.precision;
        a.n_kt = row_mh->host.n_tiles;
        a.wavestream_knob = ctx->wavestream_knob;
        if (row_band) {
            a.band_meta = row_mh->meta.as<int32_t>();
            a.band_stride = row_mh->host.band_stride;
        }
        if (row_ksplit)
            VDF_HIP(ctx, vdf::launch_resize_mfma_frames_ksplit(d_frames, n_sub, w, h, frame_stride, clip_stride, a, ctx->small.as<uint8_t>(), stream,
                                                               bd.as<vdf::CropStreamClip>() + first, bt.as<vdf::CropStreamTable>()));
        else
            VDF_HIP(ctx, vdf::launch_resize_mfma_frames_stream(d_frames, n_sub, w, h, frame_stride, clip_stride, a, ctx->small.as<uint8_t>(), stream,
                                                               bd.as<vdf::CropStreamClip>() + first, bt.as<vdf::CropStreamTable>()));
        return VDF_OK;
    };
    // boxes with side bars that share their column range: the per-wave kernel gathers the box (one launch per distinct range)
    struct BoxGroup { uint32_t x0, bw; std::vector<uint32_t> ids; DeviceMfmaTable *mh; size_t first; };
    auto launch_box = [&](const BoxGroup &g, DevBuf &bd, DevBuf &bt) -> int {
        vdf::MfmaResizeArgs a{};
        a.bh = g.mh->operand.p;
        a.bias_h = g.mh->bias.as<int32_t>();
        a.prec_h = g.mh->host.precision;
        a.n_kt = g.mh->host.n_tiles;
        a.band_meta = g.mh->meta.as<int32_t>();
        a.band_stride = g.mh->host.band_stride;
        a.wavestream_knob = ctx->wavestream_knob;
        VDF_HIP(ctx, vdf::launch_resize_mfma_box_wavestream(d_frames, g.ids.size(), w, h, frame_stride, clip_stride, a, g.x0, g.bw,
                                                            bd.as<vdf::CropStreamClip>() + g.first, bt.as<vdf::CropStreamTable>(),
                                                            ctx->small.as<uint8_t>(), stream));
        return VDF_OK;
    };
    auto launch_stream = [&](size_t n_sub, DevBuf &bd, DevBuf &bt) -> int {
        VDF_HIP(ctx, vdf::launch_resize_mfma_cropped_stream(d_frames, n_sub, w, h, frame_stride, clip_stride, bd.as<vdf::CropStreamClip>(),
                                                            bt.as<vdf::CropStreamTable>(), stream_cls, need_shift, ctx->small.as<uint8_t>(), stream));
        return VDF_OK;
    };
    auto uploads_done = [&]() -> int {
        VDF_HIP(ctx, hipStreamSynchronize(stream));  // the host vectors go out of scope with this call
        VDF_HIP(ctx, ctx->small.reserve(n_clips * 4096));
        return VDF_OK;
    };
    auto finish = [&]() -> int {
        VDF_HIP(ctx, vdf::launch_dct_hash(ctx->small.as<uint8_t>(), 4096, 256, n_clips, ctx->cos_table.as<double>(), d_out, d_dc, stream));
        return VDF_OK;
    };
    // -- the whole-line cropped kernel, over `ids` (all clips, or the side-bar boxes of a mixed call whose full-width boxes stream)
    std::vector<vdf::CropClipDesc> desc;
    std::vector<vdf::CropTableEntry> entries;
    const bool wide = w >= 192;  // frames at least 1.5 windows wide read whole 128-byte lines (resize_row_quads)
    auto build_lines = [&](const std::vector<uint32_t> &ids, int *rc) -> bool {
        std::map<uint64_t, uint32_t> index;  // (size * 2 + vertical) -> entry
        desc.assign(ids.size(), vdf::CropClipDesc{});
        entries.clear();
        auto entry_for = [&](uint32_t size, bool vertical, uint32_t *at) -> bool {
            const uint64_t key = (uint64_t)size * 2 + (vertical ? 1 : 0);
            auto it = index.find(key);
            if (it != index.end()) { *at = it->second; return true; }
            DeviceMfmaTable *t = mfma_table(ctx, size, !vertical ? vdf::kMfmaLayoutHorizontal : wide ? vdf::kMfmaLayoutVerticalWide : vdf::kMfmaLayoutVertical, stream, rc);
            if (*rc) return false;
            if (!t->host.ok) { *rc = fail(ctx, VDF_E_BAD_DIMS, "crop box size whose coefficients do not fit the i8 split"); return false; }
            entries.push_back(vdf::CropTableEntry{t->operand.p, t->bias.as<int32_t>(), t->host.n_tiles, t->host.precision});
            *at = index[key] = (uint32_t)entries.size() - 1;
            return true;
        };
        for (size_t i = 0; i < ids.size(); i++) {
            const uint32_t c = ids[i], l = crops[4 * c], r = crops[4 * c + 1], t = crops[4 * c + 2], b = crops[4 * c + 3];
            vdf::CropClipDesc &q = desc[i];
            q.x0 = l; q.y0 = t; q.w = w - l - r; q.h = h - t - b; q.src_clip = c;
            if (!entry_for(q.w, false, &q.h_table) || !entry_for(q.h, true, &q.v_table)) return false;
        }
        return true;
    };
    const uint8_t *buf_end = d_frames + (n_clips - 1) * clip_stride + (VDF_DCT_SIZE - 1) * frame_stride + (size_t)w * h;
    auto launch_lines = [&](size_t n_sub, DevBuf &bd, DevBuf &bt) -> int {
        VDF_HIP(ctx, vdf::launch_resize_mfma_cropped(d_frames, n_sub, w, frame_stride, clip_stride, buf_end, bd.as<vdf::CropClipDesc>(),
                                                     bt.as<vdf::CropTableEntry>(), ctx->small.as<uint8_t>(), wide, stream));
        return VDF_OK;
    };
    auto upload_lines = [&](DevBuf &bd, DevBuf &bt) -> int {
        // through pinned staging (consumed by the time uploads_done() returns): a descriptor per clip is 0.6 MB for 20 000 small clips
        const size_t nd = desc.size() * sizeof(vdf::CropClipDesc), nt = entries.size() * sizeof(vdf::CropTableEntry), off = (nd + 63) & ~size_t(63);
        if (!ctx->pin_desc.reserve(off + nt)) return fail(ctx, VDF_E_OOM, "host staging for the crop descriptors");
        std::memcpy(ctx->pin_desc.p, desc.data(), nd);
        std::memcpy(ctx->pin_desc.as<char>() + off, entries.data(), nt);
        int rc = upload(ctx, bd, ctx->pin_desc.p, nd, stream);
        if (rc == VDF_OK) rc = upload(ctx, bt, ctx->pin_desc.as<char>() + off, nt, stream);
        return rc;
    };
    // ---- the plan: who goes where
    int rc = VDF_OK;
    std::vector<uint32_t> rows_part, rest;
    (row_mh ? rows_part : rest) = rows_clips;
    std::vector<BoxGroup> boxes;
    if (ctx->resize_mode == 0 && tall && ends16 && (((uintptr_t)d_frames | frame_stride | clip_stride) & 15) == 0 && !side_clips.empty() &&
        !ctx->no_boxstream && !ctx->no_rowcrop) {
        std::map<uint64_t, size_t> by_range;  // (x0, width) -> group
        std::vector<BoxGroup> cand;
        for (uint32_t c : side_clips) {
            const uint32_t l = crops[4 * c], bw = w - l - crops[4 * c + 1];
            auto it = by_range.find(((uint64_t)l << 32) | bw);
            if (it == by_range.end()) {
                it = by_range.emplace(((uint64_t)l << 32) | bw, cand.size()).first;
                cand.push_back(BoxGroup{l, bw, {}, nullptr, 0});
            }
            cand[it->second].ids.push_back(c);
        }
        for (BoxGroup &g : cand) {
            const int nw = vdf::resize_wavestream_waves_box(w, g.x0, g.bw, ctx->wavestream_knob);
            // (a launch per range: ranges shared by fewer than four clips - a launch would leave most CUs idle - and the ranges beyond sixteen
            // are left to the gather kernel)
            if (nw && g.ids.size() >= 4 && boxes.size() < 16) {
                g.mh = mfma_table(ctx, g.bw, vdf::kMfmaLayoutHorizontalBand, stream, &rc);
                if (rc) return rc;
                if (g.mh->host.ok && vdf::resize_wavestream_table_fits(nw, g.mh->host.band_stride)) { boxes.push_back(std::move(g)); continue; }
            }
            rest.insert(rest.end(), g.ids.begin(), g.ids.end());
        }
    } else {
        rest.insert(rest.end(), side_clips.begin(), side_clips.end());
    }
    bool planned = !rows_part.empty() || !boxes.empty();
    if (planned) {
        planned = append_rows(rows_part, &rc);
        for (BoxGroup &g : boxes) {
            g.first = rsc.size();
            planned = planned && append_rows(g.ids, &rc);
        }
        if (rc) return rc;
    }
    if (!planned) {  // nothing streams per clip: the whole call through one general kernel
        rest.resize(n_clips);
        for (size_t c = 0; c < n_clips; c++) rest[c] = (uint32_t)c;
        rows_part.clear();
        boxes.clear();
    }
    // the general kernel for the rest: the cropped stream kernel for side-bar boxes at every pitch but 1024 and for full-width boxes where the
    // pitch is not line-aligned (measured above), else the whole-line kernel
    bool rest_stream = false;
    if (!rest.empty()) {
        bool rest_has_side = false;
        for (uint32_t c : rest) rest_has_side = rest_has_side || crops[4 * c] != 0 || crops[4 * c + 1] != 0;
        if (crop_stream_ok && (ctx->resize_mode == 5 || (rest_has_side ? w != 1024 : w % 128 != 0))) rest_stream = build_stream(rest, &rc);
        if (rc) return rc;
        if (!rest_stream && !build_lines(rest, &rc)) return rc ? rc : fail(ctx, VDF_E_BAD_DIMS, "crop box");
    }
    if (planned && (rc = upload_desc(rsc, rst, ctx->crop_desc, ctx->crop_tables))) return rc;
    if (!rest.empty() && (rc = rest_stream ? upload_desc(ssc, sst, ctx->crop_desc2, ctx->crop_tables2) : upload_lines(ctx->crop_desc2, ctx->crop_tables2))) return rc;
    if ((rc = uploads_done())) return rc;
    if (!rows_part.empty() && (rc = launch_rows(0, rows_part.size(), ctx->crop_desc, ctx->crop_tables))) return rc;
    for (const BoxGroup &g : boxes)
        if ((rc = launch_box(g, ctx->crop_desc, ctx->crop_tables))) return rc;
    if (!rest.empty() && (rc = rest_stream ? launch_stream(rest.size(), ctx->crop_desc2, ctx->crop_tables2) : launch_lines(rest.size(), ctx->crop_desc2, ctx->crop_tables2)))
        return rc;
    return finish();
}

int letterbox_hash_device_locked(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                 uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *d_out,
                                 uint32_t *d_dc, uint32_t *out_crops, hipStream_t stream, uint32_t *d_out_crops)
{
    if (frames_per_clip < VDF_DCT_SIZE) return fail(ctx, VDF_E_NOT_ENOUGH_FRAMES, "fewer than 16 frames per clip");
    if (w == 0 || h == 0) return fail(ctx, VDF_E_BAD_DIMS, "zero frame dimension");
    if (frame_stride < (size_t)w * h) return fail(ctx, VDF_E_INVAL, "frame_stride smaller than a frame");
    if (n_clips == 0) return VDF_OK;
    if (!d_frames || !d_out) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n_clips > kMaxClipsPerLaunch) {
        for (size_t c0 = 0; c0 < n_clips; c0 += kMaxClipsPerLaunch) {
            const size_t nb = std::min(kMaxClipsPerLaunch, n_clips - c0);
            int rc = letterbox_hash_device_locked(ctx, d_frames + c0 * clip_stride, nb, frames_per_clip, w, h,
                                                  frame_stride, clip_stride, d_out + c0 * VDF_HASH_WORDS,
                                                  d_dc ? d_dc + c0 : nullptr, out_crops ? out_crops + 4 * c0 : nullptr,
                                                  stream, d_out_crops ? d_out_crops + 4 * c0 : nullptr);
            if (rc) return rc;
        }
        return VDF_OK;
    }
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    // Small frames (round 6): the boxes never visit the host.  Every box size's tables are resident (box_table_set), so
    //  * frames of at most 64 x 64 take ONE kernel that detects, crops, resizes, transforms and hashes (letterbox_resize_dct_hash_small_kernel);
    //    the clips within 64 bytes of the buffer's end (the last one, as a rule) take the route below for its careful loader;
    //  * other small frames: the two detect kernels, then the one-workgroup-per-clip kernel reads each clip's box where they left it.
    // No copy to the host, no wait, no host loop over the clips between the launches; out_crops is filled by one copy queued behind them
    // and waited for at the END of the call (d_out_crops: no wait at all).
    const bool small_frames = (h + 63) / 64 <= 2 && w <= 256 && ctx->resize_mode == 0 && !ctx->no_smallcrop && !ctx->lb_host_plan &&
                              n_clips <= 0x7FFFFFFull;
    if (small_frames) {
        int rc = ensure_cos_table(ctx, stream);
        if (rc) return rc;
        BoxTableSet *set = box_table_set(ctx, w, h, stream, &rc);
        if (rc) return rc;
        if (set->usable) {
            uint32_t *d_crops = d_out_crops;
            if (!d_crops) {
                VDF_HIP(ctx, ctx->crops.reserve(n_clips * 16));
                d_crops = ctx->crops.as<uint32_t>();
            }
            const uint8_t *buf_end = d_frames + (n_clips - 1) * clip_stride + (VDF_DCT_SIZE - 1) * frame_stride + (size_t)w * h;
            size_t n_fused = 0;
            if (set->one_tile && !ctx->no_lb_fused) {
                // clip c's loads stay inside the buffer iff it ends at least 64 bytes before the last clip does: (n - 1 - c) * clip_stride >= 64
                const size_t n_tail = clip_stride == 0 ? n_clips : std::min<size_t>(n_clips, (64 + clip_stride - 1) / clip_stride);
                n_fused = n_clips - n_tail;
                VDF_HIP(ctx, vdf::launch_letterbox_hash_small(d_frames, n_fused, w, h, frame_stride, clip_stride, set->blob.p, ctx->cos_table.as<double>(),
                                                              d_out, d_dc, d_crops, ctx->hash_wgs_per_cu_set ? ctx->hash_wgs_per_cu : 0, stream));
            }
            if (n_fused < n_clips) {
                const size_t nr = n_clips - n_fused;
                const uint8_t *fr = d_frames + n_fused * clip_stride;
                VDF_HIP(ctx, ctx->crop_work.reserve(vdf::letterbox_work_bytes(nr, frames_per_clip)));
                VDF_HIP(ctx, vdf::launch_letterbox(fr, nr, frames_per_clip, w, h, frame_stride, clip_stride, d_crops + 4 * n_fused,
                                                   ctx->crop_work.as<uint32_t>(), stream, ctx->lb_side_strips));
                VDF_HIP(ctx, vdf::launch_resize_dct_cropped_small_boxes(fr, nr, w, h, frame_stride, clip_stride, buf_end, d_crops + 4 * n_fused,
                                                                        set->entries.as<vdf::CropTableEntry>(), ctx->cos_table.as<double>(),
                                                                        d_out + n_fused * VDF_HASH_WORDS, d_dc ? d_dc + n_fused : nullptr, stream));
            }
            if (out_crops) {
                if (!ctx->pin_crops.reserve(n_clips * 16)) return fail(ctx, VDF_E_OOM, "host staging for the crop boxes");
                VDF_HIP(ctx, hipMemcpyAsync(ctx->pin_crops.p, d_crops, n_clips * 16, hipMemcpyDeviceToHost, stream));
                VDF_HIP(ctx, hipEventRecord(ctx->ev_mid, stream));
                VDF_HIP(ctx, hipEventSynchronize(ctx->ev_mid));  // everything of this call is queued: the wait costs the GPU nothing
                std::memcpy(out_crops, ctx->pin_crops.p, n_clips * 16);
            }
            return VDF_OK;
        }
    }
    // (Cutting a large batch into chunks whose detect passes run on a second stream under the resize of the chunks before them was
    // built and measured in round 5 - profiles/r05_letterbox_ab.txt: the resize kernels fill every CU's LDS, so the walkers only got in
    // between the chunks, and three chunk boundaries cost more than the hidden detect time saved: 1000 pillarboxed 1080p clips 5.45 ms
    // against 5.32.  The detect pass was made faster instead: csrc/cropdetect.hip.)
    // Larger frames: which kernel a box takes (row-range stream, column-range stream, gather, whole-line) and the tables of its size are
    // decided per box SHAPE on the host - tables for every box size of a 1080p frame would be 50 MB and a second of host time per frame size -
    // so the boxes come down once and the call waits for the detect.  (The detect writes straight into the caller's d_out_crops when given.)
    uint32_t *d_crops = d_out_crops;
    if (!d_crops) {
        VDF_HIP(ctx, ctx->crops.reserve(n_clips * 16));
        d_crops = ctx->crops.as<uint32_t>();
    }
    VDF_HIP(ctx, ctx->crop_work.reserve(vdf::letterbox_work_bytes(n_clips, frames_per_clip)));
    VDF_HIP(ctx, vdf::launch_letterbox(d_frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride, d_crops, ctx->crop_work.as<uint32_t>(),
                                       stream, ctx->lb_side_strips));
    // (pinned: a pageable destination makes the copy synchronous and slow - 0.3 MB for 20 000 clips)
    if (!ctx->pin_crops.reserve(n_clips * 16)) return fail(ctx, VDF_E_OOM, "host staging for the crop boxes");
    uint32_t *crops = ctx->pin_crops.as<uint32_t>();
    VDF_HIP(ctx, hipMemcpyAsync(crops, d_crops, n_clips * 16, hipMemcpyDeviceToHost, stream));
    VDF_HIP(ctx, hipEventRecord(ctx->ev_wait, stream));
    if (int rcw = wait_event(ctx, ctx->ev_wait)) return rcw;  // (behind the frames' upload when they come from the host: milliseconds)
    if (out_crops) std::memcpy(out_crops, crops, n_clips * 16);
    return hash_cropped_locked(ctx, d_frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride, crops, d_out, d_dc, stream);
}


int search_refs_device_locked(vdf_ctx *ctx, const uint64_t *d_cand_hashes, const uint32_t *d_cand_durations,
                                     size_t n_cand, const uint64_t *d_ref_hashes, const uint32_t *d_ref_durations,
                                     size_t n_ref, uint32_t tol_int, uint32_t ref_index_base, vdf_hit *hits,
                                     uint64_t capacity, uint64_t *n_hits, hipStream_t s, vdf_ctx::HostHits *staging)
{
    ctx->stats = vdf_search_stats{};
    ctx->timing = vdf_search_timing{};
    *n_hits = 0;
    if (n_ref == 0 || n_cand == 0) return VDF_OK;
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    // References arrive in the caller's order; tiles want neighbouring rows to share a duration window, so
    // rows are visited through a stable duration-sorted permutation (reported indices stay the caller's).
    // (stable, on the device: the durations never leave HBM and nothing here waits for the GPU)
    const size_t sbytes = vdf::sort_order_scratch_bytes((uint32_t)n_ref, false);
    VDF_HIP(ctx, ctx->sort_scratch.reserve(sbytes));
    VDF_HIP(ctx, ctx->perm.reserve(std::max<size_t>(n_ref * 4, 16)));
    VDF_HIP(ctx, vdf::launch_sort_order(d_ref_durations, nullptr, (uint32_t)n_ref, ctx->perm.as<uint32_t>(), ctx->sort_scratch.p,
                                        ctx->sort_scratch.cap, s));
    int rc = VDF_OK;
    // Every hit is part of the output here (consume = false).  The hit buffer is the caller's to size (VDF_E_OVERFLOW with the
    // required size); the suspect queue of the matrix-core backend is the library's: if a launch dropped suspects, run it
    // again with a larger queue.
    for (int attempt = 0;; attempt++) {
        uint32_t overflow_row = 0;
        ctx->stats = vdf_search_stats{};
        rc = search_core(ctx, 1, d_cand_hashes, d_cand_durations, n_cand, d_ref_hashes, d_ref_durations,
                         ctx->perm.as<uint32_t>(), n_ref, tol_int, 0, 1, 0, 0xFFFFFFFFu, nullptr, ref_index_base, hits,
                         capacity, n_hits, &overflow_row, s, false, staging);
        if (rc) { ctx->cand_scale = 1; return rc; }
        if (*n_hits > capacity) { ctx->cand_scale = 1; return fail(ctx, VDF_E_OVERFLOW, "hit buffer too small; *n_hits holds the required size"); }
        if (overflow_row == 0xFFFFFFFFu) break;  // complete
        if (attempt >= 12) { ctx->cand_scale = 1; return fail(ctx, VDF_E_OVERFLOW, "suspect queue overflow"); }
        ctx->cand_scale *= 4;
    }
    ctx->cand_scale = 1;
    return VDF_OK;
}

int create_single(int device_id, vdf_ctx **out, std::string *err)
{
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        *err = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0");
        (void)hipGetLastError();
        return VDF_E_HIP;
    }
    if (device_id < 0 || device_id >= count) { *err = "device id out of range"; return VDF_E_INVAL; }
    vdf_ctx *ctx = new (std::nothrow) vdf_ctx();
    if (!ctx) return VDF_E_OOM;
    ctx->device = device_id;
    ctx->spin_wait = std::getenv("VDF_SPIN_WAIT") != nullptr;
    ctx->no_link_turns = std::getenv("VDF_NO_LINK_TURNS") != nullptr;
    const unsigned wait_flags = hipEventDisableTiming | (ctx->spin_wait ? 0u : (unsigned)hipEventBlockingSync);
    bool ok = hipSetDevice(device_id) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&ctx->ev0) == hipSuccess && hipEventCreate(&ctx->ev1) == hipSuccess &&
              hipEventCreate(&ctx->ev_mid) == hipSuccess && hipEventCreateWithFlags(&ctx->ev_wait, wait_flags) == hipSuccess;
    for (int i = 0; ok && i < 2; i++)
        ok = hipEventCreateWithFlags(&ctx->ev_copy[i], wait_flags) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->ev_done[i], wait_flags) == hipSuccess;
    if (!ok) {
        *err = std::string("context setup failed: ") + hipGetErrorString(hipGetLastError());
        delete ctx;
        return VDF_E_HIP;
    }
    if (const char *s = std::getenv("VDF_ROWS_PER_LANE")) {
        int r = std::atoi(s);
        if (r == 1 || r == 2 || r == 4) ctx->tile_rows = 256u * (uint32_t)r;
    }
    if (const char *s = std::getenv("VDF_RESIZE_MODE")) {
        int m = std::atoi(s);
        if (m >= 0 && m <= 6 && m != 2) ctx->resize_mode = m;  // 2 was the per-frame 16 x 64 B kernel, removed in round 3
    }
    if (const char *s = std::getenv("VDF_SEARCH_BACKEND")) {
        if (!std::strcmp(s, "valu")) ctx->search_backend = 0;
        else if (!std::strcmp(s, "mfma")) ctx->search_backend = 1;
    }
    if (const char *s = std::getenv("VDF_MFMA_CHUNK_COLS")) {
        long c = std::atol(s);
        if (c >= 32 && c <= (1 << 22)) ctx->mfma_chunk_cols = (uint32_t)c;
    }
    if (const char *s = std::getenv("VDF_CAND_CAPACITY")) { const long v = std::atol(s); if (v >= 8 && v <= 0x40000000l) ctx->cand_capacity_override = (uint32_t)v; }
    if (const char *s = std::getenv("VDF_MFMA_SELF_ROWS")) { const int v = std::atoi(s); if (v == 256 || v == 512) ctx->mfma_self_rows = (uint32_t)v; }
    if (const char *s = std::getenv("VDF_MFMA_REFS_ROWS")) { const int v = std::atoi(s); if (v == 256 || v == 512) ctx->mfma_refs_rows = (uint32_t)v; }
    if (const char *s = std::getenv("VDF_MFMA_PRUNE_STEP")) ctx->mfma_prune_step = std::atoi(s);
    if (const char *s = std::getenv("VDF_MFMA_MIN_WGS")) { const long v = std::atol(s); if (v >= 1 && v <= (1 << 24)) ctx->mfma_min_wgs = (uint32_t)v; }
    if (const char *s = std::getenv("VDF_MFMA_GROUP")) {
        long c = std::atol(s);
        if (c >= 1 && c <= (1 << 20)) ctx->mfma_group = (uint32_t)c;
    }
    if (const char *s = std::getenv("VDF_NO_HIT_FILTER")) ctx->no_hit_filter = std::atoi(s) != 0;
    if (std::getenv("VDF_NO_WAVESTREAM")) ctx->wavestream_knob = -1;
    else if (const char *s = std::getenv("VDF_WAVESTREAM_NW")) { const int v = std::atoi(s); if (v >= 3 && v <= 8) ctx->wavestream_knob = v; }
    ctx->no_rowcrop = std::getenv("VDF_NO_ROWCROP") != nullptr;
    ctx->rowcrop_all = std::getenv("VDF_ROWCROP_ALL") != nullptr;
    ctx->no_boxstream = std::getenv("VDF_NO_BOXSTREAM") != nullptr;
    ctx->no_smallcrop = std::getenv("VDF_NO_SMALLCROP") != nullptr;
    ctx->no_device_path_order = std::getenv("VDF_NO_DEVICE_PATH_ORDER") != nullptr;
    ctx->lb_host_plan = std::getenv("VDF_LB_HOST_PLAN") != nullptr;
    ctx->no_lb_fused = std::getenv("VDF_NO_LB_FUSED") != nullptr;
    if (std::getenv("VDF_LB_NC16")) ctx->lb_side_strips = 16;
    if (const char *s = std::getenv("VDF_COPY_THREADS")) { const int v = std::atoi(s); if (v >= 1 && v <= 64) ctx->copy_threads = v; }
    if (const char *s = std::getenv("VDF_HOST_CHUNK_MB")) { const long v = std::atol(s); if (v >= 1 && v <= 1024) ctx->host_chunk_bytes = (size_t)v << 20; }
    if (const char *s = std::getenv("VDF_HOST_DIRECT")) ctx->host_direct = std::atoi(s) != 0;
    if (const char *s = std::getenv("VDF_HASH_NO_PERSISTENT")) ctx->hash_no_persistent = std::atoi(s) != 0;
    if (const char *s = std::getenv("VDF_HASH_WGS_PER_CU")) {
        int v = std::atoi(s);
        if (v >= 1 && v <= 8) { ctx->hash_wgs_per_cu = v; ctx->hash_wgs_per_cu_set = true; }
    }
    if (const char *s = std::getenv("VDF_CHUNK_COLS")) {
        long c = std::atol(s);
        if (c >= 64 && c <= (1 << 20)) ctx->chunk_cols = (uint32_t)c;
    }
    *out = ctx;
    return VDF_OK;
}

// search() over a sorted database that is already resident on EVERY device of the context (each device's
// up_hashes / up_dur): row tiles are dealt round-robin over the devices (tile t -> device t % G), every device emits
// the thresholded pairs of its tiles, the host merges them and replays search_self's consumption ONCE
// (search_algorithm.rs:131-170) -> the same MatchGroups for every G.  Hit-buffer overflow: rows below the smallest
// row that lost a hit on any device are complete and are replayed; the consumption bitmap goes back to every device and
// the search resumes from that row.
int search_self_resident(vdf_ctx *ctx, size_t n, uint32_t tol_int, vdf_groups *out)
{
    const int G = device_count(ctx);
    uint64_t capacity = ctx->hit_capacity;
    std::vector<uint8_t> matched(n, 0);
    std::vector<uint32_t> bitmap;  // 1 bit per entry, built on the first overflow, then updated with the new members only
    bool use_bitmap = false;
    std::vector<vdf_hit> merged;
    uint32_t row_begin = 0;
    uint64_t span = n;  // rows per launch; shrinks after an overflow, grows back afterwards
    ctx->stats = vdf_search_stats{};
    const double t_call = now_ms();
    double replay_ms = 0.0;
    for (int k = 0; k < G; k++) { device_ctx(ctx, k)->stats = vdf_search_stats{}; device_ctx(ctx, k)->timing = vdf_search_timing{}; }
    while (row_begin < n) {
        const uint32_t row_end = (uint32_t)std::min<uint64_t>((uint64_t)row_begin + span, n);
        // the hits feed the replay and nothing else: rows that cannot become targets are dropped on the devices, which meet for
        // that through the context's own exchange when there are several (multi.cpp: LocalExchange)
        std::unique_ptr<ShardExchange> fx(make_local_exchange(ctx));
        int rc = for_each_device(ctx, [&](int k, vdf_ctx *d) {
            const int r = search_core(d, 0, d->up_hashes.as<uint64_t>(), d->up_dur.as<uint32_t>(), n, d->up_hashes.as<uint64_t>(),
                                      d->up_dur.as<uint32_t>(), nullptr, n, tol_int, (uint32_t)k, (uint32_t)G, row_begin, row_end,
                                      use_bitmap ? d->matched.as<uint32_t>() : nullptr, 0, nullptr, capacity,
                                      &d->r_n_hits, &d->r_overflow, d->stream, /*replay_only=*/true, &d->host_hits, fx.get());
            if (r && fx && !d->err_secondary) fx->abort(r, "device " + std::to_string(d->device) + ": " + d->err);  // the other devices must not wait for this one at the filter's meeting points
            return r;
        });
        if (rc) { vdf_groups_free(out); return rc; }
        uint32_t overflow_row = 0xFFFFFFFFu;
        for (int k = 0; k < G; k++) overflow_row = std::min(overflow_row, device_ctx(ctx, k)->r_overflow);
        const uint32_t complete_end = std::min(overflow_row, row_end);
        const vdf_hit *hp;
        uint64_t nh;
        if (G == 1) {
            hp = device_ctx(ctx, 0)->host_hits.data();
            nh = std::min(device_ctx(ctx, 0)->r_n_hits, capacity);
        } else {  // every device's list is sorted by (row, col) and the devices own disjoint rows
            merged.clear();
            for (int k = 0; k < G; k++) {
                const vdf_ctx *d = device_ctx(ctx, k);
                const vdf_hit *b = d->host_hits.data(), *e = b + std::min(d->r_n_hits, capacity);
                e = std::lower_bound(b, e, vdf_hit{complete_end, 0u}, hit_less);  // rows >= complete_end are searched again
                const size_t mid = merged.size();
                merged.insert(merged.end(), b, e);
                std::inplace_merge(merged.begin(), merged.begin() + (ptrdiff_t)mid, merged.end(), hit_less);
            }
            hp = merged.data();
            nh = merged.size();
        }
        const uint64_t old_members = out->n_groups ? out->offsets[out->n_groups] : 0;
        const double t_replay = now_ms();
        rc = vdf_replay_self(n, hp, nh, row_begin, complete_end, matched.data(), out);
        replay_ms += now_ms() - t_replay;
        if (rc) { vdf_groups_free(out); return fail(ctx, rc, "replay failed"); }
        if (overflow_row == 0xFFFFFFFFu) {
            row_begin = row_end;
            span = std::min<uint64_t>(n, span * 4);
        } else {
            const uint64_t progress = complete_end - row_begin;
            if (progress == 0) {
                // Not even one row fit: give a single row the whole buffer (its hits are < n), and a larger suspect queue
                // (a row of a dense cluster has as many suspects as the cluster has members).
                span = 1;
                if (capacity < n) capacity = n;
                for (int k = 0; k < G; k++) device_ctx(ctx, k)->cand_scale = std::min<uint64_t>(device_ctx(ctx, k)->cand_scale * 4, 1ull << 24);
            } else {
                span = std::max<uint64_t>(progress * 2, device_ctx(ctx, 0)->tile_rows);
            }
            row_begin = complete_end;
        }
        if (row_begin < n) {  // feed the consumption state back to the devices
            if (!use_bitmap) {
                bitmap.assign((n + 31) / 32, 0u);
                for (size_t i = 0; i < n; i++)
                    if (matched[i]) bitmap[i >> 5] |= 1u << (i & 31);
                use_bitmap = true;
            } else {  // entries consumed by this round = the members of the groups it appended
                const uint64_t new_members = out->n_groups ? out->offsets[out->n_groups] : 0;
                for (uint64_t q = old_members; q < new_members; q++) {
                    const uint64_t i = out->members[q];
                    bitmap[i >> 5] |= 1u << (i & 31);
                }
            }
            rc = for_each_device(ctx, [&](int, vdf_ctx *d) {
                int r = upload(d, d->matched, bitmap.data(), bitmap.size() * 4, d->stream);
                if (r) return r;
                VDF_HIP(d, hipStreamSynchronize(d->stream));  // `bitmap` is rewritten by the next round
                return (int)VDF_OK;
            });
            if (rc) { vdf_groups_free(out); return rc; }
        }
    }
    // statistics: sums over the devices; kernel time = the slowest device's
    ctx->dev_stats.assign((size_t)G, vdf_search_stats{});
    vdf_search_stats agg{};
    for (int k = 0; k < G; k++) {
        device_ctx(ctx, k)->cand_scale = 1;
        const vdf_search_stats &st = device_ctx(ctx, k)->stats;
        ctx->dev_stats[(size_t)k] = st;
        agg.pairs += st.pairs; agg.pairs_computed += st.pairs_computed; agg.n_hits += st.n_hits; agg.n_tiles += st.n_tiles;
        agg.pairs_early_exit += st.pairs_early_exit;
        agg.n_launches = std::max(agg.n_launches, st.n_launches);
        agg.kernel_ms = std::max(agg.kernel_ms, st.kernel_ms);
        agg.early_exit_bits = st.early_exit_bits;
    }
    ctx->stats = agg;
    const double t_fin = now_ms();
    const int rc_fin = vdf_groups_finish_self(out);
    vdf_search_timing tm{};
    for (int k = 0; k < G; k++) {  // device figures: the slowest device's
        const vdf_search_timing &q = device_ctx(ctx, k)->timing;
        tm.prep_ms = std::max(tm.prep_ms, q.prep_ms); tm.stream_ms = std::max(tm.stream_ms, q.stream_ms);
        tm.resolve_ms = std::max(tm.resolve_ms, q.resolve_ms); tm.download_ms = std::max(tm.download_ms, q.download_ms);
        tm.suspects += q.suspects; tm.suspect_capacity = std::max(tm.suspect_capacity, q.suspect_capacity);
        tm.hits_filtered += q.hits_filtered;
    }
    tm.replay_ms = (float)(replay_ms + now_ms() - t_fin);
    tm.total_ms = (float)(now_ms() - t_call);
    ctx->timing = tm;
    ctx->dev_timing.assign((size_t)G, vdf_search_timing{});
    for (int k = 0; k < G; k++) ctx->dev_timing[(size_t)k] = device_ctx(ctx, k)->timing;
    return rc_fin;
}

// search_with_references() with the sorted candidates resident on every device (up_hashes / up_dur) and device k holding
// the references [ref_base[k], ref_base[k] + ref_cnt[k]) of the caller's order in up_ref_hashes / up_ref_dur: every
// device searches its slice; per-device hit lists concatenate in device order = reference input order.
int search_refs_resident(vdf_ctx *ctx, size_t n_cand, const std::vector<size_t> &ref_cnt, const std::vector<size_t> &ref_base,
                         uint32_t tol_int, vdf_groups *out)
{
    const int G = device_count(ctx);
    const uint64_t capacity0 = ctx->hit_capacity;
    const double t_call = now_ms();
    int rc = for_each_device(ctx, [&](int k, vdf_ctx *d) {
        d->r_n_hits = 0;
        d->stats = vdf_search_stats{};
        d->timing = vdf_search_timing{};
        if (ref_cnt[(size_t)k] == 0) return (int)VDF_OK;
        uint64_t capacity = capacity0;
        for (int attempt = 0; attempt < 6; attempt++) {
            int r = search_refs_device_locked(d, d->up_hashes.as<uint64_t>(), d->up_dur.as<uint32_t>(), n_cand,
                                              d->up_ref_hashes.as<uint64_t>(), d->up_ref_dur.as<uint32_t>(),
                                              ref_cnt[(size_t)k], tol_int, (uint32_t)ref_base[(size_t)k], nullptr,
                                              capacity, &d->r_n_hits, d->stream, &d->host_hits);
            if (r == VDF_E_OVERFLOW && d->r_n_hits > capacity) { capacity = d->r_n_hits; continue; }  // every hit is output: size up
            return r;
        }
        return (int)VDF_E_OVERFLOW;
    });
    if (rc) return rc;
    ctx->dev_stats.assign((size_t)G, vdf_search_stats{});
    vdf_search_stats agg{};
    for (int k = 0; k < G; k++) {
        const vdf_search_stats &st = device_ctx(ctx, k)->stats;
        ctx->dev_stats[(size_t)k] = st;
        agg.pairs += st.pairs; agg.pairs_computed += st.pairs_computed; agg.n_hits += st.n_hits; agg.n_tiles += st.n_tiles;
        agg.pairs_early_exit += st.pairs_early_exit;
        agg.n_launches = std::max(agg.n_launches, st.n_launches);
        agg.kernel_ms = std::max(agg.kernel_ms, st.kernel_ms);
        agg.early_exit_bits = st.early_exit_bits;
    }
    ctx->stats = agg;
    vdf_search_timing tm{};
    for (int k = 0; k < G; k++) {
        const vdf_search_timing &q = device_ctx(ctx, k)->timing;
        tm.prep_ms = std::max(tm.prep_ms, q.prep_ms); tm.stream_ms = std::max(tm.stream_ms, q.stream_ms);
        tm.resolve_ms = std::max(tm.resolve_ms, q.resolve_ms); tm.download_ms = std::max(tm.download_ms, q.download_ms);
        tm.suspects += q.suspects; tm.suspect_capacity = std::max(tm.suspect_capacity, q.suspect_capacity);
        tm.hits_filtered += q.hits_filtered;
    }
    ctx->dev_timing.assign((size_t)G, vdf_search_timing{});
    for (int k = 0; k < G; k++) ctx->dev_timing[(size_t)k] = device_ctx(ctx, k)->timing;
    const double t_group = now_ms();
    int rcg;
    if (G == 1) {
        rcg = vdf_groups_from_ref_hits(device_ctx(ctx, 0)->host_hits.data(), device_ctx(ctx, 0)->r_n_hits, out);
    } else {
        std::vector<vdf_hit> all;
        for (int k = 0; k < G; k++) {
            const vdf_ctx *d = device_ctx(ctx, k);
            all.insert(all.end(), d->host_hits.data(), d->host_hits.data() + d->r_n_hits);
        }
        rcg = vdf_groups_from_ref_hits(all.data(), all.size(), out);
    }
    tm.replay_ms = (float)(now_ms() - t_group);
    tm.total_ms = (float)(now_ms() - t_call);
    ctx->timing = tm;
    return rcg;
}

}  // namespace vdf_impl

using namespace vdf_impl;

extern "C" {

int vdf_ctx_create(int device_id, vdf_ctx **out)
{
    if (!out) return VDF_E_INVAL;
    return create_single(device_id, out, &g_create_error);
}

void vdf_ctx_destroy(vdf_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->subs.empty()) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
    }
    delete ctx;  // a multi-GPU parent joins its workers and destroys its sub-contexts (multi.cpp)
}

const char *vdf_last_error(const vdf_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
int vdf_ctx_device(const vdf_ctx *ctx) { return ctx ? ctx->device : -1; }

int vdf_ctx_set_hit_capacity(vdf_ctx *ctx, uint64_t capacity)
{
    if (!ctx || capacity == 0 || capacity > (1ull << 32)) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->hit_capacity = capacity;
    return VDF_OK;
}

int vdf_ctx_last_search_stats(const vdf_ctx *ctx, vdf_search_stats *out)
{
    if (!ctx || !out) return VDF_E_INVAL;
    *out = ctx->stats;
    return VDF_OK;
}

long long vdf_live_device_bytes(void) { return g_live_device_bytes.load(); }
long long vdf_live_pinned_bytes(void) { return g_live_pinned_bytes.load(); }

int vdf_ctx_last_search_timing(const vdf_ctx *ctx, vdf_search_timing *out)
{
    if (!ctx || !out) return VDF_E_INVAL;
    *out = ctx->timing;
    return VDF_OK;
}

uint32_t vdf_row_tile_size(void) { return vdf::kMfmaRowPad; }  // MFMA backend (default); the VALU backend uses 256 x rows-per-lane

#define VDF_SINGLE_DEVICE_ONLY(ctx)                                                                                   \
    if (!(ctx)->subs.empty())                                                                                         \
        return fail((ctx), VDF_E_INVAL, "device-pointer entry points take a single-device context; a multi-GPU context " \
                                        "offers the host-array calls and the *_shards calls")

int vdf_ctx_pin_database(vdf_ctx *ctx, const uint64_t *d_hashes, size_t n)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    ctx->pinned_db = d_hashes;
    ctx->pinned_n = d_hashes ? n : 0;
    ctx->exp_owner = ExpOwner{};  // whatever was expanded before was not covered by this promise
    return VDF_OK;
}

int vdf_hash_frames_u8_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                              uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *d_out_hashes,
                              uint32_t *d_out_dontcare, void *stream)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    return hash_device_locked(ctx, d_frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride, d_out_hashes,
                              d_out_dontcare, s);
}

// Host frames -> hashes.  On a multi-GPU context the clips are split contiguously over the devices (clips are
// independent: no communication); every device stages and hashes its share on its own host thread.
static int hash_host_entry(vdf_ctx *ctx, const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w,
                           uint32_t h, size_t frame_stride, size_t clip_stride, int letterbox, uint64_t *out_hashes,
                           uint32_t *out_crops, uint32_t *out_dontcare)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (frames_per_clip < VDF_DCT_SIZE) return fail(ctx, VDF_E_NOT_ENOUGH_FRAMES, "fewer than 16 frames per clip");
    if (w == 0 || h == 0) return fail(ctx, VDF_E_BAD_DIMS, "zero frame dimension");
    if (frame_stride < (size_t)w * h) return fail(ctx, VDF_E_INVAL, "frame_stride smaller than a frame");
    if (n_clips == 0) return VDF_OK;
    if (!frames || !out_hashes) return fail(ctx, VDF_E_INVAL, "null pointer");
    const int G = device_count(ctx);
    return for_each_device(ctx, [&](int k, vdf_ctx *d) {
        const size_t base = n_clips / (size_t)G, rem = n_clips % (size_t)G;
        const size_t lo = (size_t)k * base + std::min<size_t>((size_t)k, rem), cnt = base + ((size_t)k < rem ? 1 : 0);
        if (cnt == 0) return (int)VDF_OK;
        return hash_host_locked(d, frames + lo * clip_stride, cnt, w, h, frame_stride, clip_stride, letterbox,
                                out_hashes + lo * VDF_HASH_WORDS, out_crops ? out_crops + 4 * lo : nullptr,
                                out_dontcare ? out_dontcare + lo : nullptr);
    });
}

int vdf_hash_frames_u8(vdf_ctx *ctx, const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w,
                       uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *out_hashes,
                       uint32_t *out_dontcare)
{
    return hash_host_entry(ctx, frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride, 0, out_hashes, nullptr,
                           out_dontcare);
}

int vdf_hash_frames_u8_letterbox(vdf_ctx *ctx, const uint8_t *frames, size_t n_clips, uint32_t frames_per_clip,
                                 uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *out_hashes,
                                 uint32_t *out_crops, uint32_t *out_dontcare)
{
    return hash_host_entry(ctx, frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride, 1, out_hashes, out_crops,
                           out_dontcare);
}

int vdf_cropdetect_letterbox_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                    uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint32_t *d_crops,
                                    void *stream)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    if (frames_per_clip == 0) return fail(ctx, VDF_E_NOT_ENOUGH_FRAMES, "no frames to detect a crop on");
    if (w == 0 || h == 0) return fail(ctx, VDF_E_BAD_DIMS, "zero frame dimension");
    if (n_clips == 0) return VDF_OK;
    if (!d_frames || !d_crops) return fail(ctx, VDF_E_INVAL, "null pointer");
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    VDF_HIP(ctx, ctx->crop_work.reserve(vdf::letterbox_work_bytes(std::min(kMaxClipsPerLaunch, n_clips), frames_per_clip)));
    for (size_t c0 = 0; c0 < n_clips; c0 += kMaxClipsPerLaunch)
        VDF_HIP(ctx, vdf::launch_letterbox(d_frames + c0 * clip_stride, std::min(kMaxClipsPerLaunch, n_clips - c0),
                                           frames_per_clip, w, h, frame_stride, clip_stride, d_crops + 4 * c0, ctx->crop_work.as<uint32_t>(),
                                           stream ? (hipStream_t)stream : ctx->stream, ctx->lb_side_strips));
    return VDF_OK;
}

int vdf_hash_frames_u8_cropped_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                      uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                      const uint32_t *crops, uint64_t *d_out_hashes, uint32_t *d_out_dontcare,
                                      void *stream)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    return hash_cropped_locked(ctx, d_frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride, crops,
                               d_out_hashes, d_out_dontcare, stream ? (hipStream_t)stream : ctx->stream);
}

int vdf_hash_frames_u8_letterbox_device(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                        uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                        uint64_t *d_out_hashes, uint32_t *d_out_dontcare, uint32_t *out_crops,
                                        void *stream)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    return letterbox_hash_device_locked(ctx, d_frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride,
                                        d_out_hashes, d_out_dontcare, out_crops,
                                        stream ? (hipStream_t)stream : ctx->stream);
}

int vdf_hash_frames_u8_letterbox_device_async(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                              uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride,
                                              uint64_t *d_out_hashes, uint32_t *d_out_dontcare, uint32_t *d_out_crops,
                                              void *stream)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    return letterbox_hash_device_locked(ctx, d_frames, n_clips, frames_per_clip, w, h, frame_stride, clip_stride,
                                        d_out_hashes, d_out_dontcare, nullptr,
                                        stream ? (hipStream_t)stream : ctx->stream, d_out_crops);
}

int vdf_groups_max_distance(vdf_ctx *ctx, const uint64_t *hashes, size_t n, const uint64_t *ref_hashes, size_t n_ref,
                            const vdf_groups *groups, uint32_t *out_max)
{
    if (!ctx || !groups) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const uint64_t ng = groups->n_groups;
    if (ng == 0) return VDF_OK;
    if (!hashes || !out_max || !groups->offsets || !groups->members) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (ng > 0x7FFFFFFFull) return fail(ctx, VDF_E_INVAL, "too many groups");
    const uint64_t nm = groups->offsets[ng];
    for (uint64_t i = 0; i < nm; i++)
        if (groups->members[i] >= n) return fail(ctx, VDF_E_INVAL, "group member index out of range");
    const bool refs = ref_hashes && groups->ref_index;
    if (refs)
        for (uint64_t g = 0; g < ng; g++)
            if (groups->ref_index[g] >= (int64_t)n_ref) return fail(ctx, VDF_E_INVAL, "reference index out of range");
    // a multi-GPU context runs this small job on its first device (the parent itself owns no stream or scratch)
    vdf_ctx *d = device_ctx(ctx, 0);
    DeviceGuard restore_device;
    VDF_HIP(ctx, hipSetDevice(d->device));
    hipStream_t s = d->stream;
    DevBuf d_off, d_mem, d_ref, d_out;
    int rc = upload(d, d->up_hashes, hashes, n * VDF_HASH_WORDS * 8, s);
    if (rc == VDF_OK && refs) rc = upload(d, d->up_ref_hashes, ref_hashes, n_ref * VDF_HASH_WORDS * 8, s);
    if (rc == VDF_OK) rc = upload(d, d_off, groups->offsets, (ng + 1) * 8, s);
    if (rc == VDF_OK) rc = upload(d, d_mem, groups->members, std::max<uint64_t>(nm, 1) * 8, s);
    if (rc == VDF_OK && refs) rc = upload(d, d_ref, groups->ref_index, ng * 8, s);
    hipError_t e = rc == VDF_OK ? d_out.reserve(ng * 4) : hipSuccess;
    if (rc == VDF_OK && e == hipSuccess)
        e = vdf::launch_group_max_distance(d->up_hashes.as<uint32_t>(), d_off.as<unsigned long long>(),
                                           d_mem.as<unsigned long long>(),
                                           refs ? d->up_ref_hashes.as<uint32_t>() : nullptr,
                                           refs ? d_ref.as<long long>() : nullptr, (uint32_t)ng, d_out.as<uint32_t>(), s);
    if (rc == VDF_OK && e == hipSuccess) e = hipMemcpyAsync(out_max, d_out.p, ng * 4, hipMemcpyDeviceToHost, s);
    if (rc == VDF_OK && e == hipSuccess) e = hipStreamSynchronize(s);
    d_off.release(); d_mem.release(); d_ref.release(); d_out.release();
    if (rc) { if (d != ctx) ctx->err = d->err; return rc; }
    if (e != hipSuccess) return fail_hip(ctx, e, "vdf_groups_max_distance");
    return VDF_OK;
}


int vdf_search_self_device(vdf_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_durations, size_t n,
                           uint32_t tol_int, uint32_t shard_index, uint32_t shard_count, uint32_t row_begin,
                           uint32_t row_end, const uint32_t *d_matched, vdf_hit *hits, uint64_t capacity,
                           uint64_t *n_hits, uint32_t *overflow_row, void *stream)
{
    if (!ctx || !n_hits || !overflow_row || (capacity && !hits)) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    ctx->stats = vdf_search_stats{};
    ctx->timing = vdf_search_timing{};
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    return search_core(ctx, 0, d_hashes, d_durations, n, d_hashes, d_durations, nullptr, n, tol_int, shard_index,
                       shard_count, row_begin, row_end, d_matched, 0, hits, capacity, n_hits, overflow_row, s);
}

namespace {
// the caller's callbacks (include/vdf.h: vdf_shard_exchange) behind the library's exchange interface
struct CallbackExchange final : vdf_impl::ShardExchange {
    const vdf_shard_exchange *x;
    explicit CallbackExchange(const vdf_shard_exchange *x_) : x(x_) {}
    int agree(uint32_t, vdf_ctx *d, bool *all_complete, uint64_t *total_hits) override
    {
        int c = *all_complete ? 1 : 0;
        const int rc = x->agree(x->user, &c, total_hits);
        if (rc) return fail(d, rc, "the shard exchange's agree callback failed");
        *all_complete = c != 0;
        return VDF_OK;
    }
    int or_bitmap(uint32_t, vdf_ctx *d, uint32_t *d_bitmap, size_t n_words, hipStream_t stream) override
    {
        const int rc = x->or_bitmap(x->user, d_bitmap, n_words, (void *)stream);
        if (rc) return fail(d, rc, "the shard exchange's or_bitmap callback failed");
        return VDF_OK;
    }
};
}  // namespace

int vdf_search_self_device_replay(vdf_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_durations, size_t n,
                                  uint32_t tol_int, uint32_t shard_index, uint32_t shard_count, uint32_t row_begin,
                                  uint32_t row_end, const uint32_t *d_matched, vdf_hit *hits, uint64_t capacity,
                                  uint64_t *n_hits, uint32_t *overflow_row, const vdf_shard_exchange *xchg, void *stream)
{
    if (!ctx || !n_hits || !overflow_row || (capacity && !hits)) return VDF_E_INVAL;
    if (xchg && (!xchg->agree || !xchg->or_bitmap)) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    ctx->stats = vdf_search_stats{};
    ctx->timing = vdf_search_timing{};
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    CallbackExchange fx(xchg);
    return search_core(ctx, 0, d_hashes, d_durations, n, d_hashes, d_durations, nullptr, n, tol_int, shard_index,
                       shard_count, row_begin, row_end, d_matched, 0, hits, capacity, n_hits, overflow_row, s,
                       /*replay_only=*/true, nullptr, xchg ? &fx : nullptr);
}

int vdf_bitmap_or_device(vdf_ctx *ctx, uint32_t *d_dst, const uint32_t *d_srcs, size_t n_words, uint32_t n_srcs, void *stream)
{
    // takes no lock and touches no scratch of the context: it is what an or_bitmap callback calls from INSIDE
    // vdf_search_self_device_replay, which holds the context's lock
    if (!ctx || !ctx->subs.empty()) return VDF_E_INVAL;
    if (n_words == 0 || n_srcs == 0) return VDF_OK;
    if (!d_dst || !d_srcs) return VDF_E_INVAL;
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    VDF_HIP(ctx, vdf::launch_bitmap_or(d_dst, d_srcs, n_words, n_srcs, stream ? (hipStream_t)stream : ctx->stream));
    return VDF_OK;
}

int vdf_search_refs_device(vdf_ctx *ctx, const uint64_t *d_cand_hashes, const uint32_t *d_cand_durations,
                           size_t n_cand, const uint64_t *d_ref_hashes, const uint32_t *d_ref_durations, size_t n_ref,
                           uint32_t tol_int, uint32_t ref_index_base, vdf_hit *hits, uint64_t capacity,
                           uint64_t *n_hits, void *stream)
{
    if (!ctx || !n_hits || (capacity && !hits)) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    return search_refs_device_locked(ctx, d_cand_hashes, d_cand_durations, n_cand, d_ref_hashes, d_ref_durations,
                                     n_ref, tol_int, ref_index_base, hits, capacity, n_hits,
                                     stream ? (hipStream_t)stream : ctx->stream);
}

int vdf_sort_order_device(vdf_ctx *ctx, const uint32_t *d_durations, const uint32_t *d_path_rank, size_t n, uint32_t *d_perm_out,
                          void *stream)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    if (n == 0) return VDF_OK;
    if (!d_durations || !d_perm_out) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    VDF_HIP(ctx, ctx->sort_scratch_pub.reserve(vdf::sort_order_scratch_bytes((uint32_t)n, d_path_rank != nullptr)));
    VDF_HIP(ctx, vdf::launch_sort_order(d_durations, d_path_rank, (uint32_t)n, d_perm_out, ctx->sort_scratch_pub.p, ctx->sort_scratch_pub.cap,
                                        stream ? (hipStream_t)stream : ctx->stream));
    return VDF_OK;
}

int vdf_apply_order_device(vdf_ctx *ctx, const uint64_t *d_hashes, const uint32_t *d_durations, const uint32_t *d_perm, size_t n,
                           uint64_t *d_hashes_out, uint32_t *d_durations_out, void *stream)
{
    if (!ctx) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VDF_SINGLE_DEVICE_ONLY(ctx);
    if (n == 0) return VDF_OK;
    if (!d_hashes || !d_perm || !d_hashes_out || (d_durations_out && !d_durations)) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    if (d_hashes == d_hashes_out || (d_durations && d_durations == d_durations_out)) return fail(ctx, VDF_E_INVAL, "the gather is not in place");
    VDF_HIP(ctx, hipSetDevice(ctx->device));
    VDF_HIP(ctx, vdf::launch_gather_hashes(d_hashes, d_durations, d_perm, (uint32_t)n, d_hashes_out, d_durations_out,
                                           stream ? (hipStream_t)stream : ctx->stream));
    return VDF_OK;
}

int vdf_search_self(vdf_ctx *ctx, const uint64_t *hashes, const uint32_t *durations, size_t n, uint32_t tol_int,
                    vdf_groups *out)
{
    if (!ctx || !out) return VDF_E_INVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::memset(out, 0, sizeof *out);
    ctx->stats = vdf_search_stats{};
    if (n == 0) return vdf_groups_finish_self(out);  // search_algorithm.rs:89-91
    if (!hashes || !durations) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    // the windows are binary searches over the durations: the arrays must be in Search::sort order (search_algorithm.rs:55-61)
    if (!is_sorted_u32(durations, n)) return fail(ctx, VDF_E_INVAL, "durations are not ascending: pass the arrays in Search::sort order");
    // every device receives the whole database straight from the host arrays (no collective needed)
    int rc = for_each_device(ctx, [&](int, vdf_ctx *d) {
        VDF_HIP(d, hipSetDevice(d->device));
        int r = upload(d, d->up_hashes, hashes, n * VDF_HASH_WORDS * 8, d->stream);
        if (r == VDF_OK) r = upload(d, d->up_dur, durations, n * 4, d->stream);
        return r;
    });
    if (rc) return rc;
    return search_self_resident(ctx, n, tol_int, out);
}

int vdf_search_refs(vdf_ctx *ctx, const uint64_t *cand_hashes, const uint32_t *cand_durations, size_t n_cand,
                    const uint64_t *ref_hashes, const uint32_t *ref_durations, size_t n_ref, uint32_t tol_int,
                    vdf_groups *out)
{
    if (!ctx || !out) return VDF_E_INVAL;
    std::memset(out, 0, sizeof *out);
    if (n_cand == 0 || n_ref == 0) return vdf_groups_from_ref_hits(nullptr, 0, out);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!cand_hashes || !cand_durations || !ref_hashes || !ref_durations) return fail(ctx, VDF_E_INVAL, "null pointer");
    if (n_cand >= 0xFFFFFFFFull || n_ref >= 0xFFFFFFFFull) return fail(ctx, VDF_E_INVAL, "more than 2^32-1 hashes");
    if (!is_sorted_u32(cand_durations, n_cand))
        return fail(ctx, VDF_E_INVAL, "candidate durations are not ascending: pass the candidates in Search::sort order");
    const int G = device_count(ctx);
    std::vector<size_t> cnt((size_t)G), base((size_t)G);
    for (int k = 0; k < G; k++) {  // contiguous, order-preserving split of the references
        const size_t b = n_ref / (size_t)G, rem = n_ref % (size_t)G;
        base[(size_t)k] = (size_t)k * b + std::min<size_t>((size_t)k, rem);
        cnt[(size_t)k] = b + ((size_t)k < rem ? 1 : 0);
    }
    int rc = for_each_device(ctx, [&](int k, vdf_ctx *d) {
        VDF_HIP(d, hipSetDevice(d->device));
        int r = upload(d, d->up_hashes, cand_hashes, n_cand * VDF_HASH_WORDS * 8, d->stream);
        if (r == VDF_OK) r = upload(d, d->up_dur, cand_durations, n_cand * 4, d->stream);
        if (r == VDF_OK) r = upload(d, d->up_ref_hashes, ref_hashes + base[(size_t)k] * VDF_HASH_WORDS, cnt[(size_t)k] * VDF_HASH_WORDS * 8, d->stream);
        if (r == VDF_OK) r = upload(d, d->up_ref_dur, ref_durations + base[(size_t)k], cnt[(size_t)k] * 4, d->stream);
        return r;
    });
    if (rc) return rc;
    return search_refs_resident(ctx, n_cand, cnt, base, tol_int, out);
}

}  // extern "C"
