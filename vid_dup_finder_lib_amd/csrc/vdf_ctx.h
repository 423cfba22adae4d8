// The private definition of vdf_ctx (include/vdf.h keeps it opaque) and the internal entry points api.cpp and
// multi.cpp share.  A context is either ONE device (stream, scratch buffers, tables) or a multi-GPU parent that owns
// one single-device sub-context plus one host worker thread per listed device (multi.cpp).
#pragma once
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "resize_tables.h"
#include <atomic>

#include "vdf_internal.h"

// bytes of device memory the library's growable buffers hold right now, over all contexts of the process (vdf_live_device_bytes: a
// context that forgets a buffer in its destructor shows up as a difference around create / use / destroy)
inline std::atomic<long long> g_live_device_bytes{0};
inline std::atomic<long long> g_live_pinned_bytes{0};  // the same for page-locked host staging (vdf_live_pinned_bytes)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        release();
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        g_live_device_bytes += (long long)cap;
        return hipSuccess;
    }
    void release()
    {
        if (p) { (void)hipFree(p); g_live_device_bytes -= (long long)cap; }
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Pinned host staging (hipHostMalloc); falls back to pageable memory if pinning fails.
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned = false;
    bool reserve(size_t bytes)
    {
        if (bytes <= cap) return true;
        release();
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) {
            pinned = true;
            g_live_pinned_bytes += (long long)bytes;
        } else {
            (void)hipGetLastError();
            p = std::malloc(bytes);
            pinned = false;
            if (!p) return false;
        }
        cap = bytes;
        return true;
    }
    void release()
    {
        if (p) { if (pinned) { (void)hipHostFree(p); g_live_pinned_bytes -= (long long)cap; } else std::free(p); }
        p = nullptr; cap = 0; pinned = false;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct DeviceAxisTable {
    DevBuf start, size, w;
    vdf::HostAxisTable host;
};

struct DeviceMfmaTable {
    DevBuf operand, bias, meta;  // meta: the band form's kt_lo[16], nt[16]
    vdf::MfmaAxisTable host;
};

// Every crop-box size of one small frame size (w <= 256, at most two 64-row groups), resident: what lets the letterbox path go from detect
// to hash without showing the boxes to the host (api.cpp: box_table_set).  blob: per index one table - operand (n_tiles x {hi, lo} x 1 KB),
// bias[16], {precision, n_tiles} padded to 64 B; index bw = horizontal table of box width bw (1 .. w), index w + 1 + bh = vertical
// (kMfmaLayoutVertical) table of box height bh (1 .. h).  entries: the same as vdf::CropTableEntry[w + h + 2] (device pointers into blob).
// one_tile (w, h <= 64): every table is vdf::kSmallBoxTableStride bytes, so the fused kernel finds a table by multiplication.
struct BoxTableSet {
    DevBuf blob, entries;
    bool usable = false;  // false: some box size's coefficients do not fit the i8 split - the caller keeps the host-planned route
    bool one_tile = false;
};

namespace vdf_impl {
struct Worker;     // one host thread bound to one device (multi.cpp)
struct RcclState;  // communicators of a multi-GPU context (multi.cpp)
struct CopyPool;   // host threads that gather caller frames into pinned staging (hash_host.cpp)

// The steps every shard of ONE sharded search() launch enters together, so that the replay filter (hamming.hip: has_in / covered are
// properties of the COMPLETE hit set) also works when the row tiles are dealt over several devices or processes.  Implementations:
// the worker threads of a multi-GPU context (multi.cpp: LocalExchange) and the caller's callbacks (include/vdf.h: vdf_shard_exchange).
// Every shard makes the same sequence of calls; abort() releases the others when a shard leaves early with an error.
struct ShardExchange {
    virtual ~ShardExchange() = default;
    // in: this shard's "my hit list is complete" and its hit count; out: AND / sum over all shards
    virtual int agree(uint32_t shard, vdf_ctx *d, bool *all_complete, uint64_t *total_hits) = 0;
    // d_bitmap[0 .. n_words) |= every other shard's, in place, ordered on `stream` (the marking kernels were queued there)
    // may return kExchangeOff (on EVERY shard alike): the exchange could not be carried out - go on without the filter
    virtual int or_bitmap(uint32_t shard, vdf_ctx *d, uint32_t *d_bitmap, size_t n_words, hipStream_t stream) = 0;
    // the calling shard leaves early with this status / message: release the others (they report it as the cause)
    virtual void abort(int rc, const std::string &msg) { (void)rc; (void)msg; }
};
constexpr int kExchangeOff = 1;  // not an error (every vdf_status error is negative)
}

// What the fp4 expansion in exp_cols was made from (reused only for a database the caller pinned)
struct ExpOwner {
    const void *hashes = nullptr;
    size_t n = 0;
    uint32_t k_steps = 0;
    const void *buffer = nullptr;  // exp_cols.p at the time: a reallocation invalidates
    bool operator==(const ExpOwner &o) const { return hashes == o.hashes && n == o.n && k_steps == o.k_steps && buffer == o.buffer && hashes; }
};

struct vdf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // H2D staging of the next batch under the current batch's kernels (hash path).  Made by the first call that has more than one
    // batch: HIP spreads a process's streams over FOUR hardware queues per GPU by default (GPU_MAX_HW_QUEUES), and streams that share one
    // wait for each other - with two streams per context, the batching queue's two slots and their parent context were six, and a slot's
    // detect kernels sat behind the other slot's 265 MB of frames (profiles/r06_hash_queue.txt).
    hipStream_t copy_stream = nullptr;
    bool one_stream = false;  // the batching queue's slot contexts: transfers on `stream` even in a call of several batches
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_copy[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    std::mutex mu;
    std::string err;
    uint64_t hit_capacity = 1ull << 24;
    vdf_search_stats stats{};
    vdf_search_timing timing{};
    hipEvent_t ev_mid = nullptr;  // between the distance kernel and the suspect resolution
    hipEvent_t ev_wait = nullptr; // waits of a millisecond and more in the hashing calls (wait_event: a short poll, then the thread sleeps)
    bool spin_wait = false;       // VDF_SPIN_WAIT: those waits spin like every other (as before round 6: A/B runs)
    bool no_link_turns = false;   // VDF_NO_LINK_TURNS: bulk host-to-device transfers of different contexts may interleave (as before round 6: A/B runs)
    uint32_t tile_rows = 256 * vdf::kDefaultRowsPerLane;
    uint32_t chunk_cols = vdf::kDefaultChunkCols;
    // search scratch
    DevBuf row_lo, row_hi, tile_lo, tile_hi, tile_first, tile_count, tile_offset, counters, hits, perm, matched;
    DevBuf up_hashes, up_dur, up_ref_hashes, up_ref_dur;
    DevBuf hits2, hit_bitmaps;  // replay filter: surviving hits, has-incoming / covered bitmaps
    DevBuf bitmap_gather;       // sharded replay filter: every shard's copy of one bitmap, before the local OR
    DevBuf sort_scratch;  // keys / indices / rocPRIM temporary storage of the device-side sorts inside a search call
    DevBuf sort_scratch_pub;  // the same for vdf_sort_order_device: the caller may sort on one stream and search on another
    // hash scratch
    DevBuf small, frames, frames2, out_hashes, out_hashes2, out_dc, out_dc2, cos_table, crops, crop_desc, crop_tables, crop_desc2, crop_tables2, crop_work;
    PinBuf pin[2], pin_out[2];
    PinBuf pin_ctrl;   // search: the counters of a launch (pageable destinations make hipMemcpyAsync synchronous)
    PinBuf pin_crops, pin_desc;  // letterbox: the detect's boxes on their way down, the crop descriptors on their way up (20 000 clips: 0.3 + 0.6 MB)
    const void *pinned_db = nullptr;  // vdf_ctx_pin_database: the caller promises these n x 16 words do not change until unpinned
    size_t pinned_n = 0;
    ExpOwner exp_owner;
    size_t refs_hint_cols = 0, refs_hint_rows = 0;  // shape of the last reference search and the chunk width it settled on
    uint32_t refs_hint_chunk = 0;
    bool no_hit_filter = false;  // VDF_NO_HIT_FILTER: host-level search() downloads and replays every thresholded pair
    uint64_t hits_guess = 0;  // hits of the previous launch: how much of the list is fetched together with the counters
    PinBuf pin_small;  // search: reference durations / permutation and small hit lists (pageable copies of 0.4 MB cost 0.3-1 ms each)
    std::map<uint32_t, DeviceAxisTable *> axis_tables;
    std::map<uint64_t, DeviceMfmaTable *> mfma_tables;  // key = in_size * 4 + layout (resize_tables.h)
    std::map<uint64_t, BoxTableSet *> box_tables;       // key = w << 32 | h
    bool no_device_path_order = false;  // VDF_NO_DEVICE_PATH_ORDER: vdf_search_cache_entries orders paths on the host (vdf_path_ranks), as before round 6
    bool lb_host_plan = false;     // VDF_LB_HOST_PLAN: small frames' boxes visit the host between detect and hash (as before round 6: A/B runs)
    bool no_lb_fused = false;      // VDF_NO_LB_FUSED: frames of at most 64 x 64 take detect kernels + cropped kernel instead of the fused kernel
    int hash_no_persistent = 0, hash_wgs_per_cu = 3;
    bool hash_wgs_per_cu_set = false;  // VDF_HASH_WGS_PER_CU given (the fused letterbox kernel otherwise takes what fits)
    uint32_t mfma_chunk_cols = 0, mfma_group = 8192;  // 0 = pick the chunk width per search (search_core); VDF_MFMA_CHUNK_COLS overrides
    DevBuf group_cmin, group_offset, group_blocks;
    uint32_t mfma_min_wgs = 8192;  // adaptive chunk width: at least this many (row tile, chunk) workgroups (VDF_MFMA_MIN_WGS)
    uint64_t cand_scale = 1;              // grows (x4) while a reference search retries after a suspect-queue overflow
    uint32_t cand_capacity_override = 0;  // VDF_CAND_CAPACITY: suspect-queue entries (0 = sized from the admitted pairs)
    uint32_t mfma_self_rows = 512;  // rows per workgroup of search() (VDF_MFMA_SELF_ROWS: 256 | 512)
    uint32_t mfma_refs_rows = 256;  // rows per workgroup of reference searches (VDF_MFMA_REFS_ROWS: 256 | 512)
    int mfma_prune_step = -1;  // -1 = from the tolerance, 16 = off (VDF_MFMA_PRUNE_STEP)
    int search_backend = 1;  // 0 = XOR + popcount on the VALU, 1 = {0, 1} fp4 Gram matrix on the matrix cores (both exact)
    DevBuf exp_cols, exp_rows, pop_cols, pop_rows, cand;
    size_t cand_dirty = SIZE_MAX;  // slots of the candidate queue the last launch may have written (SIZE_MAX: never initialised)
    // dispatch switches of measurements and tests, read ONCE per context (create_single): getenv is not safe against a concurrent setenv
    // (Python writing os.environ while the multi-GPU worker threads hash), and decision and launcher must see the same value
    int wavestream_knob = 0;       // resize_dispatch.h: -1 = VDF_NO_WAVESTREAM, n > 0 = VDF_WAVESTREAM_NW=n
    bool no_rowcrop = false, rowcrop_all = false, no_boxstream = false;  // VDF_NO_ROWCROP / VDF_ROWCROP_ALL / VDF_NO_BOXSTREAM
    bool no_smallcrop = false;  // VDF_NO_SMALLCROP: small frames' crop boxes through the one-workgroup-per-frame kernel (as before round 5)
    int lb_side_strips = 0;        // VDF_LB_NC16: 16 = the side walk never takes the 32-strip form
    int copy_threads = 0;          // VDF_COPY_THREADS (0: half the hardware threads, at most 8)
    size_t host_chunk_bytes = 32u << 20;  // VDF_HOST_CHUNK_MB: pinned staging chunk (x 2) of the host-frame path
    bool host_direct = true;       // VDF_HOST_DIRECT=0: packed input goes through the library's staging too
    bool force_rccl = false;       // VDF_FORCE_RCCL: a one-device context replicates through librccl (multi-GPU parent)
    bool test_exchange_fail = false;  // VDF_TEST_EXCHANGE_FAIL (multi-GPU parent): the replay filter's bitmap exchange reports a failure - the search must degrade to the unfiltered list
    bool err_secondary = false;    // this context's last error is "another device of the sharded launch failed" (for_each_device reports the root cause instead)
    int resize_mode = 0;  // 0 auto, 1 generic scalar kernel, 3 MFMA fused kernel, 4 MFMA per-frame kernel with whole-line loads, 5 MFMA linear-stream kernel where it applies, 6 its K-split form where it applies
    // hit list of the host-level calls: pinned (a 50 MB list comes down at the link rate), sized by what searches actually
    // produce - 64 k entries to begin with, grown to a launch's list once its length is known (search_core) - not by the hit
    // capacity: the default 16 M entries would be 128 MB of page-locked memory per device for lists that are usually empty
    struct HostHits {
        PinBuf buf;
        size_t size() const { return buf.cap / sizeof(vdf_hit); }
        bool resize(size_t n) { return buf.reserve(n * sizeof(vdf_hit)); }
        vdf_hit *data() const { return buf.as<vdf_hit>(); }
    } host_hits;
    vdf_impl::CopyPool *copy_pool = nullptr;
    // results of the last fan-out round on this device (multi-GPU parent reads them after the workers join)
    uint64_t r_n_hits = 0;
    uint32_t r_overflow = 0xFFFFFFFFu;
    int r_rc = 0;

    // ---- multi-GPU parent only ----
    std::vector<vdf_ctx *> subs;                  // one single-device context per listed device (devices may repeat)
    std::vector<vdf_impl::Worker *> workers;      // one host thread per sub-context
    vdf_impl::RcclState *rccl = nullptr;
    std::vector<vdf_search_stats> dev_stats;      // per-device statistics of the last search
    std::vector<vdf_search_timing> dev_timing;    // per-device phase times / filter counts of the last search

    ~vdf_ctx();
};

namespace vdf_impl {

void set_create_error(const std::string &msg);  // what vdf_last_error(NULL) returns on this thread
int fail(vdf_ctx *ctx, int code, const std::string &msg);
int fail_hip(vdf_ctx *ctx, hipError_t e, const char *what);

#define VDF_HIP(ctx, call)                                                      \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return vdf_impl::fail_hip((ctx), e__, #call);    \
    } while (0)

// Entry points that walk over several devices on the CALLER's thread put its current HIP device back on every exit path
// (a torch caller would otherwise silently continue on the last listed GPU).
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) { saved = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

inline bool hit_less(const vdf_hit &a, const vdf_hit &b) { return a.row != b.row ? a.row < b.row : a.col < b.col; }

// ---- single-device building blocks (api.cpp); the caller holds the lock of the context it passes -----------------
int create_single(int device_id, vdf_ctx **out, std::string *err);
int upload(vdf_ctx *ctx, DevBuf &buf, const void *src, size_t bytes, hipStream_t stream);
// Waits for an event of the hashing calls' bulk phases (frames crossing the PCIe link, the detect pass of a large batch): polls for
// ~50 us, then sleeps until the event's interrupt (ev_copy / ev_done / ev_wait are created with hipEventBlockingSync).  A spinning
// hipEventSynchronize / hipStreamSynchronize holds a core for the whole transfer - two cores per GPU under the batching queue - which
// the callers' decoders and staging copies need (measured under a 16-CPU quota: tools/bench_hash_queue.cpp, profiles/r06_hash_queue.txt).
int wait_event(vdf_ctx *ctx, hipEvent_t ev);
std::mutex &link_mutex(int device);  // one per GPU of the process: whose bulk host-to-device transfer has the PCIe link (hash_host.cpp)
// windows + tiles, distance kernel, hit download (sorted by (row, col)); mode 0 = self, 1 = references
// replay_only: the hits feed nothing but the greedy replay of search(), so hits that provably cannot matter to it may be
// dropped on the device (then *n_hits_out = the number kept; stats.n_hits still counts every thresholded pair)
int search_core(vdf_ctx *ctx, int mode, const uint64_t *d_col_hashes, const uint32_t *d_col_dur, size_t n_cols,
                const uint64_t *d_row_hashes, const uint32_t *d_row_dur, const uint32_t *d_row_perm, size_t n_rows,
                uint32_t tol_int, uint32_t shard_index, uint32_t shard_count, uint32_t row_begin, uint32_t row_end,
                const uint32_t *d_matched, uint32_t row_index_base, vdf_hit *hits, uint64_t capacity,
                uint64_t *n_hits_out, uint32_t *overflow_row_out, hipStream_t stream, bool replay_only = false,
                vdf_ctx::HostHits *staging = nullptr,  // staging: the list goes to this growable pinned buffer instead of `hits`
                ShardExchange *fx = nullptr);          // replay_only launches of shard_count > 1: how the shards meet for the filter
int search_refs_device_locked(vdf_ctx *ctx, const uint64_t *d_cand_hashes, const uint32_t *d_cand_durations,
                              size_t n_cand, const uint64_t *d_ref_hashes, const uint32_t *d_ref_durations, size_t n_ref,
                              uint32_t tol_int, uint32_t ref_index_base, vdf_hit *hits, uint64_t capacity,
                              uint64_t *n_hits, hipStream_t s, vdf_ctx::HostHits *staging = nullptr);
int hash_device_locked(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip, uint32_t w,
                       uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *d_out, uint32_t *d_dc,
                       hipStream_t stream);
// host frames -> hashes on ONE device: pinned, double-buffered staging (copy of batch k + 1 under the kernels of batch k)
int hash_host_locked(vdf_ctx *ctx, const uint8_t *frames, size_t n_clips, uint32_t w, uint32_t h, size_t frame_stride,
                     size_t clip_stride, int letterbox, uint64_t *out_hashes, uint32_t *out_crops, uint32_t *out_dontcare);
// The greedy search() loop over a database that is already resident on every device of the context
// (d_hashes(dev) / d_dur(dev) give each device's replica): overflow protocol, merge, replay.
int search_self_resident(vdf_ctx *ctx, size_t n, uint32_t tol_int, vdf_groups *out);
int search_refs_resident(vdf_ctx *ctx, size_t n_cand, const std::vector<size_t> &ref_cnt, const std::vector<size_t> &ref_base,
                         uint32_t tol_int, vdf_groups *out);
int letterbox_hash_device_locked(vdf_ctx *ctx, const uint8_t *d_frames, size_t n_clips, uint32_t frames_per_clip,
                                 uint32_t w, uint32_t h, size_t frame_stride, size_t clip_stride, uint64_t *d_out,
                                 uint32_t *d_dc, uint32_t *out_crops, hipStream_t stream, uint32_t *d_out_crops = nullptr);
void destroy_copy_pool(vdf_ctx *ctx);
bool is_sorted_u32(const uint32_t *d, size_t n);

// ---- multi-GPU parent (multi.cpp) ----------------------------------------------------------------------------
// Runs f(k, device context) for every device: inline for a single-device context, on the workers otherwise.
// Returns the first non-zero status (its message is copied to ctx->err).
int for_each_device(vdf_ctx *ctx, const std::function<int(int, vdf_ctx *)> &f);
inline int device_count(const vdf_ctx *ctx) { return ctx->subs.empty() ? 1 : (int)ctx->subs.size(); }
inline vdf_ctx *device_ctx(vdf_ctx *ctx, int k) { return ctx->subs.empty() ? ctx : ctx->subs[k]; }
void destroy_multi(vdf_ctx *ctx);  // joins the workers, destroys communicators and sub-contexts
// the exchange of the context's own worker threads (one per launch round); nullptr for a single-device context
ShardExchange *make_local_exchange(vdf_ctx *ctx);

}  // namespace vdf_impl
