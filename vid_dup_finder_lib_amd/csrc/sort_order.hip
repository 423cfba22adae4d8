// Search::sort on the device (vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:55-61): the stable order by
// (duration, src_path) of a hash database that is already resident in HBM - the step between hashing on the GPUs and
// searching there (vid_dup_finder_app/src/app/app_fns.rs:428-482 builds the Search from the cache and sorts it once).
// Paths stay with the caller: what decides between equal durations is each entry's RANK among the caller's paths in
// Rust's component-wise PathBuf order (equal paths -> equal ranks -> input order, sort_by_key is stable); without ranks all
// paths count as equal.
//
// The sort is a hand-written stable LSD radix sort, 8 bits per pass (round 6; rocPRIM's generic radix_sort_* was 627 kernel
// instantiations and 6.5 MB of an 8.2 MB library - its code object loaded on every process's first call - for keys of at most a
// few 10^7 entries that are sorted once per search): one histogram launch for all passes, one scatter launch per pass (look-back
// over the tiles in front).  A pass whose digit is the same in every key (the high bytes of durations that are seconds, of indices
// below 2^24) degenerates to a straight copy, decided on the device.
#include <algorithm>
#include <cstring>

#include "vdf_internal.h"

namespace vdf {

__global__ __launch_bounds__(256) void sort_keys_kernel(const uint32_t *__restrict__ dur, const uint32_t *__restrict__ rank,
                                                        uint32_t n, uint64_t *__restrict__ keys64,
                                                        uint32_t *__restrict__ idx)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (idx) idx[i] = (uint32_t)i;
        if (keys64) keys64[i] = ((uint64_t)dur[i] << 32) | rank[i];
    }
}

// out[k] = in[perm[k]]: one 16-byte piece per lane, 8 lanes per hash -> both sides coalesce in 128-byte lines
__global__ __launch_bounds__(256) void gather_hashes_kernel(const uint4 *__restrict__ in, const uint32_t *__restrict__ dur_in,
                                                            const uint32_t *__restrict__ perm, uint32_t n,
                                                            uint4 *__restrict__ out, uint32_t *__restrict__ dur_out)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < (size_t)n * 8; t += (size_t)gridDim.x * 256) {
        const uint32_t k = (uint32_t)(t >> 3), q = (uint32_t)t & 7u;
        const uint32_t src = perm[k];
        out[t] = in[(size_t)src * 8 + q];
        if (q == 0 && dur_out) dur_out[k] = dur_in[src];
    }
}

// ---- stable LSD radix sort, 8 bits per pass, ONE scatter launch per pass ("onesweep": chained scan with decoupled look-back) --------------
// Tile = 4096 keys per workgroup of four waves; wave w owns the tile's w-th quarter and walks it in sixteen rounds of 64 keys, so the
// order inside a tile is (wave, round, lane) = index order and every step below keeps equal digits in index order (stability).
// Launches per sort: one memset (tile status + tickets + histograms), one histogram kernel for ALL passes, one scatter per pass.  (The
// first form - histogram, per-digit scan, digit bases and scatter as four launches per pass - cost a reference search of the configs[4]
// shape 0.1 ms: its two sorts of 1e5 keys were 40 launches.)  In the scatter a tile takes a ticket (tiles are numbered in the order
// they start, so every predecessor of a tile is running or done), publishes its per-digit counts, and thread d looks back over the tiles
// before it - adding AGGREGATE counts until it meets an inclusive PREFIX - for the number of keys of digit d in front of its tile.
constexpr uint32_t kRadixTile = 4096, kRadixRounds = 16, kRadixMaxPasses = 8;
constexpr unsigned long long kFlagAggregate = 1ull << 62, kFlagPrefix = 2ull << 62, kFlagMask = 3ull << 62;

struct RadixWork {                 // device scratch of one sort (radix_work_bytes), zeroed by one memset per sort
    uint32_t *hist;                // [kRadixMaxPasses][256] keys per digit, per pass
    uint32_t *ticket;              // [kRadixMaxPasses] next tile number of each pass's scatter
    unsigned long long *status;    // [n_tiles][256]: flag (2 bits) | pass tag (6 bits) | count or inclusive prefix (32 bits)
};

struct RadixShifts { unsigned s[kRadixMaxPasses]; int n; };

// digit counts of every pass at once; a workgroup accumulates many tiles in LDS before it touches the global counters
template <class Key>
__global__ __launch_bounds__(256) void radix_hist_kernel(const Key *__restrict__ keys, size_t n, RadixShifts sh, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t s_hist[kRadixMaxPasses][256];
    for (int p = 0; p < sh.n; p++) s_hist[p][threadIdx.x] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const Key k = keys[i];
        for (int p = 0; p < sh.n; p++) atomicAdd(&s_hist[p][(uint32_t)(k >> sh.s[p]) & 255u], 1u);
    }
    __syncthreads();
    for (int p = 0; p < sh.n; p++) {
        const uint32_t c = s_hist[p][threadIdx.x];
        if (c) atomicAdd(&hist[p * 256 + threadIdx.x], c);
    }
}

// block-wide exclusive scan of one value per thread (256 threads); *total = the sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= (uint32_t)o) inc += up;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t t = s_wave[w];
        before += w < wave ? t : 0u;
        all += t;
    }
    __syncthreads();  // s_wave may be reused by the caller's next call
    *total = all;
    return before + inc - v;
}

// One pass: the keys (and values) of a tile to their places - base of the digit (scan of the pass's histogram) + keys of that digit in the
// tiles before this one (look-back) + in earlier waves of this tile + in earlier rounds of this wave + in lower lanes of this round.
// iota: the values are the keys' indices (first pass).  tag = pass + 1 (a status word of another pass, or the memset's zero, is "not ready").
template <class Key, bool HAS_VAL>
__global__ __launch_bounds__(256) void radix_onesweep_kernel(const Key *__restrict__ kin, const uint32_t *__restrict__ vin, Key *__restrict__ kout,
                                                             uint32_t *__restrict__ vout, size_t n, unsigned shift, const uint32_t *__restrict__ hist,
                                                             uint32_t *__restrict__ ticket, unsigned long long *__restrict__ status,
                                                             unsigned long long tag, int iota)
{
    __shared__ uint32_t s_run[4][256];  // per wave: keys of each digit in the wave's quarter, then the running offset while it is walked
    __shared__ uint32_t s_base[256];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_tile;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) s_run[w][tid] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const size_t first = (size_t)tile * kRadixTile + (size_t)wave * (kRadixTile / 4) + lane;
    Key k[kRadixRounds];
    uint32_t v[kRadixRounds];
#pragma unroll
    for (uint32_t r = 0; r < kRadixRounds; r++) {
        const size_t i = first + (size_t)r * 64;
        k[r] = i < n ? kin[i] : (Key)0;
        if (HAS_VAL) v[r] = iota ? (uint32_t)i : (i < n ? vin[i] : 0u);
    }
    const uint32_t total_d = hist[tid];  // keys of digit tid in the whole array (this pass's row of the histogram)
    if (__syncthreads_or((size_t)total_d == n)) {  // every key has the same digit: the order does not change
#pragma unroll
        for (uint32_t r = 0; r < kRadixRounds; r++) {
            const size_t i = first + (size_t)r * 64;
            if (i < n) { kout[i] = k[r]; if (HAS_VAL) vout[i] = v[r]; }
        }
        return;
    }
#pragma unroll
    for (uint32_t r = 0; r < kRadixRounds; r++)
        if (first + (size_t)r * 64 < n) atomicAdd(&s_run[wave][(uint32_t)(k[r] >> shift) & 255u], 1u);
    __syncthreads();
    {   // thread d = digit d
        uint32_t local = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) { const uint32_t c = s_run[w][tid]; s_run[w][tid] = local; local += c; }  // exclusive over the waves
        unsigned long long *mine = status + (size_t)tile * 256 + tid;
        if (tile != 0) __hip_atomic_store(mine, kFlagAggregate | tag | local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t ignore;
        const uint32_t digit_base = block_exclusive_scan(total_d, s_wave, &ignore);
        uint32_t before = 0;  // keys of this digit in tiles 0 .. tile - 1
        for (uint32_t t = tile; t-- > 0;) {
            unsigned long long w;
            do {
                w = __hip_atomic_load(status + (size_t)t * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((w & (63ull << 56)) != tag || (w & kFlagMask) == 0);
            before += (uint32_t)w;
            if ((w & kFlagMask) == kFlagPrefix) break;
        }
        __hip_atomic_store(mine, kFlagPrefix | tag | (unsigned long long)(before + local), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base[tid] = digit_base + before;
    }
    __syncthreads();
    uint32_t *run = s_run[wave];
#pragma unroll
    for (uint32_t r = 0; r < kRadixRounds; r++) {
        const bool valid = first + (size_t)r * 64 < n;
        const uint32_t d = (uint32_t)(k[r] >> shift) & 255u;
        // the lanes of this round that hold the same digit (8 ballots), this lane's rank among them
        uint64_t same = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(((d >> b) & 1u) != 0);
            same &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
        const uint32_t at = valid ? run[d] : 0u;  // (read by every lane of the group BEFORE its first lane advances it: LDS is in order per wave)
        if (valid) {
            const size_t pos = (size_t)s_base[d] + at + rank;
            kout[pos] = k[r];
            if (HAS_VAL) vout[pos] = v[r];
            if (rank == 0) run[d] = at + (uint32_t)__builtin_popcountll(same);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

static size_t radix_work_bytes(size_t n)
{
    const size_t n_tiles = (n + kRadixTile - 1) / kRadixTile;
    return kRadixMaxPasses * 256 * 4 + 256 + n_tiles * 256 * 8 + 256;
}

static RadixWork radix_work(void *p, size_t n)
{
    (void)n;
    char *c = static_cast<char *>(p);
    RadixWork w;
    w.hist = reinterpret_cast<uint32_t *>(c);
    w.ticket = reinterpret_cast<uint32_t *>(c + kRadixMaxPasses * 256 * 4);
    w.status = reinterpret_cast<unsigned long long *>(c + kRadixMaxPasses * 256 * 4 + 256);
    return w;
}

// One pass per entry of shifts (least significant digit first).  Pass 0 reads (k_in, v_in or the indices), pass p > 0 reads what pass
// p - 1 wrote; passes write to (k_a, v_a), (k_b, v_b) alternately, the LAST pass's values to v_final when it is given.
// Returns where the last pass's keys went (0: k_a, 1: k_b).  work_base: the RadixWork's memory (256-byte aligned, radix_work_bytes(n)).
template <class Key, bool HAS_VAL>
static hipError_t radix_sort_lsd(const Key *k_in, const uint32_t *v_in, Key *k_a, uint32_t *v_a, Key *k_b, uint32_t *v_b, uint32_t *v_final,
                                 size_t n, const unsigned *shifts, int n_shifts, const RadixWork &w, hipStream_t stream, int *last_in_b)
{
    const uint32_t n_tiles = (uint32_t)((n + kRadixTile - 1) / kRadixTile);
    if (n_shifts > (int)kRadixMaxPasses) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(w.hist, 0, radix_work_bytes(n) - 256, stream);  // histograms, tickets, tile status
    if (e != hipSuccess) return e;
    RadixShifts sh{};
    sh.n = n_shifts;
    for (int p = 0; p < n_shifts; p++) sh.s[p] = shifts[p];
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipLaunchKernelGGL((radix_hist_kernel<Key>), dim3(std::min<uint32_t>((uint32_t)((n + 2047) / 2048), (uint32_t)cus * 4u)), dim3(256), 0, stream,
                       k_in, n, sh, w.hist);
    const Key *src_k = k_in;
    const uint32_t *src_v = v_in;
    for (int p = 0; p < n_shifts; p++) {
        Key *dst_k = (p & 1) ? k_b : k_a;
        uint32_t *dst_v = (p & 1) ? v_b : v_a;
        if (HAS_VAL && p == n_shifts - 1 && v_final) dst_v = v_final;
        hipLaunchKernelGGL((radix_onesweep_kernel<Key, HAS_VAL>), dim3(n_tiles), dim3(256), 0, stream, src_k, src_v, dst_k, dst_v, n, shifts[p],
                           w.hist + p * 256, w.ticket + p, w.status, (unsigned long long)(p + 1) << 56, (HAS_VAL && p == 0 && v_in == nullptr) ? 1 : 0);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        src_k = dst_k;
        src_v = dst_v;
    }
    if (last_in_b) *last_in_b = (n_shifts - 1) & 1;
    return hipSuccess;
}

// scratch: [keys A n x (u32 | u64)][keys B][vals A n u32][vals B][keys in n u64 if with rank][work]; perm_out receives the order.
size_t sort_order_scratch_bytes(uint32_t n, bool with_rank)
{
    const size_t a = 255, kb = ((size_t)n * (with_rank ? 8 : 4) + a) & ~a, vb = ((size_t)n * 4 + a) & ~a;
    return 2 * kb + 2 * vb + (with_rank ? kb : 0) + radix_work_bytes(n) + 256;
}

hipError_t launch_sort_order(const uint32_t *dur, const uint32_t *rank, uint32_t n, uint32_t *perm_out, void *scratch,
                             size_t scratch_bytes, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const bool with_rank = rank != nullptr;
    if (scratch_bytes < sort_order_scratch_bytes(n, with_rank)) return hipErrorInvalidValue;
    const size_t a = 255, kb = ((size_t)n * (with_rank ? 8 : 4) + a) & ~a, vb = ((size_t)n * 4 + a) & ~a;
    char *p = static_cast<char *>(scratch);
    void *k_a = p; p += kb;
    void *k_b = p; p += kb;
    uint32_t *v_a = reinterpret_cast<uint32_t *>(p); p += vb;
    uint32_t *v_b = reinterpret_cast<uint32_t *>(p); p += vb;
    uint64_t *keys_in = nullptr;
    if (with_rank) { keys_in = reinterpret_cast<uint64_t *>(p); p += kb; }
    const RadixWork w = radix_work(p, n);
    unsigned shifts[8];
    int ns = 0;
    if (with_rank) {
        hipLaunchKernelGGL(sort_keys_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 4096u)), dim3(256), 0, stream, dur, rank, n, keys_in,
                           (uint32_t *)nullptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        // a rank is below n (path_order.cpp: ranks are positions in the sorted list of distinct paths)... but the caller's array is the
        // caller's: all 32 bits are sorted, and the passes whose byte is the same everywhere cost a copy
        for (unsigned b = 0; b < 64; b += 8) shifts[ns++] = b;
        return radix_sort_lsd<uint64_t, true>(keys_in, nullptr, (uint64_t *)k_a, v_a, (uint64_t *)k_b, v_b, perm_out, n, shifts, ns, w, stream, nullptr);
    }
    for (unsigned b = 0; b < 32; b += 8) shifts[ns++] = b;
    return radix_sort_lsd<uint32_t, true>(dur, nullptr, (uint32_t *)k_a, v_a, (uint32_t *)k_b, v_b, perm_out, n, shifts, ns, w, stream, nullptr);
}

hipError_t launch_gather_hashes(const uint64_t *hashes, const uint32_t *dur, const uint32_t *perm, uint32_t n, uint64_t *hashes_out,
                                uint32_t *dur_out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const size_t total = (size_t)n * 8;
    hipLaunchKernelGGL(gather_hashes_kernel, dim3((uint32_t)std::min<size_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, stream,
                       reinterpret_cast<const uint4 *>(hashes), dur, perm, n, reinterpret_cast<uint4 *>(hashes_out), dur_out);
    return hipGetLastError();
}

// ---- Search::sort's PATH half on the device (round 6) ----------------------------------------------------------------------------
// PathBuf order of the paths of a decoded cache (blob + offsets, as vdf_cache_soa holds them), without ranks and without the host:
// for PLAIN paths (path_order.cpp: is_plain - what a directory walk produces: no empty, "." or ".." component, no NUL byte)
// std::path's component order is the byte order in which '/' sorts below every other byte and a path that ends sorts before one that
// goes on.  That is an LSD radix sort over 8-byte words of the paths, last word first: byte b of a path becomes END 0 < '/' 1 <
// bytes 0x01 .. 0x2E as b + 1 < bytes 0x30 .. 0xFF as they are (NUL does not occur, so the codes fit a byte), eight codes big-endian in a
// u64 key; every round is the stable radix sort above over (key, entry).  Equal paths keep their input order.  The duration sort that
// follows (stable, by the same machinery) makes it the (duration, path) order of search_algorithm.rs:55-61.
struct PathFacts {           // filled by path_facts_kernel
    uint32_t not_plain;      // some path is not plain (the caller falls back to the host's component comparator)
    uint32_t max_len;        // longest path, bytes
    uint32_t shared;         // leading bytes every path shares with the first one
    uint32_t pad;
};

__global__ __launch_bounds__(256) void path_facts_kernel(const char *__restrict__ blob, const unsigned long long *__restrict__ off,
                                                         const uint32_t *__restrict__ sel, uint32_t n, PathFacts *__restrict__ facts)
{
    const uint32_t first = sel ? sel[0] : 0u;
    const char *p0 = blob + off[first];
    const size_t l0 = (size_t)(off[first + 1] - off[first]);
    uint32_t bad = 0, longest = 0, shared = 0xFFFFFFFFu;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t e = sel ? sel[i] : (uint32_t)i;
        const char *s = blob + off[e];
        const size_t len = (size_t)(off[e + 1] - off[e]);
        longest = max(longest, (uint32_t)min(len, (size_t)0xFFFFFFFFu));
        // is_plain (path_order.cpp), and the bytes shared with the first path
        if (!(len == 0 || (len == 1 && s[0] == '/'))) {
            if (s[len - 1] == '/') bad = 1;
            size_t clen = 0, dots = 0;
            for (size_t k = s[0] == '/' ? 1 : 0; k < len; k++) {
                const char c = s[k];
                if (c == '/') {
                    if (clen == 0 || (dots == clen && clen <= 2)) bad = 1;
                    clen = dots = 0;
                } else {
                    if (c == 0) bad = 1;
                    clen++;
                    dots += c == '.';
                }
            }
            if (dots == clen && clen <= 2) bad = 1;
        }
        size_t m = 0;
        const size_t lim = min(min(len, l0), (size_t)shared);
        while (m < lim && s[m] == p0[m]) m++;
        shared = (uint32_t)m;
    }
    if (bad) atomicOr(&facts->not_plain, 1u);
    atomicMax(&facts->max_len, longest);
    atomicMin(&facts->shared, shared);
}

// key of word `word` (bytes 8 word .. 8 word + 7) of the path of the candidate at position pos[i] (entry sel[pos[i]], or pos[i] itself)
__global__ __launch_bounds__(256) void path_keys_kernel(const char *__restrict__ blob, const unsigned long long *__restrict__ off,
                                                        const uint32_t *__restrict__ sel, const uint32_t *__restrict__ pos, uint32_t n,
                                                        uint32_t word, uint64_t *__restrict__ keys)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t e = sel ? sel[pos[i]] : pos[i];
        const unsigned long long o = off[e];
        const size_t len = (size_t)(off[e + 1] - o);
        const unsigned char *s = reinterpret_cast<const unsigned char *>(blob + o);
        uint64_t key = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const size_t at = (size_t)word * 8 + j;
            uint32_t code = 0;  // END
            if (at < len) {
                const uint32_t b = s[at];
                code = b == '/' ? 1u : b < '/' ? b + 1u : b;
            }
            key = (key << 8) | code;
        }
        keys[i] = key;
    }
}

__global__ __launch_bounds__(256) void iota_or_copy_kernel(const uint32_t *__restrict__ src, uint32_t n, uint32_t *__restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src ? src[i] : (uint32_t)i;
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n,
                                                         uint32_t *__restrict__ dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[idx[i]];
}

size_t path_order_scratch_bytes(uint32_t n)
{
    const size_t a = 255, kb = ((size_t)n * 8 + a) & ~a, vb = ((size_t)n * 4 + a) & ~a;
    return 2 * kb + 3 * vb + 256 + radix_work_bytes(n) + 256;
}

hipError_t launch_path_facts(const char *d_blob, const unsigned long long *d_off, const uint32_t *d_sel, uint32_t n, void *d_facts16, hipStream_t stream)
{
    PathFacts init{0u, 0u, 0xFFFFFFFFu, 0u};
    hipError_t e = hipMemcpyAsync(d_facts16, &init, sizeof init, hipMemcpyHostToDevice, stream);  // (16 bytes from the stack: copied before the call returns)
    if (e != hipSuccess || n == 0) return e;
    hipLaunchKernelGGL(path_facts_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 8192u)), dim3(256), 0, stream, d_blob, d_off, d_sel, n,
                       static_cast<PathFacts *>(d_facts16));
    return hipGetLastError();
}

// order_out[k] = which of the n candidates (candidate c = entry d_sel[c], or entry c when d_sel is null; duration d_dur[c]) stands at
// position k of the stable order by (duration, path).  Paths must be plain, at most 8 * n_words bytes long, and share their first
// 8 * first_word bytes.
hipError_t launch_path_duration_order(const char *d_blob, const unsigned long long *d_off, const uint32_t *d_sel, const uint32_t *d_dur, uint32_t n,
                                      uint32_t first_word, uint32_t n_words, uint32_t *order_out, void *scratch, size_t scratch_bytes,
                                      hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (scratch_bytes < path_order_scratch_bytes(n)) return hipErrorInvalidValue;
    const size_t a = 255, kb = ((size_t)n * 8 + a) & ~a, vb = ((size_t)n * 4 + a) & ~a;
    char *p = static_cast<char *>(scratch);
    uint64_t *k_in = reinterpret_cast<uint64_t *>(p); p += kb;   // this round's keys, in the current order
    uint64_t *k_out = reinterpret_cast<uint64_t *>(p); p += kb;  // where the passes write keys (both buffers alternate inside a round)
    uint32_t *v[3];
    for (auto &q : v) { q = reinterpret_cast<uint32_t *>(p); p += vb; }
    p += 256;
    const RadixWork w = radix_work(p, n);
    const dim3 grid(std::min<uint32_t>((n + 255) / 256, 16384u));
    unsigned shifts[8];
    for (unsigned b = 0; b < 8; b++) shifts[b] = 8 * b;
    // v[cur] holds the entries in the current order
    int cur = 0;
    hipLaunchKernelGGL(iota_or_copy_kernel, grid, dim3(256), 0, stream, (const uint32_t *)nullptr, n, v[0]);
    for (uint32_t word = n_words; word-- > first_word;) {
        hipLaunchKernelGGL(path_keys_kernel, grid, dim3(256), 0, stream, d_blob, d_off, d_sel, v[cur], n, word, k_in);
        // eight passes: pass 0 reads (k_in, v[cur]) and writes (k_out, A); pass 1 (k_out, A) -> (k_in, B); ... an even count ends in (k_in, B)
        uint32_t *A = v[(cur + 1) % 3], *B = v[(cur + 2) % 3];
        hipError_t e = radix_sort_lsd<uint64_t, true>(k_in, v[cur], k_out, A, k_in, B, nullptr, n, shifts, 8, w, stream, nullptr);
        if (e != hipSuccess) return e;
        cur = (cur + 2) % 3;
    }
    // stable by duration: keys = the durations in the current (path) order
    uint32_t *dur_in = reinterpret_cast<uint32_t *>(k_in), *dk_a = reinterpret_cast<uint32_t *>(k_out), *dk_b = dk_a + n;
    hipLaunchKernelGGL(gather_u32_kernel, grid, dim3(256), 0, stream, d_dur, v[cur], n, dur_in);
    return radix_sort_lsd<uint32_t, true>(dur_in, v[cur], dk_a, v[(cur + 1) % 3], dk_b, v[(cur + 2) % 3], order_out, n, shifts, 4, w, stream, nullptr);
}

// ---- hit lists into (row, col) order on the device ---------------------------------------------------------------
// Dense near-duplicates produce 1e6 - 1e7 thresholded pairs per search; the host replay (search_algorithm.rs:131-170) and the
// reference grouping want them in (row, col) order, and a host radix sort of 1e7 pairs costs as much as the search kernel.
// vdf_hit is {row, col}: as a little-endian u64 the row is the LOW half - the passes simply take the col's bytes (bits 32 ..) first
// and the row's bytes (bits 0 ..) last; no swapped copy of the keys is made.
size_t sort_hits_scratch_bytes(size_t n) { return ((n * 8 + 255) & ~(size_t)255) + radix_work_bytes(n) + 256; }

// rows are < 2^row_bits, columns < 2^col_bits (fewer passes)
hipError_t launch_sort_hits(vdf_hit *hits, size_t n, unsigned row_bits, void *scratch, size_t scratch_bytes, hipStream_t stream, unsigned col_bits)
{
    if (n < 2) return hipSuccess;
    if (scratch_bytes < sort_hits_scratch_bytes(n)) return hipErrorInvalidValue;
    uint64_t *other = static_cast<uint64_t *>(scratch);
    const RadixWork w = radix_work(static_cast<char *>(scratch) + ((n * 8 + 255) & ~(size_t)255), n);
    unsigned shifts[8];
    int ns = 0;
    for (unsigned b = 0; b < std::min(std::max(col_bits, 1u), 32u); b += 8) shifts[ns++] = 32 + b;
    for (unsigned b = 0; b < std::min(std::max(row_bits, 1u), 32u); b += 8) shifts[ns++] = b;
    uint64_t *h = reinterpret_cast<uint64_t *>(hits);
    int in_b = 0;
    // pass 0: hits -> scratch, pass 1: scratch -> hits, ...
    hipError_t e = radix_sort_lsd<uint64_t, false>(h, nullptr, other, nullptr, h, nullptr, nullptr, n, shifts, ns, w, stream, &in_b);
    if (e != hipSuccess) return e;
    if (!in_b) e = hipMemcpyAsync(h, other, n * 8, hipMemcpyDeviceToDevice, stream);  // an odd number of passes ended in the scratch
    return e;
}

}  // namespace vdf
