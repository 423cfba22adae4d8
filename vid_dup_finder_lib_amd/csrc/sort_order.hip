// Search::sort on the device (vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:55-61): the stable order by
// (duration, src_path) of a hash database that is already resident in HBM - the step between hashing on the GPUs and
// searching there (vid_dup_finder_app/src/app/app_fns.rs:428-482 builds the Search from the cache and sorts it once).
// Paths stay with the caller: what decides between equal durations is each entry's RANK among the caller's paths in
// Rust's component-wise PathBuf order (equal paths -> equal ranks -> input order, sort_by_key is stable); without ranks all
// paths count as equal.
//
// The sort is a hand-written stable LSD radix sort, 8 bits per pass (round 6; rocPRIM's generic radix_sort_* was 627 kernel
// instantiations and 6.5 MB of an 8.2 MB library - its code object loaded on every process's first call - for keys of at most a
// few 10^7 entries that are sorted once per search).  Per pass four small launches: per-tile digit histogram, per-digit scan over
// the tiles, digit bases, stable scatter.  A pass whose digit is the same in every key (the high bytes of durations that are
// seconds, of indices below 2^24) degenerates to a straight copy, decided on the device.
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

// ---- stable LSD radix sort, 8 bits per pass ---------------------------------------------------------------------------------------
// Tile = 4096 keys per workgroup of four waves; wave w owns the tile's w-th quarter and walks it in sixteen rounds of 64 keys, so the
// order inside a tile is (wave, round, lane) = index order and every step below keeps equal digits in index order (stability).
constexpr uint32_t kRadixTile = 4096, kRadixRounds = 16;

struct RadixWork {       // device scratch of one sort (radix_work_bytes)
    uint32_t *counts;    // [256][n_tiles]: per-tile digit counts, then (scan) their exclusive prefix along the tiles of each digit
    uint32_t *totals;    // [256] keys per digit
    uint32_t *bases;     // [256] exclusive prefix of totals, [256] = 1 if one digit holds every key (the pass is a copy)
};

template <class Key>
__global__ __launch_bounds__(256) void radix_hist_kernel(const Key *__restrict__ keys, size_t n, unsigned shift, uint32_t *__restrict__ counts,
                                                         uint32_t n_tiles)
{
    __shared__ uint32_t s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kRadixTile;
    for (uint32_t r = 0; r < kRadixRounds; r++) {
        const size_t i = base + (size_t)r * 256 + threadIdx.x;  // (any order: a histogram)
        if (i < n) atomicAdd(&s_hist[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * n_tiles + blockIdx.x] = s_hist[threadIdx.x];
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

// workgroup d: counts[d][0 .. n_tiles) -> its exclusive prefix, totals[d] = its sum
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t *__restrict__ counts, uint32_t n_tiles, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t s_wave[4];
    uint32_t *row = counts + (size_t)blockIdx.x * n_tiles;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n_tiles; c0 += 256) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < n_tiles ? row[i] : 0u;
        uint32_t sum;
        const uint32_t ex = block_exclusive_scan(v, s_wave, &sum);
        if (i < n_tiles) row[i] = carry + ex;
        carry += sum;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void radix_bases_kernel(const uint32_t *__restrict__ totals, uint32_t *__restrict__ bases, size_t n)
{
    __shared__ uint32_t s_wave[4];
    const uint32_t t = totals[threadIdx.x];
    uint32_t sum;
    bases[threadIdx.x] = block_exclusive_scan(t, s_wave, &sum);
    const bool whole = (size_t)t == n;
    if (__syncthreads_or(whole)) { if (threadIdx.x == 0) bases[256] = 1u; }
    else if (threadIdx.x == 0) bases[256] = 0u;
}

// keys (and values) of tile blockIdx.x to their places: base of the digit + keys of that digit in earlier tiles + in earlier waves of
// this tile + in earlier rounds of this wave + in lower lanes of this round.  iota: the values are the keys' indices (first pass).
template <class Key, bool HAS_VAL>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const Key *__restrict__ kin, const uint32_t *__restrict__ vin, Key *__restrict__ kout,
                                                            uint32_t *__restrict__ vout, size_t n, unsigned shift,
                                                            const uint32_t *__restrict__ counts, const uint32_t *__restrict__ bases,
                                                            uint32_t n_tiles, int iota)
{
    __shared__ uint32_t s_run[4][256];  // per wave: keys of each digit in the wave's quarter, then the running offset while it is walked
    __shared__ uint32_t s_base[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t first = (size_t)blockIdx.x * kRadixTile + (size_t)wave * (kRadixTile / 4) + lane;
    Key k[kRadixRounds];
    uint32_t v[kRadixRounds];
#pragma unroll
    for (uint32_t r = 0; r < kRadixRounds; r++) {
        const size_t i = first + (size_t)r * 64;
        k[r] = i < n ? kin[i] : (Key)0;
        if (HAS_VAL) v[r] = iota ? (uint32_t)i : (i < n ? vin[i] : 0u);
    }
    if (bases[256]) {  // every key has the same digit: the order does not change (workgroup-uniform)
#pragma unroll
        for (uint32_t r = 0; r < kRadixRounds; r++) {
            const size_t i = first + (size_t)r * 64;
            if (i < n) { kout[i] = k[r]; if (HAS_VAL) vout[i] = v[r]; }
        }
        return;
    }
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) s_run[w][tid] = 0;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kRadixRounds; r++)
        if (first + (size_t)r * 64 < n) atomicAdd(&s_run[wave][(uint32_t)(k[r] >> shift) & 255u], 1u);
    __syncthreads();
    {   // thread d: where digit d of this tile begins; the waves' counts become their exclusive prefix over the waves
        s_base[tid] = bases[tid] + counts[(size_t)tid * n_tiles + blockIdx.x];
        uint32_t before = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) { const uint32_t c = s_run[w][tid]; s_run[w][tid] = before; before += c; }
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
    return ((256 * n_tiles * 4 + 255) & ~(size_t)255) + 256 * 4 + 260 * 4 + 256;
}

static RadixWork radix_work(void *p, size_t n)
{
    const size_t n_tiles = (n + kRadixTile - 1) / kRadixTile;
    char *c = static_cast<char *>(p);
    RadixWork w;
    w.counts = reinterpret_cast<uint32_t *>(c);
    c += (256 * n_tiles * 4 + 255) & ~(size_t)255;
    w.totals = reinterpret_cast<uint32_t *>(c);
    w.bases = w.totals + 256;
    return w;
}

// One pass per entry of shifts (least significant digit first).  Pass 0 reads (k_in, v_in or the indices), pass p > 0 reads what pass
// p - 1 wrote; passes write to (k_a, v_a), (k_b, v_b) alternately, the LAST pass's values to v_final when it is given.
// Returns where the last pass's keys went (0: k_a, 1: k_b).
template <class Key, bool HAS_VAL>
static hipError_t radix_sort_lsd(const Key *k_in, const uint32_t *v_in, Key *k_a, uint32_t *v_a, Key *k_b, uint32_t *v_b, uint32_t *v_final,
                                 size_t n, const unsigned *shifts, int n_shifts, const RadixWork &w, hipStream_t stream, int *last_in_b)
{
    const uint32_t n_tiles = (uint32_t)((n + kRadixTile - 1) / kRadixTile);
    const Key *src_k = k_in;
    const uint32_t *src_v = v_in;
    for (int p = 0; p < n_shifts; p++) {
        Key *dst_k = (p & 1) ? k_b : k_a;
        uint32_t *dst_v = (p & 1) ? v_b : v_a;
        if (HAS_VAL && p == n_shifts - 1 && v_final) dst_v = v_final;
        hipLaunchKernelGGL((radix_hist_kernel<Key>), dim3(n_tiles), dim3(256), 0, stream, src_k, n, shifts[p], w.counts, n_tiles);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(256), 0, stream, w.counts, n_tiles, w.totals);
        hipLaunchKernelGGL(radix_bases_kernel, dim3(1), dim3(256), 0, stream, w.totals, w.bases, n);
        hipLaunchKernelGGL((radix_scatter_kernel<Key, HAS_VAL>), dim3(n_tiles), dim3(256), 0, stream, src_k, src_v, dst_k, dst_v, n, shifts[p],
                           w.counts, w.bases, n_tiles, (HAS_VAL && p == 0 && v_in == nullptr) ? 1 : 0);
        hipError_t e = hipGetLastError();
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
