// Tiled XOR + popcount Hamming search for gfx950 (MI355X), wave64.
//
// Replaces the O(n^2) loops of the reference:
//   search_self  hot loop  vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:150-156
//   search_one   hot loop  vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:67-74
//   hamming_distance       vid_dup_finder_lib/src/video_hashing/video_hash.rs:311-317
//
// Layout of one workgroup (4 waves, 256 threads), R = rows per lane:
//   - every lane keeps R target hashes (R x 32 dwords) in VGPRs for the whole tile;
//   - candidate hashes are wave-uniform: they are fetched with scalar loads (s_load_dwordx16 x2
//     through the scalar cache) and used as the SGPR operand of v_xor_b32; v_bcnt_u32_b32
//     accumulates the popcount for free.  64 VALU lane-ops per pair, no LDS, no VGPR traffic;
//   - per candidate one v_min + one v_cmp + one scalar branch guard the rare slow path that
//     applies the duration window / consumption bitmap and appends (row, col) to the hit buffer.
// The kernel is VALU-bound (see DESIGN.md "Hamming kernel"): HBM traffic is ~0.25 B per pair.
#include <algorithm>

#include "vdf_internal.h"

namespace vdf {

constexpr uint32_t kMaxBlocksPerLaunch = 4u << 20;  // x <= 512 threads <= 2^31 work-items, under HIP's 2^32 grid limit

typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;  // forces s_load for uniform addresses

__device__ __forceinline__ uint32_t sat_u32(double x)
{  // Rust `f64 as u32`: truncating, saturating, NaN -> 0
    if (!(x == x)) return 0u;
    if (x <= 0.0) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// first index with !(d[idx] < v)
__device__ __forceinline__ uint32_t partition_point_lt(const uint32_t *__restrict__ d, uint32_t n, uint32_t v)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (d[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first index with !(d[idx] <= v)
__device__ __forceinline__ uint32_t partition_point_le(const uint32_t *__restrict__ d, uint32_t n, uint32_t v)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (d[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}

// Append the hits of the calling wave (lanes with `hit`) behind ONE atomic on the shared counter; must be reached by all
// active lanes together.  With dense near-duplicates (1e7 hits per search) per-hit atomics on one address serialise in L2.
__device__ __forceinline__ void wave_append_hits(bool hit, uint32_t row, uint32_t col, vdf_hit *__restrict__ hits,
                                                 unsigned long long capacity, unsigned long long *__restrict__ counters,
                                                 uint32_t *__restrict__ overflow_row)
{
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
    if (bal == 0ull) return;
    const int leader = __builtin_ctzll(bal);
    unsigned long long base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&counters[0], (unsigned long long)__builtin_popcountll(bal));
    base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), leader, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, leader, 64);
    if (hit) {
        const unsigned long long at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (at < capacity) {
            vdf_hit hp; hp.row = row; hp.col = col;
            hits[at] = hp;
        } else {
            atomicMin(overflow_row, row);
        }
    }
}

// One workgroup per row tile: per-row candidate windows, per-tile candidate range and chunk count.
//   mode 0 (search_self): row i -> [i + 1, first j with dur[j] > (f64(dur[i]) * 1.1) as u32)
//                         (search_algorithm.rs:93-117: rhs pointer; candidates are the later entries)
//   mode 1 (search_one):  ref r -> [partition_point(dur < (d*0.95) as u32), partition_point(dur <= (d*1.05) as u32))
//                         (search_algorithm.rs:173-185)
__global__ __launch_bounds__(256) void windows_tiles_kernel(
    int mode, const uint32_t *__restrict__ col_dur, uint32_t n_cols, const uint32_t *__restrict__ row_dur,
    const uint32_t *__restrict__ row_perm, uint32_t n_rows, uint32_t row_begin, uint32_t row_end, uint32_t shard_index,
    uint32_t shard_count, uint32_t tile_rows, uint32_t chunk_cols, uint32_t *__restrict__ row_lo,
    uint32_t *__restrict__ row_hi, uint32_t *__restrict__ tile_lo, uint32_t *__restrict__ tile_hi,
    uint32_t *__restrict__ tile_first, uint32_t *__restrict__ tile_count, unsigned long long *__restrict__ counters)
{
    __shared__ uint32_t s_min[4], s_max[4];
    __shared__ unsigned long long s_pairs[4];
    const uint32_t t = blockIdx.x;
    const bool mine = (t % shard_count) == shard_index;
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    unsigned long long pairs = 0;
    for (uint32_t q = threadIdx.x; q < tile_rows; q += 256) {
        const uint32_t p = t * tile_rows + q;
        uint32_t lo = 0, hi = 0;
        if (mine && p < n_rows && p >= row_begin && p < row_end) {
            const uint32_t r = row_perm ? row_perm[p] : p;
            const double d = (double)row_dur[r];
            if (mode == 0) {
                lo = p + 1;
                hi = partition_point_le(col_dur, n_cols, sat_u32(d * 1.1));
            } else {
                lo = partition_point_lt(col_dur, n_cols, sat_u32(d * 0.95));
                hi = partition_point_le(col_dur, n_cols, sat_u32(d * 1.05));
            }
            if (hi <= lo) { lo = 0; hi = 0; }
        }
        row_lo[p] = lo;
        row_hi[p] = hi;
        if (hi > lo) {
            mn = min(mn, lo);
            mx = max(mx, hi);
            pairs += hi - lo;
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pairs += __shfl_xor(pairs, o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_min[wave] = mn; s_max[wave] = mx; s_pairs[wave] = pairs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        pairs = s_pairs[0] + s_pairs[1] + s_pairs[2] + s_pairs[3];
        uint32_t cnt = 0, first = 0;
        if (mx > mn) {
            first = mn / chunk_cols;
            cnt = (mx + chunk_cols - 1) / chunk_cols - first;
        } else {
            mn = 0; mx = 0;
        }
        tile_lo[t] = mn;
        tile_hi[t] = mx;
        tile_first[t] = first;
        tile_count[t] = cnt;
        if (pairs) atomicAdd(&counters[2], pairs);
    }
}

// counters[5] != 0 <=> the candidate durations are not ascending (the device-pointer entry points cannot check on the host).
__global__ __launch_bounds__(256) void check_sorted_kernel(const uint32_t *__restrict__ d, uint32_t n,
                                                           unsigned long long *__restrict__ counters)
{
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 1 < n; i += (size_t)gridDim.x * 256) bad |= d[i] > d[i + 1];
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) counters[5] = 1ull;
}

// Exclusive scan of tile_count[0..n) into tile_offset[0..n]; single workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void scan_tiles_kernel(const uint32_t *__restrict__ tile_count, uint32_t n,
                                                          uint32_t *__restrict__ tile_offset)
{
    __shared__ uint32_t s_part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b = min(tid * per, n), e = min(b + per, n);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; i++) sum += tile_count[i];
    s_part[tid] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
        uint32_t v = (tid >= o) ? s_part[tid - o] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (uint32_t i = b; i < e; i++) { tile_offset[i] = run; run += tile_count[i]; }
    if (tid == 1023) tile_offset[n] = s_part[1023];
}

// Grouped, chunk-major work order for the MFMA kernel: G consecutive row tiles form a group; inside a group
// the workgroups walk candidate chunk by candidate chunk, so the ~2 x 256 workgroups resident at any time all
// stream the SAME few MB of candidates and find them in their XCD's L2 instead of HBM.  Workgroups whose
// (row tile, chunk) is outside the tile's range exit at once (a few % of the grid near the diagonal).
__global__ __launch_bounds__(256) void group_tiles_kernel(const uint32_t *__restrict__ tile_first,
                                                          const uint32_t *__restrict__ tile_count, uint32_t n_row_tiles,
                                                          uint32_t group_size, uint32_t shard_count,
                                                          uint32_t *__restrict__ group_cmin,
                                                          uint32_t *__restrict__ group_blocks)
{  // one workgroup per group: chunk range covered by its row tiles; only every shard_count-th tile is this rank's
    __shared__ uint32_t s_min[4], s_max[4];
    const uint32_t g = blockIdx.x;
    uint32_t cmin = 0xFFFFFFFFu, cmax = 0;
    const uint32_t t1 = min((g + 1) * group_size, n_row_tiles);
    for (uint32_t t = g * group_size + threadIdx.x; t < t1; t += 256) {
        if (tile_count[t]) {
            cmin = min(cmin, tile_first[t]);
            cmax = max(cmax, tile_first[t] + tile_count[t]);
        }
    }
    cmin = wave_min(cmin);
    cmax = wave_max(cmax);
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = cmin; s_max[threadIdx.x >> 6] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cmin = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        cmax = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (cmax <= cmin) { cmin = 0; cmax = 0; }
        group_cmin[g] = cmin;
        const uint32_t n_chunks = cmax - cmin;
        group_blocks[g] = n_chunks * ((group_size + shard_count - 1) / shard_count);
    }
}

__global__ void group_scan_kernel(const uint32_t *__restrict__ group_blocks, uint32_t n_groups,
                                  uint32_t *__restrict__ group_offset)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t g = 0; g < n_groups; g++) { group_offset[g] = run; run += group_blocks[g]; }
        group_offset[n_groups] = run;
    }
}

// CHKW (< 32): exact early exit, as in the MFMA kernel below - after CHKW of the 32 dwords, a candidate that every lane's rows
// are already more than `tol` away from cannot be a hit for this wave and the remaining dwords are skipped.
template <int R, int CHKW>
__global__ __launch_bounds__(256) void hamming_tile_kernel(
    const uint32_t *__restrict__ row_hashes, const uint32_t *__restrict__ row_perm, uint32_t n_rows,
    uint32_t row_index_base, const uint32_t *__restrict__ col_hashes, const uint32_t *__restrict__ row_lo,
    const uint32_t *__restrict__ row_hi, const uint32_t *__restrict__ tile_lo, const uint32_t *__restrict__ tile_hi,
    const uint32_t *__restrict__ tile_first, const uint32_t *__restrict__ tile_offset, uint32_t n_row_tiles,
    uint32_t chunk_cols, uint32_t tol, const uint32_t *__restrict__ matched, int self_mode,
    vdf_hit *__restrict__ hits, unsigned long long capacity, unsigned long long *__restrict__ counters,
    uint32_t *__restrict__ overflow_row, uint32_t block_base)
{
    constexpr uint32_t TILE_ROWS = 256 * R;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t bid = blockIdx.x + block_base;  // grids above 2^32 work-items are launched in slices

    // workgroup -> (row tile, candidate chunk): largest t with tile_offset[t] <= blockIdx.x
    const_u32_ptr off = (const_u32_ptr)(uintptr_t)tile_offset;
    uint32_t tl = 0, th = n_row_tiles;
    while (th - tl > 1) {
        const uint32_t mid = (tl + th) >> 1;
        if (off[mid] <= bid) tl = mid; else th = mid;
    }
    const uint32_t t = tl;
    const uint32_t chunk = ((const_u32_ptr)(uintptr_t)tile_first)[t] + (bid - off[t]);
    const uint32_t t_lo = ((const_u32_ptr)(uintptr_t)tile_lo)[t];
    const uint32_t t_hi = ((const_u32_ptr)(uintptr_t)tile_hi)[t];
    const uint32_t c_begin = max(chunk * chunk_cols, t_lo);
    const uint32_t c_end = min((chunk + 1) * chunk_cols, t_hi);
    if (c_begin >= c_end) return;

    // targets -> VGPRs
    uint32_t rw[R][32];
    uint32_t rlo[R], rhi[R], rid[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const uint32_t p = t * TILE_ROWS + wave * (64 * R) + k * 64 + lane;
        uint32_t lo = row_lo[p], hi = row_hi[p];  // arrays are padded to whole tiles
        uint32_t src = p;
        if (p < n_rows) {
            if (row_perm) src = row_perm[p];
            const uint4 *rp = reinterpret_cast<const uint4 *>(row_hashes + (size_t)src * 32);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint4 v = rp[q];
                rw[k][4 * q + 0] = v.x; rw[k][4 * q + 1] = v.y; rw[k][4 * q + 2] = v.z; rw[k][4 * q + 3] = v.w;
            }
            if (self_mode && matched && ((matched[src >> 5] >> (src & 31)) & 1u)) { lo = 0; hi = 0; }
        } else {
#pragma unroll
            for (int q = 0; q < 32; q++) rw[k][q] = 0u;
            lo = 0; hi = 0;
        }
        rlo[k] = lo; rhi[k] = hi; rid[k] = row_index_base + src;
    }

    const_u32_ptr cols = (const_u32_ptr)(uintptr_t)col_hashes;
    uint32_t n_early = 0;  // candidates this wave left early (x 64 R pairs each)
    for (uint32_t c = c_begin; c < c_end; ++c) {
        const_u32_ptr cp = cols + (size_t)c * 32;
        uint32_t d[R];
#pragma unroll
        for (int k = 0; k < R; k++) d[k] = 0u;
#pragma unroll
        for (int w = 0; w < (CHKW < 32 ? CHKW : 32); ++w) {
            const uint32_t cw = cp[w];  // SGPR
#pragma unroll
            for (int k = 0; k < R; k++) d[k] += __builtin_popcount(rw[k][w] ^ cw);
        }
        uint32_t m = d[0];
#pragma unroll
        for (int k = 1; k < R; k++) m = min(m, d[k]);
        if (CHKW < 32) {
            if (__builtin_amdgcn_ballot_w64(m <= tol) == 0ull) { n_early += 1; continue; }  // partial distances only grow
#pragma unroll
            for (int w = CHKW; w < 32; ++w) {
                const uint32_t cw = cp[w];
#pragma unroll
                for (int k = 0; k < R; k++) d[k] += __builtin_popcount(rw[k][w] ^ cw);
            }
            m = d[0];
#pragma unroll
            for (int k = 1; k < R; k++) m = min(m, d[k]);
        }
        if (__builtin_amdgcn_ballot_w64(m <= tol) != 0ull) {
            // rare path: window, consumption bitmap, append
            bool col_ok = true;
            if (matched) col_ok = ((((const_u32_ptr)(uintptr_t)matched)[c >> 5] >> (c & 31)) & 1u) == 0u;
            if (col_ok) {
#pragma unroll
                for (int k = 0; k < R; k++)
                    wave_append_hits(d[k] <= tol && c >= rlo[k] && c < rhi[k], rid[k], c, hits, capacity, counters, overflow_row);
            }
        }
    }
    if (threadIdx.x == 0) atomicAdd(&counters[1], (unsigned long long)(c_end - c_begin) * TILE_ROWS);
    if (CHKW < 32 && lane == 0 && n_early) atomicAdd(&counters[3], (unsigned long long)n_early * (64u * R));
}

// ---- exact Hamming distances on the matrix cores ---------------------------------------------------------
// With every hash bit encoded as an fp4 value (e2m1: bit 0 -> 0x0 = 0.0, bit 1 -> 0x2 = +1.0) the dot product of two hashes
// is popcount(a & b), an integer <= 1024 that f32 accumulation represents exactly, and
// hamming = pop(a) + pop(b) - 2 dot  bit for bit.  v_mfma_f32_32x32x64_f8f6f4 (both operands fp4) evaluates 32 x 32 pairs
// x 64 bit positions per instruction; measured 15 ns per instruction per SIMD (tools/ubench_mfma.hip) = 4.4e12 pairs/s
// chip-wide, 7x the VALU issue ceiling of the XOR + popcount formulation (tools/ubench_valu.hip).  Layout probed with exact
// data (tools/probe_mfma_fp4.hip): A row / B col = lane & 31, k-group = lane >> 5 (32 nibbles = 16 B per lane),
// C row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), C col = lane & 31.
// (Round 1's kernel encoded the bits as +-1 - dot = 1024 - 2 hamming - and tested inside the stream; round 2's {0, 1} form
// below replaced it - 10 % less energy per MFMA at the chip's power cap, tools/ubench_mfma_energy.hip - and it was deleted
// in round 4: DESIGN.md 4.2 keeps its measurements.)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// packed [n][32] u32  ->  expanded [n_pad][32 chunks][16 B]: chunk d = the 32 nibbles of packed dword d
// (nibble q <-> bit q).  Rows n .. n_pad are zero (fp4 +0.0: dot 0, never inside a window).
// Bit 0 -> 0x0 (0.0), bit 1 -> 0x2 (+1.0): three quarters of the products are zero and nothing is negative.  The popcounts
// the threshold needs are written here: pop[h] over all 1024 bits, popk[h] over the bits the first k_steps k-steps of the
// kernel cover (k-step s multiplies packed dwords s and 16 + s: the two halves of the wave take chunks s and 16 + s),
// and popkT = popk transposed inside groups of 128 hashes ([group][h & 31][(h >> 5) & 3]) so that a lane fetches the
// values of its column in the four 32-column sub-tiles of a stage with one 16-byte load.
__global__ __launch_bounds__(256) void expand_fp4_kernel(const uint32_t *__restrict__ packed, uint32_t n,
                                                        uint32_t n_pad, uint4 *__restrict__ expanded,
                                                        uint32_t k_steps, float *__restrict__ pop,
                                                        float *__restrict__ popk, float *__restrict__ popkT)
{
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)n_pad * 32;
         idx += (size_t)gridDim.x * 256) {  // (hash, dword), grid-stride: the grid is capped under HIP's 2^32 limit
    const uint32_t hsh = (uint32_t)(idx >> 5), dw = (uint32_t)idx & 31u;
    uint4 out = {0u, 0u, 0u, 0u};
    uint32_t w = 0;
    if (hsh < n) {
        w = packed[idx];
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t x = (w >> (8 * j)) & 0xFFu;          // 8 bits -> 8 nibbles
            x = (x | (x << 12)) & 0x000F000Fu;
            x = (x | (x << 6)) & 0x03030303u;
            x = (x | (x << 3)) & 0x11111111u;
            o[j] = x << 1;
        }
        out = make_uint4(o[0], o[1], o[2], o[3]);
    }
    expanded[idx] = out;
    if (pop) {  // the 32 dwords of a hash sit in 32 consecutive lanes (n_pad is a multiple of 128: no partial waves)
        uint32_t pc = (uint32_t)__builtin_popcount(w), pk = (dw & 15u) < k_steps ? pc : 0u;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { pc += (uint32_t)__shfl_xor((int)pc, o, 32); pk += (uint32_t)__shfl_xor((int)pk, o, 32); }
        if (dw == 0) {
            pop[hsh] = (float)pc;
            popk[hsh] = (float)pk;
            popkT[(size_t)(hsh >> 7) * 128 + (hsh & 31u) * 4 + ((hsh >> 5) & 3u)] = (float)pk;
        }
    }
    }
}

// ---- second-generation MFMA search kernel ----------------------------------------------------------------------
// Two findings of round 2 shape it (profiles/r02_*, DESIGN.md "Hamming search"):
//  (1) Round 1's kernel (+-1 encoding, test inside the stream) did not run at the matrix pipe's issue rate but at the chip's POWER cap: a stream with 87 % pipe
//      utilisation and one with 81 % take the same wall time, the chip just holds a lower clock.  What decides pairs/s is
//      the energy per pair, and that depends on the operand VALUES: with hash bits encoded as {0, 1} instead of {-1, +1}
//      (dot = popcount(a & b); three quarters of the products are zero, nothing is negative) the same MFMA stream takes
//      10 % less time; the unscaled opcode (v_mfma_f32_32x32x64_f8f6f4, no E8M0 scale operands) another 1 %
//      (tools/ubench_mfma_energy.hip).  With pa, pb the popcounts:  hamming = pa + pb - 2 dot.  The row term goes in as
//      the C operand of a block's first MFMA (acc = dot - paK / 2, a persistent register vector per row tile), the
//      column term into a per-lane threshold ((pbK - tol) / 2: the C layout has one column per lane), so the test is still
//      one max over the accumulators and one compare: partial distance <= tol <=> acc >= (pbK - tol) / 2.  All values are
//      half-integers below 2^11: exact in f32.
//  (2) Once the clock is no longer the limit the pipe's idle cycles count again, so the stream is branch-free: a stage
//      (kSub sub-tiles of 32 candidates) is cut into blocks (sub-tile, row tile) of K = CHK + 1 MFMAs that run one after
//      the other, alternating between the wave's two row tiles; while block b accumulates into one accumulator set the
//      finished block b - 1 sits in the other and its test (8 v_max3 + a compare) issues in the shadow of block b's MFMAs.
//      Each B fragment is read from LDS twice (once per row tile: 1 ds_read_b128 per MFMA, half of what the LDS array
//      sustains).  The test only sets a bit in a scalar mask; flagged blocks (they may contain a pair within the
//      tolerance: ~0.3 % on unrelated hashes) are evaluated over all 1024 bits at the end of the stage - operands from
//      the LDS image and the target registers - where window / consumption bitmap / append are applied as in the VALU
//      kernel's slow path.  LDS reads run P fragments ahead in ONE stream across block boundaries; sched_barrier(0)
//      between slots makes the source order the schedule.
// CHK (< 15) is the exact early exit: a Hamming distance only grows as more bit positions are counted, so if after k-steps
// 0..CHK (64 (CHK + 1) bits) every pair of a 32 x 32 block is already MORE than `tol` apart, none can be a hit - exact for any
// data.  Unrelated hashes sit at (bits / 2) +- sqrt(bits) / 2, so at tolerance 350 the test after 832 bits (CHK = 12) passes
// for ~99.7 % of the blocks and saves 3 of 16 MFMA steps; the host picks CHK from the tolerance (16 = no test: K = 16).
template <int N>
struct IntC { static constexpr int value = N; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(IntC<I>{});
        static_for<I + 1, N>(f);
    }
}

// Second pass of the second-generation MFMA search.  A block of the stream that cannot be ruled out after K k-steps
// (~0.3 % on unrelated hashes) names its suspects: per lane (= candidate column) the 16-bit mask of accumulator registers
// (= target rows) whose partial distance is still within the tolerance - typically ONE pair of the block's 1024.  The
// stream appends them to a queue (slots come in per-wave chunks, so appending costs no round trip) and moves on; this
// kernel evaluates each suspect pair exactly - XOR + popcount over all 16 words of the PACKED hashes, as
// VideoHash::hamming_distance does (video_hash.rs:311-317) - and applies window / consumption bitmap / append like the
// reference's loop (search_algorithm.rs:150-156, :67-74).  One lane per queue entry.
struct CandEntry {
    uint32_t row_base;  // position (tile order) of the row of accumulator register 0 for this lane group; 0xFFFFFFFF = empty slot
    uint32_t col;
    uint32_t mask;      // bit r set: register r, i.e. row row_base + (r & 3) + 8 (r >> 2)
    uint32_t pad;
};

__global__ __launch_bounds__(256) void resolve_candidates_kernel(
    const CandEntry *__restrict__ cand, const unsigned long long *__restrict__ cand_head, uint32_t cand_capacity,
    const uint32_t *__restrict__ row_hashes, const uint32_t *__restrict__ row_perm, uint32_t n_rows, uint32_t row_index_base,
    const uint32_t *__restrict__ col_hashes, uint32_t n_cols, const uint32_t *__restrict__ row_lo,
    const uint32_t *__restrict__ row_hi, uint32_t tol, const uint32_t *__restrict__ matched, int self_mode,
    vdf_hit *__restrict__ hits, unsigned long long capacity, unsigned long long *__restrict__ counters,
    uint32_t *__restrict__ overflow_row)
{
    const uint32_t n = (uint32_t)min(*cand_head, (unsigned long long)cand_capacity);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Hits are staged per wave in LDS and appended kStage at a time: ONE atomic on the shared counter and a coalesced copy
    // per ~500 hits.  With dense near-duplicates (1e7 hits per search) even one atomic per wave-round on the one address
    // is what the kernel waits for (measured: 3.6 ms for 6.6 M hits, 0.05 ms for 4 k).  The loops are wave-uniform.
    constexpr uint32_t kStage = 512;
    __shared__ vdf_hit s_stage[4][kStage];
    uint32_t staged = 0;  // wave-uniform
    auto flush = [&]() {
        if (staged == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave's own LDS writes, read back by other lanes
        __builtin_amdgcn_wave_barrier();
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&counters[0], (unsigned long long)staged);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        for (uint32_t k = lane; k < staged; k += 64) {
            const vdf_hit hp = s_stage[wave][k];
            if (base + k < capacity) hits[base + k] = hp;
            else atomicMin(overflow_row, hp.row);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        staged = 0;
    };
    for (uint32_t e0 = blockIdx.x * 256 + (threadIdx.x & ~63u); e0 < n; e0 += gridDim.x * 256) {
        const uint32_t e = e0 + lane;
        CandEntry ce;
        ce.row_base = 0xFFFFFFFFu; ce.col = 0; ce.mask = 0; ce.pad = 0;
        if (e < n) ce = cand[e];
        uint32_t m = ce.mask;
        if (ce.row_base == 0xFFFFFFFFu || ce.col >= n_cols) m = 0;
        if (m && matched && ((matched[ce.col >> 5] >> (ce.col & 31)) & 1u)) m = 0;
        uint4 cw[8];
        if (m) {
            const uint4 *cp = reinterpret_cast<const uint4 *>(col_hashes + (size_t)ce.col * 32);
#pragma unroll
            for (int q = 0; q < 8; q++) cw[q] = cp[q];
        }
        while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
            bool hit = false;
            uint32_t src = 0;
            if (m) {
                const uint32_t r = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                const uint32_t p = ce.row_base + (r & 3) + 8 * (r >> 2);
                if (p < n_rows) {
                    const uint32_t lo = row_lo[p], hi = row_hi[p];
                    if (ce.col >= lo && ce.col < hi) {
                        src = row_perm ? row_perm[p] : p;
                        if (!(self_mode && matched && ((matched[src >> 5] >> (src & 31)) & 1u))) {
                            const uint4 *rp = reinterpret_cast<const uint4 *>(row_hashes + (size_t)src * 32);
                            uint32_t d = 0;
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const uint4 v = rp[q];
                                d += __builtin_popcount(v.x ^ cw[q].x) + __builtin_popcount(v.y ^ cw[q].y) +
                                     __builtin_popcount(v.z ^ cw[q].z) + __builtin_popcount(v.w ^ cw[q].w);
                            }
                            hit = d <= tol;
                        }
                    }
                }
            }
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            if (bal == 0ull) continue;
            const uint32_t cnt = (uint32_t)__builtin_popcountll(bal);
            if (staged + cnt > kStage) flush();
            if (hit) {
                vdf_hit hp; hp.row = row_index_base + src; hp.col = ce.col;
                s_stage[wave][staged + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = hp;
            }
            staged += cnt;
        }
    }
    flush();
}

template <int CHK, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void hamming_mfma2_kernel(
    const uint4 *__restrict__ row_exp, const uint32_t *__restrict__ row_perm, uint32_t n_rows,
    uint32_t row_index_base, const uint4 *__restrict__ col_exp, const uint32_t *__restrict__ row_lo,
    const uint32_t *__restrict__ row_hi, const uint32_t *__restrict__ tile_lo, const uint32_t *__restrict__ tile_hi,
    const uint32_t *__restrict__ tile_first, const uint32_t *__restrict__ tile_count,
    const uint32_t *__restrict__ group_offset, const uint32_t *__restrict__ group_cmin, uint32_t n_groups,
    uint32_t group_size, uint32_t shard_index, uint32_t shard_count, uint32_t n_row_tiles, uint32_t chunk_cols,
    uint32_t tol, const uint32_t *__restrict__ matched, int self_mode, vdf_hit *__restrict__ hits,
    unsigned long long capacity, unsigned long long *__restrict__ counters, uint32_t *__restrict__ overflow_row,
    uint32_t block_base, const float *__restrict__ row_pop3, uint32_t row_pad, const float *__restrict__ col_pop3,
    uint32_t col_pad, CandEntry *__restrict__ cand, uint32_t cand_capacity, unsigned long long *__restrict__ cand_head)
{
    constexpr int K = CHK < 15 ? CHK + 1 : 16;            // k-steps of the main stream (64 bit positions each)
    // The row-term vectors (the C operand of a block's first MFMA: -paK / 2 of the block's 32 rows) take 32 registers if they are kept.
    // Up to K = 13 (tolerance <= 0.357; 104 registers of target fragments) they are; the longer streams of larger tolerances (K = 14,
    // 15, 16: 112 - 128 registers of targets, which spilled 6 - 9 VGPRs to scratch inside the stream) fetch them from LDS instead - a
    // vector depends on (row tile, lane group) only: 256 B per wave - straight into the accumulator set of the NEXT block, in the four
    // slots behind the test that last read it.
    constexpr bool kRowcLds = K > 13;
    constexpr uint32_t kSub = WAVES >= 8 ? 4 : 2;         // 32-candidate sub-tiles per LDS stage
    constexpr uint32_t kColStep = 32 * kSub;
    constexpr int NBLK = 2 * (int)kSub;                   // blocks per stage and wave: (sub-tile, row tile)
    constexpr int NM = NBLK * K;                          // MFMAs per stage and wave
    constexpr int P = (K == 16 && WAVES >= 8) ? 2 : 4, NB = P + 1;  // LDS fragments in flight / fragment buffers (the full-length stream of the
                                                                           // 512-row workgroup - 128 registers of targets, four thresholds - has room for two)
    constexpr uint32_t kTileRows = 64 * WAVES;
    constexpr int kDmaPerWave = (int)(kColStep * 32 / (64 * WAVES));  // 1 KB LDS-DMA pieces per wave and stage
    constexpr int kDmaEvery = NM / kDmaPerWave;  // slots between DMA pieces: spread evenly over the stage
    static_assert(kDmaEvery >= 2, "stage too short for its DMA pieces");
    constexpr int kTestAt = 2;                            // the previous block's test issues under MFMAs kTestAt .. kTestAt + 3 of a block
    static_assert(K >= kTestAt + 4, "block too short to hide the previous block's test");
    const uint32_t bid = blockIdx.x + block_base;
    __shared__ __attribute__((aligned(16))) uint4 s_b0[kColStep * 32];
    __shared__ __attribute__((aligned(16))) uint4 s_b1[kColStep * 32];
    const uint32_t tid = threadIdx.x, lane = tid & 63, g = lane >> 5, c31 = lane & 31;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // scalar: LDS-DMA targets (M0) need no VALU

    // workgroup -> (group, chunk, row tile in group): chunk-major inside a group; this rank's tiles of the group are
    // t0, t0 + shard_count, ... (t % shard_count == shard_index)
    const_u32_ptr goff = (const_u32_ptr)(uintptr_t)group_offset;
    uint32_t gl = 0, gh = n_groups;
    while (gh - gl > 1) {
        const uint32_t mid = (gl + gh) >> 1;
        if (goff[mid] <= bid) gl = mid; else gh = mid;
    }
    const uint32_t idx = bid - goff[gl];
    const uint32_t per_group = (group_size + shard_count - 1) / shard_count;
    const uint32_t g0 = gl * group_size;
    const uint32_t t0 = g0 + (shard_index + shard_count - g0 % shard_count) % shard_count;
    const uint32_t t = t0 + (idx % per_group) * shard_count;
    const uint32_t chunk = ((const_u32_ptr)(uintptr_t)group_cmin)[gl] + idx / per_group;
    if (t >= n_row_tiles || t >= g0 + group_size) return;
    {
        const uint32_t f = ((const_u32_ptr)(uintptr_t)tile_first)[t], cnt = ((const_u32_ptr)(uintptr_t)tile_count)[t];
        if (chunk < f || chunk >= f + cnt) return;
    }
    const uint32_t t_lo = ((const_u32_ptr)(uintptr_t)tile_lo)[t];
    const uint32_t t_hi = ((const_u32_ptr)(uintptr_t)tile_hi)[t];
    const uint32_t c_begin = max(chunk * chunk_cols, t_lo);
    const uint32_t c_end = min((chunk + 1) * chunk_cols, t_hi);
    if (c_begin >= c_end) return;

    // targets: 2 row tiles of 32 per wave, the fragments of the K streamed k-steps in registers; their C-operand
    // vectors -paK / 2 (C layout: register r of lane group g <-> row (r & 3) + 8 (r >> 2) + 4 g)
    const float tol_f = (float)min(tol, 1024u);
    const float *row_popk = row_pop3 + row_pad;
    const float *col_popkT = col_pop3 + 2 * (size_t)col_pad;
    const uint32_t row0 = t * kTileRows + wave * 64;
    v4i a[2][K];
    const float4 *s_rc = nullptr;  // kRowcLds: [wave][row tile][lane group][4 x float4]
    v16f rowc[2];
    uint32_t live[2];  // bit r: the row of accumulator register r has a non-empty window in this launch (suspects of other rows are noise)
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
        live[rt] = 0;
        const uint32_t p = row0 + 32 * rt + c31;
        uint32_t src = p;
        if (p < n_rows && row_perm) src = row_perm[p];
        const uint4 *rp = row_exp + (size_t)src * 32 + 16 * g;  // rows >= n_rows read the zero padding
#pragma unroll
        for (int s = 0; s < K; s++) {
            const uint4 v = rp[s];
            a[rt][s] = (v4i){(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t pr = row0 + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * g;
            const uint32_t sr = (pr < n_rows && row_perm) ? row_perm[pr] : pr;
            rowc[rt][r] = -0.5f * row_popk[sr];
            if (row_hi[pr] > row_lo[pr]) live[rt] |= 1u << r;  // the window arrays are padded to whole tiles
        }
    }
    if constexpr (kRowcLds) {  // the vectors go to LDS (one lane of each lane group writes; read back by this wave only, behind the barrier
                               // in front of the stream) and their 32 registers are free for the longer stream's target fragments
        __shared__ __attribute__((aligned(16))) float4 s_rowc[WAVES * 2 * 2 * 4];
        float4 *mine = s_rowc + ((size_t)wave * 2 * 2 + g) * 4;
        if (c31 == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    mine[rt * 2 * 4 + q] = float4{rowc[rt][4 * q], rowc[rt][4 * q + 1], rowc[rt][4 * q + 2], rowc[rt][4 * q + 3]};
        }
        s_rc = mine;
    }
    // Stage loader: a stage is kColStep columns of 32 chunks of 16 B; LDS slot L = (col << 5 | q) holds chunk q ^ col of that
    // column (XOR swizzle: the 16-lane groups of ds_read_b128 then hit 16 different 4-bank groups).  Staged by LDS-DMA
    // (buffer_load ... lds, MUBUF: after a FLAT-encoded LDS-DMA the compiler drains every outstanding ds_read with lgkmcnt(0)):
    // the DMA writes LDS linearly (wave base + lane * 16), the swizzle goes on the per-lane SOURCE offset; no VGPRs are spent
    // and nothing waits until the barrier that publishes the stage.  Piece i of this wave fills slots 64 W i + 64 wave + lane
    // (W waves): column c = 2 W i + c0 with c0 = 2 wave + g < 2 W, chunk q = lane & 31, source byte offset c * 512 +
    // ((q ^ c) << 4); since c0 < 2 W, q ^ c = (q ^ c0) ^ (2 W i & 31): 16 / W per-lane offsets cover all pieces
    const uint32_t cb0 = c_begin & ~(kColStep - 1);
    constexpr uint32_t kColsPerRound = 2 * WAVES;
    constexpr int kOffsets = 32 / kColsPerRound;
    static_assert(kOffsets >= 1 && (kOffsets & (kOffsets - 1)) == 0, "piece addressing");
    const uint32_t c0 = 2 * wave + g;
    uint32_t lane_off[kOffsets];
#pragma unroll
    for (int j = 0; j < kOffsets; j++) lane_off[j] = c0 * 512u + ((((lane & 31) ^ c0) << 4) ^ (16u * kColsPerRound * j));
    auto stage_rsrc = [&](uint32_t cb) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(col_exp) + (size_t)cb * 512), 0, kColStep * 512u, 0x00020000);
    };
    auto load_piece = [&](__amdgpu_buffer_rsrc_t rs, uint4 *dst, int i) __attribute__((always_inline)) {
        auto *lds = (__attribute__((address_space(3))) void *)&dst[64 * WAVES * i + 64 * wave];
        const int voff = (int)lane_off[i & (kOffsets - 1)], soff = (int)(512u * kColsPerRound * (uint32_t)i);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds, 16, voff, soff, 0, 0);
    };
    // per-lane thresholds (pbK - tol) / 2 of the lane's column in the kSub sub-tiles of the stage starting at cb
    auto load_thr = [&](uint32_t cb, float (&out)[kSub]) __attribute__((always_inline)) {
        const float *q = col_popkT + (size_t)(cb >> 7) * 128 + c31 * 4 + ((cb >> 5) & 3u);
        if constexpr (kSub == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(q);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(q);
            out[0] = v.x; out[1] = v.y;
        }
    };

    v16f acc[2] = {v16f{}, v16f{}};
    if constexpr (kRowcLds) acc[0] = rowc[0];  // the stream's first block starts from its row term (the later ones fetch theirs from LDS)
    uint32_t n_early = 0;  // blocks that stopped after K steps
    float thr[kSub], thr_next[kSub];

    // A flagged block names its suspect pairs (rare path, ~0.3 % of the blocks): per lane the mask of accumulator registers
    // still within the tolerance; lanes with a non-empty mask and a column inside this workgroup's range append one entry
    // to the candidate queue.  Slots come in chunks of kCandChunk per wave (one returning atomic per chunk, not per entry; a wave sees ~12 suspects in
    // its 512 x 65536 share of unrelated hashes, so small chunks strand little),
    // the entries are fire-and-forget stores; resolve_candidates_kernel evaluates them exactly.
    constexpr uint32_t kCandChunk = 8, kCandChunkMax = 512;
    uint32_t q_chunk = kCandChunk, q_taken = 0;  // chunk size (grows, see emit) and chunks taken so far
    unsigned long long q_next = 0, q_end = 0;  // this wave's slots [q_next, q_end); 64 bit: the head keeps counting past a full queue
    auto emit = [&](const v16f &c, float thrv, uint32_t live_rows, uint32_t rows_first, uint32_t col_first) __attribute__((always_inline)) {
        uint32_t mask = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) mask |= (c[r] >= thrv) ? (1u << r) : 0u;
        mask &= live_rows;
        const uint32_t j = col_first + c31;
        if (j < c_begin || j >= c_end) mask = 0;  // the neighbouring chunk's workgroup owns those columns
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(mask != 0u);
        if (bal == 0ull) return;
        const uint32_t need = (uint32_t)__builtin_popcountll(bal);
        if (q_next + need > q_end) {  // new chunk (the rest of the old one stays empty: slots are pre-filled with 0xFF)
            // A wave that keeps coming back for slots is inside near-duplicate territory (unrelated hashes give a wave ~12
            // suspects in its whole 512 x 65536 share: two chunks), and there one returning atomic on the one head counter
            // per chunk is what the whole search then waits for (6.6 M hits: stream 4.1 -> 7.3 ms with fixed chunks of 8 -
            // 800 k same-address atomics).  From the third chunk on the chunks double, up to kCandChunkMax.
            if (q_taken >= 2) q_chunk = min(2u * q_chunk, kCandChunkMax);
            q_taken++;
            const uint32_t take = max((need + kCandChunk - 1) / kCandChunk * kCandChunk, q_chunk);
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(cand_head, (unsigned long long)take);
            q_next = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
            q_end = q_next + take;
        }
        const unsigned long long slot = q_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (mask != 0u) {
            if (slot < cand_capacity) {
                CandEntry ce; ce.row_base = rows_first + 4 * g; ce.col = j; ce.mask = mask; ce.pad = 0;
                *reinterpret_cast<uint4 *>(&cand[slot]) = *reinterpret_cast<const uint4 *>(&ce);
            } else {  // queue full: the caller's overflow protocol restarts from the smallest row that lost a suspect
                const uint32_t p = rows_first + 4 * g + (uint32_t)(__builtin_ctz(mask) & 3) + 8 * (uint32_t)(__builtin_ctz(mask) >> 2);
                if (p < n_rows) atomicMin(overflow_row, row_index_base + (row_perm ? row_perm[p] : p));
            }
        }
        q_next += need;
    };

    // One stage: NBLK blocks of K MFMAs over the candidates in `cur`; DMA of the next stage into `nxt`.
    auto run_stage = [&](uint32_t cb, const uint4 *cur, uint4 *nxt) __attribute__((always_inline)) {
        const uint32_t cb_next = cb + kColStep < c_end ? cb + kColStep : cb;
        const __amdgpu_buffer_rsrc_t rs_next = stage_rsrc(cb_next);
        load_thr(cb_next, thr_next);  // lands long before the barrier that ends this stage
        uint32_t flags = 0;
        uint4 fq[NB];
        auto frag = [&](int i) __attribute__((always_inline)) {  // fragment of stream position i: block i / K, step i % K
            const uint32_t sub = (uint32_t)((i / K) >> 1), s = (uint32_t)(i % K);
            return cur[((32u * sub + c31) << 5) | ((s + 16u * g) ^ c31)];
        };
#pragma unroll
        for (int i = 0; i < P; i++) fq[i] = frag(i);
        __builtin_amdgcn_sched_barrier(0);
        // The source order below IS the schedule: one slot = one MFMA + the LDS read P fragments ahead + at most one DMA
        // piece + a quarter of the previous block's test; sched_barrier(0) between slots keeps the machine scheduler from
        // moving anything across (left alone it sinks every read next to its MFMA and bunches the VALU work).
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f;
        static_for<0, NM>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int blk = i / K, s = i % K, rt = blk & 1;
            const uint4 bv = fq[i % NB];
            const v8i b = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w, 0, 0, 0, 0};
            const v8i ar = {a[rt][s].x, a[rt][s].y, a[rt][s].z, a[rt][s].w, 0, 0, 0, 0};
            // scale operands 0 / 0: the compiler selects the unscaled opcode (same values as scales 2^0, probed)
            if constexpr (s == 0 && !kRowcLds) acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ar, b, rowc[rt], 4, 4, 0, 0, 0, 0);
            else acc[rt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ar, b, acc[rt], 4, 4, 0, 0, 0, 0);  // (kRowcLds: acc holds the row term at s == 0)
            if constexpr (i + P < NM) fq[(i + P) % NB] = frag(i + P);
            if constexpr ((i % kDmaEvery) == 1 && (i / kDmaEvery) < kDmaPerWave) load_piece(rs_next, nxt, i / kDmaEvery);
            // the block that finished kTestAt MFMAs ago sits in the other accumulator set: max over its 16 registers as a
            // tree, two or three v_max3 per slot
            if constexpr (blk >= 1) {
                const v16f &c = acc[rt ^ 1];
                if constexpr (s == kTestAt) {
                    m0 = fmaxf(fmaxf(c[0], c[1]), c[2]); m1 = fmaxf(fmaxf(c[3], c[4]), c[5]); m2 = fmaxf(fmaxf(c[6], c[7]), c[8]);
                } else if constexpr (s == kTestAt + 1) {
                    m3 = fmaxf(fmaxf(c[9], c[10]), c[11]); m4 = fmaxf(fmaxf(c[12], c[13]), c[14]);
                } else if constexpr (s == kTestAt + 2) {
                    m5 = fmaxf(fmaxf(m0, m1), m2); m3 = fmaxf(fmaxf(m3, m4), c[15]);
                } else if constexpr (s == kTestAt + 3) {
                    if (__builtin_amdgcn_ballot_w64(fmaxf(m5, m3) >= thr[(blk - 1) >> 1]) != 0ull) {
                        flags |= 1u << (blk - 1);
                        emit(c, thr[(blk - 1) >> 1], live[rt ^ 1], row0 + 32u * (uint32_t)(rt ^ 1), cb + 32u * (uint32_t)((blk - 1) >> 1));
                    }
                }
            }
            // kRowcLds: the other accumulator set is free once its test (slots kTestAt .. kTestAt + 3) is through; the next block - which
            // accumulates into it - starts from the row term, a quarter of the vector per slot
            if constexpr (kRowcLds && s >= kTestAt + 4 && s < kTestAt + 8) {
                constexpr int q = s - (kTestAt + 4);
                const float4 v = s_rc[((rt ^ 1) * 2) * 4 + q];
                acc[rt ^ 1][4 * q + 0] = v.x; acc[rt ^ 1][4 * q + 1] = v.y; acc[rt ^ 1][4 * q + 2] = v.z; acc[rt ^ 1][4 * q + 3] = v.w;
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        {   // the last block of the stage has nothing of its own stage to hide behind
            const v16f &c = acc[(NBLK - 1) & 1];
            m0 = fmaxf(fmaxf(c[0], c[1]), c[2]); m1 = fmaxf(fmaxf(c[3], c[4]), c[5]); m2 = fmaxf(fmaxf(c[6], c[7]), c[8]);
            m3 = fmaxf(fmaxf(c[9], c[10]), c[11]); m4 = fmaxf(fmaxf(c[12], c[13]), c[14]);
            const float pm = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), c[15]));
            if (__builtin_amdgcn_ballot_w64(pm >= thr[kSub - 1]) != 0ull) {
                flags |= 1u << (NBLK - 1);
                emit(c, thr[kSub - 1], live[(NBLK - 1) & 1], row0 + 32u * (uint32_t)((NBLK - 1) & 1), cb + 32u * (kSub - 1));
            }
        }
        // blocks of sub-tiles at or beyond the end of the chunk hold nothing
        uint32_t valid = 0;
#pragma unroll
        for (uint32_t sub = 0; sub < kSub; sub++)
            if (cb + 32u * sub < c_end) valid |= 3u << (2 * sub);
        flags &= valid;
        n_early += (uint32_t)__builtin_popcount(valid & ~flags);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // explicit: the fence's vmcnt wait is the compiler's to drop (it did, in a resize kernel: DESIGN 4.3)
        __syncthreads();  // the DMA and the threshold load have landed (every wave waited for its own) and every wave is done with `cur`
#pragma unroll
        for (uint32_t sub = 0; sub < kSub; sub++) thr[sub] = 0.5f * (thr_next[sub] - tol_f);
    };

#pragma unroll
    for (int i = 0; i < kDmaPerWave; i++) load_piece(stage_rsrc(cb0), s_b0, i);
    load_thr(cb0, thr_next);
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int s = 0; s < K; s++) asm volatile("" ::"v"(a[rt][s]));  // retire the target loads before the loop (vmcnt bookkeeping)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // explicit: the fence's vmcnt wait is the compiler's to drop (it did, in a resize kernel: DESIGN 4.3)
    __syncthreads();
#pragma unroll
    for (uint32_t sub = 0; sub < kSub; sub++) thr[sub] = 0.5f * (thr_next[sub] - tol_f);
    for (uint32_t cb = cb0; cb < c_end; cb += 2 * kColStep) {
        run_stage(cb, s_b0, s_b1);
        if (cb + kColStep >= c_end) break;
        run_stage(cb + kColStep, s_b1, s_b0);
    }
    if (tid == 0) atomicAdd(&counters[1], (unsigned long long)(c_end - c_begin) * kTileRows);
    if (CHK < 15 && lane == 0 && n_early) atomicAdd(&counters[3], (unsigned long long)n_early * (32u * 32u));
}

// ---- hits that cannot matter to the greedy replay (search_algorithm.rs:131-170) -------------------------------------
// The replay walks the targets in ascending order; a target's hit list is looked at only if the target is still unmatched
// when its turn comes.  Call k a ROOT if no hit (x, k) exists: nothing can consume a root before its turn, so it IS a target,
// and every candidate i of one of its hits (k, i) is matched once k's turn is over (by k, or earlier by someone else).  i > k,
// so i is never a target: its own hit list is dead weight.  In a cluster of s mutual near-duplicates that leaves the
// smallest member's s - 1 hits of the s (s - 1) / 2 - what the reference's consuming loop looks at, too.  Needs the COMPLETE
// hit set of the launch (no buffer overflow, one shard): the caller checks that.  Bitmaps are 1 bit per entry, zeroed.
__global__ __launch_bounds__(256) void hits_mark_incoming_kernel(const vdf_hit *__restrict__ hits, unsigned long long n,
                                                                 uint32_t *__restrict__ has_in)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t j = hits[i].col;
        // bits only ever go 0 -> 1: a plain look first skips the atomic for the ~98 % of hits whose column is already marked
        // (measured 0.59 ms for 6.6 M hits with unconditional atomics: thousands of them per word); a stale 0 only costs the atomic
        if (!((has_in[j >> 5] >> (j & 31)) & 1u)) atomicOr(&has_in[j >> 5], 1u << (j & 31));
    }
}

__global__ __launch_bounds__(256) void hits_mark_covered_kernel(const vdf_hit *__restrict__ hits, unsigned long long n,
                                                                const uint32_t *__restrict__ has_in, uint32_t *__restrict__ covered)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const vdf_hit h = hits[i];
        if (!((has_in[h.row >> 5] >> (h.row & 31)) & 1u) && !((covered[h.col >> 5] >> (h.col & 31)) & 1u))
            atomicOr(&covered[h.col >> 5], 1u << (h.col & 31));
    }
}

__global__ __launch_bounds__(256) void hits_compact_kernel(const vdf_hit *__restrict__ hits, unsigned long long n,
                                                           const uint32_t *__restrict__ covered, vdf_hit *__restrict__ out,
                                                           unsigned long long *__restrict__ counter)
{
    const size_t n_round = (n + 63) & ~(size_t)63;  // whole waves reach the ballot
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_round; i += (size_t)gridDim.x * 256) {
        vdf_hit h{0u, 0u};
        bool keep = false;
        if (i < n) {
            h = hits[i];
            keep = !((covered[h.row >> 5] >> (h.row & 31)) & 1u);
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
        if (bal == 0ull) continue;
        const int leader = __builtin_ctzll(bal);
        unsigned long long base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, (unsigned long long)__builtin_popcountll(bal));
        base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), leader, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)base, leader, 64);
        if (keep) out[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = h;
    }
}

// dst[w] |= srcs[0][w] | srcs[1][w] | ... (n_srcs bitmaps of n_words words each, laid out back to back): the local half of the
// cross-shard OR of the filter's bitmaps, after an all-gather put every shard's copy into `srcs`
__global__ __launch_bounds__(256) void bitmap_or_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ srcs, size_t n_words,
                                                        uint32_t n_srcs)
{
    for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < n_words; w += (size_t)gridDim.x * 256) {
        uint32_t v = dst[w];
        for (uint32_t r = 0; r < n_srcs; r++) v |= srcs[(size_t)r * n_words + w];
        dst[w] = v;
    }
}

hipError_t launch_bitmap_or(uint32_t *dst, const uint32_t *srcs, size_t n_words, uint32_t n_srcs, hipStream_t stream)
{
    if (n_words == 0 || n_srcs == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<size_t>((n_words + 255) / 256, 4096);
    hipLaunchKernelGGL(bitmap_or_kernel, dim3(grid), dim3(256), 0, stream, dst, srcs, n_words, n_srcs);
    return hipGetLastError();
}

// The filter in three steps, so that a SHARDED launch (every shard holds the hits of its own row tiles) can OR the bitmaps of
// all shards between them: has_in and covered are properties of the COMPLETE hit set.
//   1. filter_mark_incoming: zero both bitmaps (2 x ceil(n_entries / 32) words: has_in | covered), mark has_in from this shard's hits
//      -> OR has_in over the shards
//   2. filter_mark_covered: covered[col] for every hit whose row is a root          -> OR covered over the shards
//   3. filter_compact: keep the hits whose row is not covered; *counter (device) receives the survivors' count
static inline uint32_t filter_grid(unsigned long long n_hits) { return (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>((n_hits + 255) / 256, 1ull), 8192ull); }

hipError_t launch_filter_mark_incoming(const vdf_hit *hits, unsigned long long n_hits, uint32_t n_entries, uint32_t *bitmaps, hipStream_t stream)
{
    const size_t words = ((size_t)n_entries + 31) / 32;
    hipError_t e = hipMemsetAsync(bitmaps, 0, 2 * words * 4, stream);
    if (e != hipSuccess) return e;
    if (n_hits) hipLaunchKernelGGL(hits_mark_incoming_kernel, dim3(filter_grid(n_hits)), dim3(256), 0, stream, hits, n_hits, bitmaps);
    return hipGetLastError();
}

hipError_t launch_filter_mark_covered(const vdf_hit *hits, unsigned long long n_hits, uint32_t n_entries, uint32_t *bitmaps, hipStream_t stream)
{
    const size_t words = ((size_t)n_entries + 31) / 32;
    if (n_hits) hipLaunchKernelGGL(hits_mark_covered_kernel, dim3(filter_grid(n_hits)), dim3(256), 0, stream, hits, n_hits, bitmaps, bitmaps + words);
    return hipGetLastError();
}

hipError_t launch_filter_compact(const vdf_hit *hits, unsigned long long n_hits, uint32_t n_entries, const uint32_t *bitmaps, vdf_hit *out,
                                 unsigned long long *counter, hipStream_t stream)
{
    const size_t words = ((size_t)n_entries + 31) / 32;
    hipError_t e = hipMemsetAsync(counter, 0, 8, stream);
    if (e != hipSuccess) return e;
    if (n_hits) hipLaunchKernelGGL(hits_compact_kernel, dim3(filter_grid(n_hits)), dim3(256), 0, stream, hits, n_hits, bitmaps + words, out, counter);
    return hipGetLastError();
}

hipError_t launch_windows_tiles(int mode, const uint32_t *col_dur, uint32_t n_cols, const uint32_t *row_dur,
                                const uint32_t *row_perm, uint32_t n_rows, uint32_t row_begin, uint32_t row_end,
                                uint32_t shard_index, uint32_t shard_count, const SearchLaunch &L, hipStream_t stream)
{
    if (L.n_row_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(check_sorted_kernel, dim3(std::min<uint32_t>((n_cols + 255) / 256, 2048u)), dim3(256), 0, stream, col_dur,
                       n_cols, L.counters);
    hipLaunchKernelGGL(windows_tiles_kernel, dim3(L.n_row_tiles), dim3(256), 0, stream, mode, col_dur, n_cols, row_dur,
                       row_perm, n_rows, row_begin, row_end, shard_index, shard_count, (uint32_t)L.tile_rows,
                       L.chunk_cols, L.row_lo, L.row_hi, L.tile_lo, L.tile_hi, L.tile_first, L.tile_count, L.counters);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, stream, L.tile_count, L.n_row_tiles, L.tile_offset);
    if (L.n_groups) {
        hipLaunchKernelGGL(group_tiles_kernel, dim3(L.n_groups), dim3(256), 0, stream, L.tile_first, L.tile_count,
                           L.n_row_tiles, L.group_size, L.shard_count, L.group_cmin, L.group_blocks);
        hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(64), 0, stream, L.group_blocks, L.n_groups, L.group_offset);
    }
    return hipGetLastError();
}

hipError_t launch_hamming_tiles(const SearchLaunch &L, uint32_t total_tiles, hipStream_t stream)
{
    if (total_tiles == 0) return hipSuccess;
#define VDF_LAUNCH(RR, CW)                                                                                          \
    hipLaunchKernelGGL((hamming_tile_kernel<RR, CW>), dim3(nb), dim3(256), 0, stream, L.row_hashes, L.row_perm,      \
                       L.n_rows, L.row_index_base, L.col_hashes, L.row_lo, L.row_hi, L.tile_lo, L.tile_hi,          \
                       L.tile_first, L.tile_offset, L.n_row_tiles, L.chunk_cols, L.tol, L.matched, L.self_mode,     \
                       L.hits, L.capacity, L.counters, L.overflow_row, base)
#define VDF_LAUNCH_R(RR)                                                                                            \
    switch (L.prune_step) { /* instantiated: after 14, 22, 26 dwords = k-steps 6, 10, 12 of the MFMA kernel */      \
    case 6: VDF_LAUNCH(RR, 14); break;                                                                              \
    case 8: case 10: VDF_LAUNCH(RR, 22); break;                                                                     \
    case 11: case 12: VDF_LAUNCH(RR, 26); break;                                                                    \
    default: VDF_LAUNCH(RR, 32); break;                                                                             \
    }
    // HIP limits a grid to 2^32 work-items in x: slices of at most kMaxBlocksPerLaunch workgroups
    for (uint32_t base = 0; base < total_tiles; base += kMaxBlocksPerLaunch) {
        const uint32_t nb = std::min(kMaxBlocksPerLaunch, total_tiles - base);
        switch (L.tile_rows / 256) {
        case 1: VDF_LAUNCH_R(1); break;
        case 2: VDF_LAUNCH_R(2); break;
        case 4: VDF_LAUNCH_R(4); break;
        default: return hipErrorInvalidValue;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
#undef VDF_LAUNCH_R
#undef VDF_LAUNCH
    return hipSuccess;
}

// ---- the app's Sorting::Distance key ------------------------------------------------------
// vid_dup_finder_app/src/app/search_output.rs:43-60: for every MatchGroup, the maximum hamming_distance over all
// pairs of its contained_paths (duplicates, then the reference if any).  One workgroup per group; member i is
// wave-uniform (scalar loads), lanes stride over j > i.  members index `hashes`; a group with a reference takes
// its extra member from ref_hashes[ref_index].
__global__ __launch_bounds__(256) void group_max_distance_kernel(const uint32_t *__restrict__ hashes,
                                                                 const unsigned long long *__restrict__ offsets,
                                                                 const unsigned long long *__restrict__ members,
                                                                 const uint32_t *__restrict__ ref_hashes,
                                                                 const long long *__restrict__ ref_index,
                                                                 uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_max[4];
    const uint32_t g = blockIdx.x;
    const unsigned long long a = offsets[g], b = offsets[g + 1];
    const bool has_ref = ref_hashes != nullptr && ref_index != nullptr && ref_index[g] >= 0;
    const uint32_t k = (uint32_t)(b - a) + (has_ref ? 1u : 0u);
    auto member_ptr = [&](uint32_t idx) -> const uint32_t * {
        return idx < (uint32_t)(b - a) ? hashes + (size_t)members[a + idx] * 32 : ref_hashes + (size_t)ref_index[g] * 32;
    };
    uint32_t best = 0;
    for (uint32_t i = 0; i + 1 < k; i++) {
        const_u32_ptr hi = (const_u32_ptr)(uintptr_t)member_ptr(i);  // wave-uniform
        for (uint32_t j = i + 1 + threadIdx.x; j < k; j += 256) {
            const uint4 *hj = reinterpret_cast<const uint4 *>(member_ptr(j));
            uint32_t d = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint4 v = hj[q];
                d += __builtin_popcount(v.x ^ hi[4 * q]) + __builtin_popcount(v.y ^ hi[4 * q + 1]) +
                     __builtin_popcount(v.z ^ hi[4 * q + 2]) + __builtin_popcount(v.w ^ hi[4 * q + 3]);
            }
            best = max(best, d);
        }
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) out[g] = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
}

hipError_t launch_group_max_distance(const uint32_t *hashes, const unsigned long long *offsets,
                                     const unsigned long long *members, const uint32_t *ref_hashes,
                                     const long long *ref_index, uint32_t n_groups, uint32_t *out, hipStream_t stream)
{
    if (n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(group_max_distance_kernel, dim3(n_groups), dim3(256), 0, stream, hashes, offsets, members,
                       ref_hashes, ref_index, out);
    return hipGetLastError();
}

hipError_t launch_expand_fp4(const uint32_t *packed, uint32_t n, uint32_t n_pad, void *expanded, uint32_t k_steps,
                             float *pop3 /* nullable: [3][n_pad] = pop, popk, popkT */, hipStream_t stream)
{
    if (n_pad == 0) return hipSuccess;
    if (pop3 && (n_pad % 128u)) return hipErrorInvalidValue;
    const size_t total = (size_t)n_pad * 32;
    hipLaunchKernelGGL(expand_fp4_kernel, dim3((uint32_t)std::min<size_t>((total + 255) / 256, kMaxBlocksPerLaunch)), dim3(256), 0, stream, packed, n, n_pad,
                       reinterpret_cast<uint4 *>(expanded), k_steps, pop3, pop3 ? pop3 + n_pad : nullptr,
                       pop3 ? pop3 + 2 * (size_t)n_pad : nullptr);
    return hipGetLastError();
}

hipError_t launch_hamming_tiles_mfma2(const SearchLaunch &L, uint32_t total_tiles, hipStream_t stream)
{
    if (total_tiles == 0) return hipSuccess;
    if (L.tile_rows != 512 && L.tile_rows != 256) return hipErrorInvalidValue;
    for (uint32_t base = 0; base < total_tiles; base += kMaxBlocksPerLaunch) {
        const uint32_t nb = std::min(kMaxBlocksPerLaunch, total_tiles - base);
#define VDF_MFMA2_LAUNCH(CK, WV)                                                                                    \
    hipLaunchKernelGGL((hamming_mfma2_kernel<CK, WV>), dim3(nb), dim3(64 * WV), 0, stream,                           \
                       reinterpret_cast<const uint4 *>(L.row_exp), L.row_perm, L.n_rows, L.row_index_base,           \
                       reinterpret_cast<const uint4 *>(L.col_exp), L.row_lo, L.row_hi, L.tile_lo, L.tile_hi,         \
                       L.tile_first, L.tile_count, L.group_offset, L.group_cmin, L.n_groups, L.group_size,           \
                       L.shard_index, L.shard_count, L.n_row_tiles, L.chunk_cols, L.tol, L.matched, L.self_mode,     \
                       L.hits, L.capacity, L.counters, L.overflow_row, base, L.row_pop3, L.row_pad, L.col_pop3, L.col_pad,                   \
                       reinterpret_cast<CandEntry *>(L.cand), L.cand_capacity, L.cand_head)
#define VDF_MFMA2_STEPS(WV)                                                                                         \
    switch (L.prune_step) { /* smallest instantiated step >= the requested one */                                   \
    case 0: case 1: case 2: case 3: case 4: case 5: case 6: VDF_MFMA2_LAUNCH(6, WV); break;                          \
    case 7: case 8: VDF_MFMA2_LAUNCH(8, WV); break;                                                                  \
    case 9: case 10: VDF_MFMA2_LAUNCH(10, WV); break;                                                                \
    case 11: VDF_MFMA2_LAUNCH(11, WV); break;                                                                        \
    case 12: VDF_MFMA2_LAUNCH(12, WV); break;                                                                        \
    case 13: VDF_MFMA2_LAUNCH(13, WV); break;                                                                        \
    case 14: VDF_MFMA2_LAUNCH(14, WV); break;                                                                        \
    default: VDF_MFMA2_LAUNCH(16, WV); break;                                                                        \
    }
        if (L.tile_rows == 512) { VDF_MFMA2_STEPS(8) } else { VDF_MFMA2_STEPS(4) }
#undef VDF_MFMA2_STEPS
#undef VDF_MFMA2_LAUNCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// second pass of the second-generation kernel: exact evaluation of the suspect pairs
hipError_t launch_resolve_candidates(const SearchLaunch &L, hipStream_t stream)
{
    hipLaunchKernelGGL(resolve_candidates_kernel, dim3(1024), dim3(256), 0, stream, reinterpret_cast<const CandEntry *>(L.cand),
                       L.cand_head, L.cand_capacity, L.row_hashes, L.row_perm, L.n_rows, L.row_index_base, L.col_hashes, L.n_cols,
                       L.row_lo, L.row_hi, L.tol, L.matched, L.self_mode, L.hits, L.capacity, L.counters, L.overflow_row);
    return hipGetLastError();
}

}  // namespace vdf
