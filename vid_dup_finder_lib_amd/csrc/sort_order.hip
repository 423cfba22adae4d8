// Search::sort on the device (vid_dup_finder_lib/src/video_hashing/search_algorithm.rs:55-61): the stable order by
// (duration, src_path) of a hash database that is already resident in HBM - the step between hashing on the GPUs and
// searching there (vid_dup_finder_app/src/app/app_fns.rs:428-482 builds the Search from the cache and sorts it once).
// Paths stay with the caller: what decides between equal durations is each entry's RANK among the caller's paths in
// Rust's component-wise PathBuf order (equal paths -> equal ranks -> input order, sort_by_key is stable); without ranks all
// paths count as equal.  The sort itself is rocPRIM's stable LSD radix sort (a library primitive, not part of the hot path:
// 1 M entries take ~0.1 ms); the key / index preparation and the gather of the 128-byte hashes are the kernels below.
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "vdf_internal.h"

namespace vdf {

__global__ __launch_bounds__(256) void sort_keys_kernel(const uint32_t *__restrict__ dur, const uint32_t *__restrict__ rank,
                                                        uint32_t n, uint64_t *__restrict__ keys64,
                                                        uint32_t *__restrict__ idx)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        idx[i] = (uint32_t)i;
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

size_t sort_order_temp_bytes(uint32_t n, bool with_rank)
{
    size_t bytes = 0;
    if (with_rank)
        (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                                        (uint32_t *)nullptr, n);
    else
        (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                        (uint32_t *)nullptr, n);
    return bytes;
}

// scratch: [idx n u32][keys_out n (u32 | u64)][keys_in n u64 if with rank][rocprim temp]; perm_out receives the order.
hipError_t launch_sort_order(const uint32_t *dur, const uint32_t *rank, uint32_t n, uint32_t *perm_out, void *scratch,
                             size_t scratch_bytes, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const bool with_rank = rank != nullptr;
    const size_t a16 = 255;
    char *p = static_cast<char *>(scratch);
    uint32_t *idx = reinterpret_cast<uint32_t *>(p);
    p += ((size_t)n * 4 + a16) & ~a16;
    void *keys_out = p;
    p += ((size_t)n * (with_rank ? 8 : 4) + a16) & ~a16;
    uint64_t *keys_in = nullptr;
    if (with_rank) {
        keys_in = reinterpret_cast<uint64_t *>(p);
        p += ((size_t)n * 8 + a16) & ~a16;
    }
    size_t temp = sort_order_temp_bytes(n, with_rank);
    if ((size_t)(p - static_cast<char *>(scratch)) + temp > scratch_bytes) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sort_keys_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 4096u)), dim3(256), 0, stream, dur, rank, n,
                       keys_in, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (with_rank)
        return rocprim::radix_sort_pairs(p, temp, (const uint64_t *)keys_in, (uint64_t *)keys_out, (const uint32_t *)idx, perm_out, n, 0,
                                         64, stream);
    return rocprim::radix_sort_pairs(p, temp, dur, (uint32_t *)keys_out, (const uint32_t *)idx, perm_out, n, 0, 32, stream);
}

size_t sort_order_scratch_bytes(uint32_t n, bool with_rank)
{
    const size_t a16 = 255;
    size_t b = (((size_t)n * 4 + a16) & ~a16) + (((size_t)n * (with_rank ? 8 : 4) + a16) & ~a16);
    if (with_rank) b += ((size_t)n * 8 + a16) & ~a16;
    return b + sort_order_temp_bytes(n, with_rank) + 256;
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
// vdf_hit is {row, col}: as a little-endian u64 the row is the LOW half, so the keys are the halves swapped.
struct SwapHalves {
    __device__ __host__ uint64_t operator()(uint64_t v) const { return (v << 32) | (v >> 32); }
};

__global__ __launch_bounds__(256) void unswap_hits_kernel(const uint64_t *__restrict__ keys, size_t n, uint64_t *__restrict__ hits)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint64_t v = keys[i];
        hits[i] = (v << 32) | (v >> 32);
    }
}

static size_t sort_hits_temp_bytes(size_t n, unsigned end_bit)
{
    size_t bytes = 0;
    auto in = rocprim::make_transform_iterator((const uint64_t *)nullptr, SwapHalves{});
    (void)rocprim::radix_sort_keys(nullptr, bytes, in, (uint64_t *)nullptr, n, 0u, end_bit);
    return bytes;
}

size_t sort_hits_scratch_bytes(size_t n) { return ((n * 8 + 255) & ~(size_t)255) + sort_hits_temp_bytes(n, 64) + 256; }

// row_bits: rows are < 2^row_bits (fewer radix passes)
hipError_t launch_sort_hits(vdf_hit *hits, size_t n, unsigned row_bits, void *scratch, size_t scratch_bytes, hipStream_t stream)
{
    if (n < 2) return hipSuccess;
    const unsigned end_bit = std::min(64u, 32u + std::max(row_bits, 1u));
    uint64_t *keys = static_cast<uint64_t *>(scratch);
    char *temp = static_cast<char *>(scratch) + ((n * 8 + 255) & ~(size_t)255);
    size_t temp_bytes = sort_hits_temp_bytes(n, end_bit);
    if ((size_t)(temp - static_cast<char *>(scratch)) + temp_bytes > scratch_bytes) return hipErrorInvalidValue;
    auto in = rocprim::make_transform_iterator(reinterpret_cast<const uint64_t *>(hits), SwapHalves{});
    hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, in, keys, n, 0u, end_bit, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(unswap_hits_kernel, dim3((uint32_t)std::min<size_t>((n + 255) / 256, 1u << 16)), dim3(256), 0, stream, keys, n,
                       reinterpret_cast<uint64_t *>(hits));
    return hipGetLastError();
}

}  // namespace vdf
