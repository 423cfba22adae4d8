#include "host_sort.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace vdf_impl {

void sort_hits(vdf_hit *hits, size_t n)
{
    auto less = [](const vdf_hit &a, const vdf_hit &b) { return a.row != b.row ? a.row < b.row : a.col < b.col; };
    if (n < 4096) { std::sort(hits, hits + n, less); return; }
    auto key = [](const vdf_hit &h) { return ((uint64_t)h.row << 32) | h.col; };
    uint64_t all_or = 0, all_and = ~0ull;
    for (size_t i = 0; i < n; i++) { const uint64_t k = key(hits[i]); all_or |= k; all_and &= k; }
    std::vector<vdf_hit> tmp(n);
    vdf_hit *src = hits, *dst = tmp.data();
    for (int shift = 0; shift < 64; shift += 11) {
        if ((((all_or ^ all_and) >> shift) & 0x7FFull) == 0) continue;  // every key has the same digit here
        size_t count[2049] = {0};
        for (size_t i = 0; i < n; i++) count[((key(src[i]) >> shift) & 0x7FFull) + 1]++;
        for (int d = 0; d < 2048; d++) count[d + 1] += count[d];
        for (size_t i = 0; i < n; i++) dst[count[(key(src[i]) >> shift) & 0x7FFull]++] = src[i];
        std::swap(src, dst);
    }
    if (src != hits) std::memcpy(hits, src, n * sizeof(vdf_hit));
}

}  // namespace vdf_impl

extern "C" int vdf_sort_hits(vdf_hit *hits, uint64_t n_hits)
{
    if (n_hits && !hits) return VDF_E_INVAL;
    vdf_impl::sort_hits(hits, (size_t)n_hits);
    return VDF_OK;
}
