// Host-side radix sorts of the search path (no HIP): hit lists into (row, col) order, stable argsort of durations.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/vdf.h"

namespace vdf_impl {

// Hits into (row, col) order.  LSD radix sort over the 64-bit key row << 32 | col, 11-bit digits, digits that every key
// shares skipped (row < 2^17 and col < 2^20 leave four passes): 50 k hits 0.3 ms, where std::sort through a comparator took 2-3 ms.
void sort_hits(vdf_hit *hits, size_t n);

// perm[0..n) = the stable ascending order of keys (LSD radix sort, 11-bit digits; a digit every key shares is skipped -
// durations rarely need the third).  std::stable_sort through an index comparator took 8 ms for 100 k references, ten times
// the search kernel; this takes ~0.4 ms.
void stable_argsort_u32(const uint32_t *keys, size_t n, uint32_t *perm);

}  // namespace vdf_impl
