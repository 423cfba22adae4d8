#pragma once
#include <cstdint>
#include <vector>

namespace vdf {

struct HostAxisTable {
    std::vector<int32_t> start;  // [out] first source index
    std::vector<int32_t> size;   // [out] taps
    std::vector<int16_t> w;      // [out][window]
    int32_t window = 0;
    int32_t precision = 0;
};

bool build_axis_table(uint32_t in_size, uint32_t out_size, HostAxisTable &t);

}  // namespace vdf
