// sortscan.h — device radix sort / prefix sums used by the neighbour rebuild (rocPRIM via hipCUB).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace mhip {
// stable key/value radix sort on bits [0, end_bit); temp == nullptr → only report temp_bytes
hipError_t sort_pairs_u32(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                          const int32_t* vals_in, int32_t* vals_out, int n, int end_bit, hipStream_t s);
hipError_t exclusive_sum_i32(void* temp, size_t& temp_bytes, const int32_t* in, int32_t* out, int n, hipStream_t s);
}  // namespace mhip
