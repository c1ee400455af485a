// sortscan.hip — thin wrappers so that only this translation unit pays for the rocPRIM templates.
#include "sortscan.h"

#include <hipcub/hipcub.hpp>

namespace mhip {

hipError_t sort_pairs_u32(void* temp, size_t& temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                          const int32_t* vals_in, int32_t* vals_out, int n, int end_bit, hipStream_t s) {
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit, s);
}

hipError_t exclusive_sum_i32(void* temp, size_t& temp_bytes, const int32_t* in, int32_t* out, int n, hipStream_t s) {
    return hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, n, s);
}

}  // namespace mhip
