// sort.h — device-wide stable radix sorts used by the sorter builders (implemented in sort.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zkw {
size_t radix_temp_bytes(size_t n);
hipError_t radix_sort_pairs_u32(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout,
                                const uint32_t* vin, uint32_t* vout, size_t n, unsigned end_bit, hipStream_t s);
hipError_t radix_sort_pairs_u64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout,
                                const uint32_t* vin, uint32_t* vout, size_t n, unsigned end_bit, hipStream_t s);
}  // namespace zkw
