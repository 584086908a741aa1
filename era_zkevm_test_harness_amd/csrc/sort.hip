// sort.hip — ordering step of the sorter circuits (K7). Replaces rayon `par_sort_by` at
// src/witness/individual_circuits/ram_permutation.rs:48-53.
//
// Device-wide radix sorts (rocPRIM) of the keys ram_sort (zkw_api.hip) packs: one sort of (block, page, index, timestamp) when the
// batch fits them into 64 bits, else a stable LSD composition: timestamp (32 bit), then cell =
// page<<32|index (64 bit), then block id (only as many bits as there are blocks). Stability of each
// pass makes the result identical to the reference's stable comparison sort by (page, index, ts).
// This is a plain library primitive (like a library GEMM); the kernels that are specific to this
// path are hand-written in ram_kernels.cuh.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "sort.h"

namespace zkw {

template <class K>
static hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const K* kin, K* kout, const uint32_t* vin,
                             uint32_t* vout, size_t n, unsigned end_bit, hipStream_t s) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0u, end_bit, s, false);
}

size_t radix_temp_bytes(size_t n) {
    size_t a = 0, b = 0;
    (void)sort_pairs<uint32_t>(nullptr, a, nullptr, nullptr, nullptr, nullptr, n, 32, 0);
    (void)sort_pairs<uint64_t>(nullptr, b, nullptr, nullptr, nullptr, nullptr, n, 64, 0);
    return a > b ? a : b;
}

hipError_t radix_sort_pairs_u32(void* tmp, size_t tmp_bytes, const uint32_t* kin, uint32_t* kout,
                                const uint32_t* vin, uint32_t* vout, size_t n, unsigned end_bit, hipStream_t s) {
    return sort_pairs<uint32_t>(tmp, tmp_bytes, kin, kout, vin, vout, n, end_bit, s);
}

hipError_t radix_sort_pairs_u64(void* tmp, size_t tmp_bytes, const uint64_t* kin, uint64_t* kout,
                                const uint32_t* vin, uint32_t* vout, size_t n, unsigned end_bit, hipStream_t s) {
    return sort_pairs<uint64_t>(tmp, tmp_bytes, kin, kout, vin, vout, n, end_bit, s);
}

}  // namespace zkw
