// callstack_kernels.cuh — callstack entry encoding and the full-width stack simulator.
//
// Reference functions replaced:
//   k_encode_callstack   ExtendedCallstackEntry::encoding_witness              circuit_encodings/src/callstack_entry.rs:36-179
//   k_stack_*            FullWidthStackSimulator::{push,pop}_and_output_intermediate_data   circuit_encodings/src/lib.rs:558-644
//
// The reference replays pushes and pops one after another. The state of a stack is a function of its contents
// only: state(e1..ek) = absorb(state(e1..ek-1), enc(ek)). So every push is a node of a forest whose parent is
// the push that is on top of the stack at that moment (the latest earlier push one level lower), and all nodes
// of one depth can be hashed at once: depth of every operation by a prefix sum, parents and push/pop matches by
// a stable sort of the pushes by depth + binary searches, then one launch per depth level (4 permutations per
// node), then a gather per operation.
#pragma once
#include "scan_kernels.cuh"
#include "../../include/zkw_types.h"
#include "poseidon2.cuh"

namespace zkw {
using gl::u32;
using gl::u64;

constexpr u32 STACK_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ void encode_callstack_entry(const zkw_callstack_entry& e, u64 o[32]) {
#pragma unroll
    for (int k = 0; k < 4; k++) { o[k] = e.rollback_queue_head[k]; o[4 + k] = e.rollback_queue_tail[k]; }
#pragma unroll
    for (int k = 0; k < 5; k++) { o[8 + k] = e.code_address[k]; o[13 + k] = e.this_address[k]; o[18 + k] = e.msg_sender[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) o[23 + k] = e.context_u128_value[k];
    const u32* a = e.this_address;
    const u64 kernel = (a[4] | a[3] | a[2] | a[1]) == 0 && a[0] < (1u << 16);
    o[27] = (u64)e.code_page | ((u64)e.pc << 32) | ((u64)e.this_shard_id << 48) | ((u64)(e.is_static ? 1 : 0) << 56);
    o[28] = (u64)e.base_memory_page | ((u64)e.sp << 32) | ((u64)e.caller_shard_id << 48) | (kernel << 56);
    o[29] = (u64)e.ergs_remaining | ((u64)e.exception_handler_location << 32) | ((u64)e.code_shard_id << 48) |
            ((u64)(e.is_local_frame ? 1 : 0) << 56);
    const u32 len = e.rollback_queue_segment_length;
    o[30] = (u64)e.heap_bound | ((u64)(len & 0xFF) << 32) | ((u64)((len >> 8) & 0xFF) << 40);
    o[31] = (u64)e.aux_heap_bound | ((u64)((len >> 16) & 0xFF) << 32) | ((u64)(len >> 24) << 40);
}

static __global__ __launch_bounds__(64) void k_encode_callstack(const zkw_callstack_entry* __restrict__ e, size_t n, u64* __restrict__ enc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 o[32];
    encode_callstack_entry(e[i], o);
    for (int k = 0; k < 32; k++) enc[32 * i + k] = o[k];
}

// The depth after every operation and the number of pushes before it are two prefix sums over the operations (+1 / -1 per push / pop,
// 1 per push): sum_prefix<2> (scan_kernels.cuh, tiled: any number of operations), then one lane per operation.
struct StackDelta {
    const uint8_t* is_push;
    __device__ void operator()(size_t i, u64 v[2]) const {
        const bool push = is_push[i] != 0;
        v[0] = push ? 1ull : ~0ull;  // -1 in wrap-around arithmetic
        v[1] = push ? 1ull : 0ull;
    }
};
// meta[0] = number of pushes, meta[1] = maximum depth, meta[2] = error (1: pop from the empty stack); meta zeroed by the caller
static __global__ __launch_bounds__(256) void k_stack_depth(const uint8_t* __restrict__ is_push, size_t n, const u64* __restrict__ prefix /* [2][n + 1] */,
                                                            u32* __restrict__ depth_after, u32* __restrict__ push_rank, u32* __restrict__ meta) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) meta[0] = (u32)prefix[(n + 1) + n];
    if (i >= n) return;
    const long long d = (long long)prefix[i] + (is_push[i] ? 1 : -1);
    if (d < 0) atomicOr(&meta[2], 1u);
    depth_after[i] = (u32)d;
    push_rank[i] = (u32)prefix[(n + 1) + i];  // pushes strictly before i
    if (d > 0) atomicMax(&meta[1], (u32)d);
}

static __global__ __launch_bounds__(256) void k_stack_push_keys(const uint8_t* __restrict__ is_push, size_t n, const u32* __restrict__ depth_after,
                                                         const u32* __restrict__ push_rank, u32* __restrict__ push_depth,
                                                         u32* __restrict__ push_id) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !is_push[i]) return;
    const u32 k = push_rank[i];
    push_depth[k] = depth_after[i];
    push_id[k] = k;
}

__device__ __forceinline__ u32 lower_bound_u32(const u32* a, u32 n, u32 v) {
    u32 lo = 0, hi = n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// the latest push k < limit whose depth is d (sorted_depth ascending, sorted_id ascending inside a depth)
__device__ __forceinline__ u32 stack_find(const u32* sorted_depth, const u32* sorted_id, u32 n_push, u32 d, u32 limit) {
    const u32 lo = lower_bound_u32(sorted_depth, n_push, d), hi = lower_bound_u32(sorted_depth, n_push, d + 1);
    const u32 pos = lo + lower_bound_u32(sorted_id + lo, hi - lo, limit);
    return pos > lo ? sorted_id[pos - 1] : STACK_NONE;
}

// per operation: for a push its parent node, for a pop the node it removes
static __global__ __launch_bounds__(256) void k_stack_links(const uint8_t* __restrict__ is_push, size_t n, const u32* __restrict__ depth_after,
                                                     const u32* __restrict__ push_rank, const u32* __restrict__ sorted_depth,
                                                     const u32* __restrict__ sorted_id, const u32* __restrict__ meta,
                                                     u32* __restrict__ parent, u32* __restrict__ op_node) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || meta[2]) return;
    const u32 n_push = meta[0], rank = push_rank[i], d = depth_after[i];
    if (is_push[i]) {
        parent[rank] = d > 1 ? stack_find(sorted_depth, sorted_id, n_push, d - 1, rank) : STACK_NONE;
        op_node[i] = rank;
    } else {
        op_node[i] = stack_find(sorted_depth, sorted_id, n_push, d + 1, rank);
    }
}

// all nodes of depth d: rounds[k] = the 4 sponge states of absorbing enc(entry k) into the parent's state
static __global__ __launch_bounds__(64) void k_stack_level(const zkw_callstack_entry* __restrict__ pushed, const u32* __restrict__ push_depth,
                                                    const u32* __restrict__ parent, const u32* __restrict__ meta, u32 d,
                                                    u64* __restrict__ rounds) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= meta[0] || push_depth[k] != d) return;
    __shared__ u64 sh_enc[64 * 32];  // a slice of LDS per lane instead of a per-lane array in scratch memory
    u64* enc = sh_enc + 32 * threadIdx.x;
    u64 s[12];
    encode_callstack_entry(pushed[k], enc);
    const u32 par = parent[k];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = par == STACK_NONE ? 0 : rounds[(size_t)48 * par + 36 + j];
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = enc[8 * r + j];
        p2::permute(s);
#pragma unroll
        for (int j = 0; j < 12; j++) { s[j] = gl::canon(s[j]); rounds[(size_t)48 * k + 12 * r + j] = s[j]; }
    }
}

static __global__ __launch_bounds__(256) void k_stack_emit(const uint8_t* __restrict__ is_push, size_t n, const u32* __restrict__ depth_after,
                                                    const u32* __restrict__ parent, const u32* __restrict__ op_node,
                                                    const u64* __restrict__ rounds, const u32* __restrict__ meta,
                                                    u64* __restrict__ previous_state, u64* __restrict__ new_state,
                                                    u32* __restrict__ depth, u64* __restrict__ round_states, u32* __restrict__ entry_index) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || meta[2]) return;
    const u32 k = op_node[i], par = parent[k];
    const bool push = is_push[i];
    for (int j = 0; j < 12; j++) {
        const u64 own = rounds[(size_t)48 * k + 36 + j];
        const u64 below = par == STACK_NONE ? 0 : rounds[(size_t)48 * par + 36 + j];
        previous_state[12 * i + j] = push ? below : own;
        new_state[12 * i + j] = push ? own : below;
    }
    for (int j = 0; j < 48; j++) round_states[48 * i + j] = rounds[(size_t)48 * k + j];
    depth[i] = depth_after[i];
    entry_index[i] = k;
}

}  // namespace zkw
