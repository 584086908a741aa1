// vm_kernels.cuh — MainVM instance slicing (a19): the `vm_snapshots.windows(2)` loop of src/witness/oracle.rs:1229-1469 and
// the closed-form parts of vm_instance_witness_to_circuit_formal_input (src/witness/utils.rs:428-496) for gfx950.
// The reference walks every stream with skip_while / take_while per instance (O(instances x stream length)); on sorted
// cycle stamps those are lower bounds, so an instance is 8 x 2 + 4 binary searches and the read / write split of the
// memory stream is one stable partition (prefix counts) shared by all instances.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/zkw_types.h"

namespace zkw {
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int VM_TILE = 4096;  // memory-stream items per workgroup of the read/write partition

// reads per tile
static __device__ __forceinline__ void k_vm_rw_tile_counts(const VB& vb, const zkw_mem_query* __restrict__ q, u64 n, u32* __restrict__ tile_reads) {
    __shared__ u32 s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const u64 base = (u64)vb.x * VM_TILE;
    u32 c = 0;
    for (int k = threadIdx.x; k < VM_TILE; k += 256) {
        const u64 i = base + k;
        if (i < n && !q[i].rw_flag) c++;
    }
    for (int off = 32; off; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) tile_reads[vb.x] = s_cnt;
}

// exclusive scan of the tile counts (one workgroup: a block has at most a few thousand tiles)
static __device__ __forceinline__ void k_vm_rw_scan_tiles(const VB& vb, u32* __restrict__ tile_reads, u32 n_tiles, u64* __restrict__ total_reads) {
    __shared__ u32 s_wave[16];
    __shared__ u32 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n_tiles; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < n_tiles ? tile_reads[i] : 0;
        u32 x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(x, off);
            if ((threadIdx.x & 63) >= off) x += y;
        }
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
        __syncthreads();
        u32 before = s_carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += s_wave[w];
        if (i < n_tiles) tile_reads[i] = before + x - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_reads = s_carry;
}

// stable partition: read_prefix[i] = reads among [0, i); index arrays of the reads / writes in order
static __device__ __forceinline__ void k_vm_rw_scatter(const VB& vb, const zkw_mem_query* __restrict__ q, u64 n, const u32* __restrict__ tile_offsets,
                                                       u32* __restrict__ read_prefix, u32* __restrict__ read_index, u32* __restrict__ write_index) {
    __shared__ u32 s_wave[4];
    const u64 base = (u64)vb.x * VM_TILE + (u64)threadIdx.x * 16;  // 16 consecutive items per thread
    u32 flags = 0, c = 0;
    for (int k = 0; k < 16; k++) {
        const u64 i = base + k;
        if (i < n && !q[i].rw_flag) { flags |= 1u << k; c++; }
    }
    u32 x = c;
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off);
        if ((threadIdx.x & 63) >= off) x += y;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    u32 r = tile_offsets[vb.x] + x - c;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) r += s_wave[w];
    for (int k = 0; k < 16; k++) {
        const u64 i = base + k;
        if (i > n) break;
        read_prefix[i] = r;  // i == n included: the total
        if (i == n) break;
        if (flags >> k & 1) { if (read_index) read_index[r] = (u32)i; r++; }
        else if (write_index) write_index[i - r] = (u32)i;
    }
}

struct VmSliceJob {
    zkw_vm_tracer_streams s;       // device pointers
    const u32* read_prefix;        // [n_mem + 1]
    zkw_vm_instance* out;          // [n_snapshots - 1]
};

__device__ __forceinline__ u64 vm_lower_bound(const u32* a, u64 n, u32 c) {  // first index with a[i] >= c
    u64 lo = 0, hi = n;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if (a[mid] < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// the state of a full-width queue after its item k (FullWidthQueueIntermediateStates -> QueueStateWitness,
// transform_sponge_like_queue_state src/witness/utils.rs:73-85): head = the SIMULATOR's head (push_and_output_intermediate_data
// records `head: self.head`, circuit_encodings/src/lib.rs:419-421) = zero for the push-only memory / decommitment queues, tail =
// the state after the item, length = k + 1
__device__ __forceinline__ void vm_full_queue_state(const u64* tails, u64 k_plus_one, zkw_queue_state12* st) {
    for (int j = 0; j < 12; j++) {
        st->head[j] = 0;
        st->tail[j] = k_plus_one >= 1 ? tails[12 * (k_plus_one - 1) + j] : 0;
    }
    st->length = (u32)k_plus_one;
    st->_pad = 0;
}

__device__ void vm_aux_at(const zkw_vm_tracer_streams& s, u32 at_cycle, bool final_of_block, zkw_vm_aux_parameters* a) {
    memset(a, 0, sizeof *a);
    const u64 n_mem = s.stream_len[ZKW_VMS_MEMORY];
    // memory queue: the latest state with cycle < at_cycle (oracle.rs:1245-1253); the block's last state at the end (:1428-1434)
    const u64 m = final_of_block ? n_mem : vm_lower_bound(s.stream_cycles[ZKW_VMS_MEMORY], n_mem, at_cycle);
    vm_full_queue_state(s.memory_queue_tails, m, &a->memory_queue_state);
    const u64 d = final_of_block ? s.n_decommit_states : vm_lower_bound(s.decommit_state_cycles, s.n_decommit_states, at_cycle);
    vm_full_queue_state(s.decommit_queue_tails, d, &a->decommittment_queue_state);
    if (!final_of_block) {  // "always an empty one" on the last instance (:1418-1426)
        const u64 c = vm_lower_bound(s.callstack_sponge_cycles, s.n_callstack_sponges, at_cycle);
        if (c)
            for (int j = 0; j < 12; j++) a->callstack_state[j] = s.callstack_sponge_states[12 * (c - 1) + j];
    }
    const u64 l = final_of_block ? s.n_storage_log_states : vm_lower_bound(s.storage_log_state_cycles, s.n_storage_log_states, at_cycle);
    if (l) {
        const zkw_storage_log_detailed_state& st = s.storage_log_states[l - 1];
        for (int j = 0; j < 4; j++) {
            a->storage_log_queue_state.tail[j] = st.forward_tail[j];
            a->current_frame_rollback_queue_tail[j] = st.rollback_tail[j];
            a->current_frame_rollback_queue_head[j] = st.rollback_head[j];
        }
        a->storage_log_queue_state.length = st.forward_length;
        a->current_frame_rollback_queue_segment_length = st.rollback_length;
    } else if (!final_of_block) {  // StorageLogDetailedState::default() with both rollback ends at the block's end (:1359-1367)
        for (int j = 0; j < 4; j++) {
            a->current_frame_rollback_queue_tail[j] = s.global_end_of_storage_log[j];
            a->current_frame_rollback_queue_head[j] = s.global_end_of_storage_log[j];
        }
    }  // the last instance with an empty history: StorageLogDetailedState::default() as is (:1443-1447)
}

static __device__ __forceinline__ void k_vm_slice(const VB& vb, VmSliceJob job) {
    const zkw_vm_tracer_streams& s = job.s;
    const u64 n_inst = s.n_snapshots - 1;
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x;
    if (i >= n_inst) return;
    zkw_vm_instance& v = job.out[i];  // filled in place: a local copy would live in scratch memory
    memset(&v, 0, sizeof v);
    const u32 from = s.snapshot_cycles[i], to = s.snapshot_cycles[i + 1];
    v.start_flag = i == 0;
    v.completion_flag = i + 1 == n_inst;
    v.cycle_from = from;
    v.cycle_to = to;
    v.snapshot_initial = (u32)i;
    v.snapshot_final = (u32)i + 1;
    for (int k = 0; k < ZKW_VM_NUM_STREAMS; k++) {
        v.range[k][0] = vm_lower_bound(s.stream_cycles[k], s.stream_len[k], from);
        v.range[k][1] = vm_lower_bound(s.stream_cycles[k], s.stream_len[k], to);
    }
    const u64 lo = v.range[ZKW_VMS_MEMORY][0], hi = v.range[ZKW_VMS_MEMORY][1];
    const u64 r_lo = job.read_prefix ? job.read_prefix[lo] : 0, r_hi = job.read_prefix ? job.read_prefix[hi] : 0;
    v.first_memory_read = r_lo;
    v.num_memory_reads = r_hi - r_lo;
    v.first_memory_write = lo - r_lo;
    v.num_memory_writes = (hi - lo) - (r_hi - r_lo);
    vm_aux_at(s, from, false, &v.auxilary_initial_parameters);
    // "we will use next circuit's initial as final here" (:1399-1409); the block's final states on the last (:1414-1468)
    vm_aux_at(s, to, v.completion_flag != 0, &v.auxilary_final_parameters);
    if (v.start_flag) {  // utils.rs:456-469
        const zkw_vm_aux_parameters& a = v.auxilary_initial_parameters;
        for (int j = 0; j < 4; j++) v.rollback_queue_tail_for_block[j] = a.current_frame_rollback_queue_tail[j];
        for (int j = 0; j < 12; j++) {
            v.memory_queue_initial_tail[j] = a.memory_queue_state.tail[j];
            v.decommitment_queue_initial_tail[j] = a.decommittment_queue_state.tail[j];
        }
        v.memory_queue_initial_length = a.memory_queue_state.length;
        v.decommitment_queue_initial_length = a.decommittment_queue_state.length;
    }
    if (v.completion_flag) {  // utils.rs:471-483
        v.memory_queue_final_state = v.auxilary_final_parameters.memory_queue_state;
        v.decommitment_queue_final_state = v.auxilary_final_parameters.decommittment_queue_state;
        v.log_queue_final_state = v.auxilary_final_parameters.storage_log_queue_state;
    }
}

}  // namespace zkw
