// public_input_kernels.cuh — closed-form input commitments and the recursion queue.
//
// Reference functions replaced:
//   k_commit_encodings   commit_variable_length_encodable_item, as driven by
//                        simulate_public_input_value_from_witness         src/witness/utils.rs:269-306
//   k_ram_commitments    ClosedFormInputCompactForm::from_full_form for RamPermutationInputOutput
//                        (+ the shared observable input)                  src/witness/postprocessing/mod.rs:353-369
//   k_encode_recursion   RecursionRequest::encoding_witness               circuit_encodings/src/recursion_request.rs:13-28
//
// Every commitment is a short serial sponge (<= 9 permutations), so the parallel axis is (instance, part):
// one lane each, the three non-trivial parts of an instance in adjacent lanes.
#pragma once
#include "../../include/zkw_types.h"
#include "poseidon2.cuh"

namespace zkw {
using gl::u32;
using gl::u64;

constexpr int RAM_INPUT_ENC_LEN = 51;
constexpr int RAM_FSM_ENC_LEN = 69;
constexpr int COMPACT_FORM_LEN = 18;

// sponge in overwrite mode from the zero state, length in the last capacity word, last chunk zero padded
__device__ inline void commit_var_length(const u64* enc, int n, u64 out[4]) {
    u64 s[12];
    for (int i = 0; i < 12; i++) s[i] = 0;
    s[11] = (u64)n;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        for (int k = 0; k < 8; k++) s[k] = enc[i + k];
        p2::permute(s);
    }
    if (i < n) {
        for (int k = 0; k < 8; k++) s[k] = (i + k < n) ? enc[i + k] : 0;
        p2::permute(s);
    }
    for (int k = 0; k < 4; k++) out[k] = gl::canon(s[k]);
}

__global__ __launch_bounds__(64) void k_commit_encodings(const u64* __restrict__ enc, size_t n_items, u32 item_len,
                                                         u64* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    u64 c[4];
    commit_var_length(enc + (size_t)item_len * i, (int)item_len, c);
    for (int k = 0; k < 4; k++) out[4 * i + k] = c[k];
}

__device__ inline int put_queue12(const zkw_queue_state12& q, u64* o) {
    for (int k = 0; k < 12; k++) { o[k] = q.head[k]; o[12 + k] = q.tail[k]; }
    o[24] = q.length;
    return 25;
}

__device__ inline int ram_encode_fsm(const zkw_ram_fsm& f, u64* o) {
    int m = 0;
    for (int r = 0; r < 2; r++) o[m++] = f.lhs_accumulator[r];
    for (int r = 0; r < 2; r++) o[m++] = f.rhs_accumulator[r];
    m += put_queue12(f.current_unsorted_queue_state, o + m);
    m += put_queue12(f.current_sorted_queue_state, o + m);
    for (int k = 0; k < 3; k++) o[m++] = f.previous_sorting_key[k];
    for (int k = 0; k < 2; k++) o[m++] = f.previous_full_key[k];
    for (int k = 0; k < 8; k++) o[m++] = f.previous_value[k];
    o[m++] = f.previous_is_ptr ? 1 : 0;
    o[m++] = f.num_nondeterministic_writes;
    return m;
}

// lane = 4 * instance + part; part 0: observable input of the block's FIRST instance, 1: flags + empty output,
// 2: hidden FSM input, 3: hidden FSM output
__global__ __launch_bounds__(64) void k_ram_commitments(const zkw_ram_instance* __restrict__ inst, size_t n,
                                                        u64* __restrict__ compact) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t >> 2;
    const int part = (int)(t & 3);
    if (i >= n) return;
    u64* cf = compact + COMPACT_FORM_LEN * i;
    __shared__ u64 sh_buf[64 * RAM_FSM_ENC_LEN];  // in LDS, one slice per lane: a run-time-indexed per-lane array would live in scratch memory
    u64* buf = sh_buf + threadIdx.x * RAM_FSM_ENC_LEN;
    u64 c[4];
    if (part == 1) {
        cf[0] = inst[i].start_flag ? 1 : 0;
        cf[1] = inst[i].completion_flag ? 1 : 0;
        for (int k = 0; k < 4; k++) cf[6 + k] = 0;  // observable output is (): nothing absorbed
        return;
    }
    int m;
    if (part == 0) {
        size_t j = i;
        while (j > 0 && !inst[j].start_flag) j--;
        m = put_queue12(inst[j].unsorted_queue_initial_state, buf);
        m += put_queue12(inst[j].sorted_queue_initial_state, buf + m);
        buf[m++] = inst[j].non_deterministic_bootloader_memory_snapshot_length;
    } else {
        m = ram_encode_fsm(part == 2 ? inst[i].hidden_fsm_input : inst[i].hidden_fsm_output, buf);
    }
    commit_var_length(buf, m, c);
    const int at = part == 0 ? 2 : (part == 2 ? 10 : 14);
    for (int k = 0; k < 4; k++) cf[at + k] = c[k];
}

// CodeDecommittmentsSorter (type 2): same lane plan; part 1 commits the observable output (the final queue state)
constexpr int DS_FSM_ENC_LEN = 100;
__device__ inline int ds_encode_fsm(const zkw_decommit_sorter_fsm& f, u64* o) {
    int m = put_queue12(f.initial_queue_state, o);
    m += put_queue12(f.sorted_queue_state, o + m);
    m += put_queue12(f.final_queue_state, o + m);
    for (int r = 0; r < 2; r++) o[m++] = f.lhs_accumulator[r];
    for (int r = 0; r < 2; r++) o[m++] = f.rhs_accumulator[r];
    for (int k = 0; k < 9; k++) o[m++] = f.previous_packed_key[k];
    for (int k = 0; k < 8; k++) o[m++] = f.previous_record.hash[k];
    o[m++] = f.previous_record.memory_page;
    o[m++] = f.previous_record.is_fresh ? 1 : 0;
    o[m++] = f.previous_record.timestamp;
    o[m++] = f.first_encountered_timestamp;
    return m;
}
__global__ __launch_bounds__(64) void k_ds_commitments(const zkw_decommit_sorter_instance* __restrict__ inst, size_t n,
                                                       u64* __restrict__ compact) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t >> 2;
    const int part = (int)(t & 3);
    if (i >= n) return;
    u64* cf = compact + COMPACT_FORM_LEN * i;
    __shared__ u64 sh_buf[64 * DS_FSM_ENC_LEN];  // in LDS, one slice per lane: a run-time-indexed per-lane array would live in scratch memory
    u64* buf = sh_buf + threadIdx.x * DS_FSM_ENC_LEN;
    u64 c[4];
    int m;
    if (part == 0) {
        size_t j = i;
        while (j > 0 && !inst[j].start_flag) j--;
        m = put_queue12(inst[j].initial_queue_state, buf);
        m += put_queue12(inst[j].sorted_queue_initial_state, buf + m);
    } else if (part == 1) {
        cf[0] = inst[i].start_flag ? 1 : 0;
        cf[1] = inst[i].completion_flag ? 1 : 0;
        m = put_queue12(inst[i].final_queue_state, buf);
    } else {
        m = ds_encode_fsm(part == 2 ? inst[i].hidden_fsm_input : inst[i].hidden_fsm_output, buf);
    }
    commit_var_length(buf, m, c);
    const int at = part == 0 ? 2 : (part == 1 ? 6 : (part == 2 ? 10 : 14));
    for (int k = 0; k < 4; k++) cf[at + k] = c[k];
}

// ---- the 4-wide log-queue circuits (LogDemuxer 4, StorageSorter 9, EventsSorter / L1MessagesSorter 11 / 12): one
// templated kernel over a per-type encoder; same lane plan as above. Field order = the struct declarations as
// mirrored by include/zkw_types.h (the reference's struct literals: log_demux.rs:283-301, storage_sort_dedup.rs:
// 577-612, events_sort_dedup.rs:426-455); LogQuery in the declaration order of the in-circuit struct.
__device__ inline int put_queue4(const zkw_queue_state4& q, u64* o) {
    for (int k = 0; k < 4; k++) { o[k] = q.head[k]; o[4 + k] = q.tail[k]; }
    o[8] = q.length;
    return 9;
}
__device__ inline int put_log_query(const zkw_log_query& q, u64* o) {
    int m = 0;
    for (int k = 0; k < 5; k++) o[m++] = q.address[k];
    for (int k = 0; k < 8; k++) o[m++] = q.key[k];
    for (int k = 0; k < 8; k++) o[m++] = q.read_value[k];
    for (int k = 0; k < 8; k++) o[m++] = q.written_value[k];
    o[m++] = q.rw_flag ? 1 : 0;
    o[m++] = q.aux_byte;
    o[m++] = q.rollback ? 1 : 0;
    o[m++] = q.is_service ? 1 : 0;
    o[m++] = q.shard_id;
    o[m++] = q.tx_number_in_block;
    o[m++] = q.timestamp;
    return m;
}
struct CfLogDemux {
    using Inst = zkw_log_demux_instance;
    static constexpr int MAXLEN = 63;
    __device__ static int input(const Inst& w, u64* o) { return put_queue4(w.initial_log_queue_state, o); }
    __device__ static int output(const Inst& w, u64* o) {
        int m = 0;
        for (int c = 0; c < ZKW_DEMUX_NUM_QUEUES; c++) m += put_queue4(w.output_queue_state[c], o + m);
        return m;
    }
    __device__ static int fsm(const zkw_log_demux_fsm& f, u64* o) {
        int m = put_queue4(f.initial_log_queue_state, o);
        for (int c = 0; c < ZKW_DEMUX_NUM_QUEUES; c++) m += put_queue4(f.queue_state[c], o + m);
        return m;
    }
    __device__ static const zkw_log_demux_fsm& fsm_in(const Inst& w) { return w.hidden_fsm_input; }
    __device__ static const zkw_log_demux_fsm& fsm_out(const Inst& w) { return w.hidden_fsm_output; }
};
struct CfEventsSorter {
    using Inst = zkw_events_sorter_instance;
    static constexpr int MAXLEN = 68;
    __device__ static int input(const Inst& w, u64* o) {
        int m = put_queue4(w.initial_log_queue_state, o);
        return m + put_queue4(w.intermediate_sorted_queue_state, o + m);
    }
    __device__ static int output(const Inst& w, u64* o) { return put_queue4(w.final_queue_state, o); }
    __device__ static int fsm(const zkw_events_sorter_fsm& f, u64* o) {
        int m = 0;
        for (int r = 0; r < 2; r++) o[m++] = f.lhs_accumulator[r];
        for (int r = 0; r < 2; r++) o[m++] = f.rhs_accumulator[r];
        m += put_queue4(f.initial_unsorted_queue_state, o + m);
        m += put_queue4(f.intermediate_sorted_queue_state, o + m);
        m += put_queue4(f.final_result_queue_state, o + m);
        o[m++] = f.previous_key;
        return m + put_log_query(f.previous_item, o + m);
    }
    __device__ static const zkw_events_sorter_fsm& fsm_in(const Inst& w) { return w.hidden_fsm_input; }
    __device__ static const zkw_events_sorter_fsm& fsm_out(const Inst& w) { return w.hidden_fsm_output; }
};
struct CfStorageSorter {
    using Inst = zkw_storage_sorter_instance;
    static constexpr int MAXLEN = 77;
    __device__ static int input(const Inst& w, u64* o) {
        o[0] = w.shard_id_to_process;
        int m = 1 + put_queue4(w.unsorted_log_queue_state, o + 1);
        return m + put_queue4(w.intermediate_sorted_queue_state, o + m);
    }
    __device__ static int output(const Inst& w, u64* o) { return put_queue4(w.final_sorted_queue_state, o); }
    __device__ static int fsm(const zkw_storage_sorter_fsm& f, u64* o) {
        int m = 0;
        for (int r = 0; r < 2; r++) o[m++] = f.lhs_accumulator[r];
        for (int r = 0; r < 2; r++) o[m++] = f.rhs_accumulator[r];
        m += put_queue4(f.current_unsorted_queue_state, o + m);
        m += put_queue4(f.current_intermediate_sorted_queue_state, o + m);
        m += put_queue4(f.current_final_sorted_queue_state, o + m);
        o[m++] = f.cycle_idx;
        for (int k = 0; k < ZKW_STORAGE_PACKED_KEY_LENGTH; k++) o[m++] = f.previous_packed_key[k];
        for (int k = 0; k < 8; k++) o[m++] = f.previous_key[k];
        for (int k = 0; k < 5; k++) o[m++] = f.previous_address[k];
        o[m++] = f.previous_timestamp;
        o[m++] = f.this_cell_has_explicit_read_and_rollback_depth_zero ? 1 : 0;
        for (int k = 0; k < 8; k++) o[m++] = f.this_cell_base_value[k];
        for (int k = 0; k < 8; k++) o[m++] = f.this_cell_current_value[k];
        o[m++] = f.this_cell_current_depth;
        return m;
    }
    __device__ static const zkw_storage_sorter_fsm& fsm_in(const Inst& w) { return w.hidden_fsm_input; }
    __device__ static const zkw_storage_sorter_fsm& fsm_out(const Inst& w) { return w.hidden_fsm_output; }
};

template <class T>
__global__ __launch_bounds__(64) void k_closed_form_commitments(const typename T::Inst* __restrict__ inst, size_t n,
                                                                u64* __restrict__ compact) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = t >> 2;
    const int part = (int)(t & 3);
    if (i >= n) return;
    u64* cf = compact + COMPACT_FORM_LEN * i;
    __shared__ u64 sh_buf[64 * T::MAXLEN];  // in LDS, one slice per lane: a run-time-indexed per-lane array would live in scratch memory
    u64* buf = sh_buf + threadIdx.x * T::MAXLEN;
    u64 c[4];
    int m;
    if (part == 0) {  // the observable input is the one of the block's first instance (postprocessing/mod.rs:358-364)
        size_t j = i;
        while (j > 0 && !inst[j].start_flag) j--;
        m = T::input(inst[j], buf);
    } else if (part == 1) {
        cf[0] = inst[i].start_flag ? 1 : 0;
        cf[1] = inst[i].completion_flag ? 1 : 0;
        m = T::output(inst[i], buf);
    } else {
        m = T::fsm(part == 2 ? T::fsm_in(inst[i]) : T::fsm_out(inst[i]), buf);
    }
    commit_var_length(buf, m, c);
    const int at = part == 0 ? 2 : (part == 1 ? 6 : (part == 2 ? 10 : 14));
    for (int k = 0; k < 4; k++) cf[at + k] = c[k];
}

__global__ __launch_bounds__(64) void k_encode_recursion(u64 circuit_type, const u64* __restrict__ pi, size_t n,
                                                         u64* __restrict__ enc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    enc[8 * i] = circuit_type;
    for (int k = 0; k < 4; k++) enc[8 * i + 1 + k] = pi[4 * i + k];
    for (int k = 5; k < 8; k++) enc[8 * i + k] = 0;
}

}  // namespace zkw
