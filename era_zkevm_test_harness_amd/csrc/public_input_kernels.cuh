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
        p2::permute_lat(s);  // rolled one-lane sponge: see p2::permute<LAT>
    }
    if (i < n) {
        for (int k = 0; k < 8; k++) s[k] = (i + k < n) ? enc[i + k] : 0;
        p2::permute_lat(s);  // rolled one-lane sponge: see p2::permute<LAT>
    }
    for (int k = 0; k < 4; k++) out[k] = gl::canon(s[k]);
}

static __device__ __forceinline__ void k_commit_encodings(const VB& vb, const u64* __restrict__ enc, size_t n_items, u32 item_len,
                                                         u64* __restrict__ out) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
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
static __device__ __forceinline__ void k_ram_commitments(const VB& vb, const zkw_ram_instance* __restrict__ inst, size_t n,
                                                        u64* __restrict__ compact) {
    const size_t t = (size_t)vb.x * blockDim.x + threadIdx.x;
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
static __device__ __forceinline__ void k_ds_commitments(const VB& vb, const zkw_decommit_sorter_instance* __restrict__ inst, size_t n,
                                                       u64* __restrict__ compact) {
    const size_t t = (size_t)vb.x * blockDim.x + threadIdx.x;
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

// ---- the remaining non-VM circuits: CodeDecommitter 3, Keccak256 / Sha256 / ECRecover round functions 5 / 6 / 7,
// StorageApplication 10, LinearHasher 13. Struct declarations of the absent zkevm_circuits crate (v1.4.1, */input.rs),
// field sets as the reference's builders fill them (decommit_code.rs:172-199,363-401; keccak256_round_function.rs:
// 420-441; sha256_round_function.rs:302-316; ecrecover.rs:215-233; storage_application.rs:286-336;
// data_hasher_and_merklizer.rs:34-60). One field element per Boolean / UInt8 / UInt16 / UInt32.
__device__ inline int put_bytes(const uint8_t* b, int n, u64* o) {
    for (int k = 0; k < n; k++) o[k] = b[k];
    return n;
}
__device__ inline int put_u32s(const u32* w, int n, u64* o) {
    for (int k = 0; k < n; k++) o[k] = w[k];
    return n;
}
struct CfDecommitter {
    using Inst = zkw_decommitter_instance;
    static constexpr int MAXLEN = 80, LANES = 64;  // FSM: 8 + 8 + 5 + 3 + 25 + 25 = 74
    __device__ static int input(const Inst& w, u64* o) {
        int m = put_queue12(w.memory_queue_initial_state, o);
        return m + put_queue12(w.sorted_requests_queue_initial_state, o + m);
    }
    __device__ static int output(const Inst& w, u64* o) { return put_queue12(w.memory_queue_final_state, o); }
    __device__ static int fsm(const zkw_decommitter_fsm& f, u64* o) {
        int m = put_u32s(f.sha256_inner_state, 8, o);
        m += put_u32s(f.hash_to_compare_against, 8, o + m);
        o[m++] = f.current_index;
        o[m++] = f.current_page;
        o[m++] = f.timestamp;
        o[m++] = f.num_rounds_left;
        o[m++] = f.length_in_bits;
        o[m++] = f.state_get_from_queue ? 1 : 0;
        o[m++] = f.state_decommit ? 1 : 0;
        o[m++] = f.finished ? 1 : 0;
        m += put_queue12(f.decommittment_requests_queue_state, o + m);
        return m + put_queue12(f.memory_queue_state, o + m);
    }
    __device__ static const zkw_decommitter_fsm& fsm_in(const Inst& w) { return w.hidden_fsm_input; }
    __device__ static const zkw_decommitter_fsm& fsm_out(const Inst& w) { return w.hidden_fsm_output; }
};
template <int KIND>
struct CfPrecompile {
    using Inst = zkw_precompile_instance;
    // keccak: 4 flags + 200 + 2 timestamps + 6 parameters + 192 + 1 buffer + 9 + 25 queue words
    static constexpr int MAXLEN = KIND == ZKW_PRECOMPILE_KECCAK256 ? 440 : 56, LANES = KIND == ZKW_PRECOMPILE_KECCAK256 ? 16 : 64;
    __device__ static int input(const Inst& w, u64* o) {
        int m = put_queue4(w.initial_log_queue_state, o);
        return m + put_queue12(w.initial_memory_queue_state, o + m);
    }
    __device__ static int output(const Inst& w, u64* o) { return put_queue12(w.final_memory_state, o); }
    __device__ static int fsm(const zkw_precompile_fsm& f, u64* o) {
        int m = 0;
        if (KIND == ZKW_PRECOMPILE_KECCAK256) {
            o[m++] = f.read_precompile_call ? 1 : 0;
            o[m++] = f.read_words_for_round ? 1 : 0;
            o[m++] = f.completed ? 1 : 0;
            o[m++] = f.padding_round ? 1 : 0;
            m += put_bytes(f.keccak_internal_state, 200, o + m);
            o[m++] = f.timestamp_to_use_for_read;
            o[m++] = f.timestamp_to_use_for_write;
            o[m++] = f.input_page;
            o[m++] = f.input_offset;
            o[m++] = f.input_length;
            o[m++] = f.output_page;
            o[m++] = f.output_offset;
            o[m++] = f.needs_full_padding_round ? 1 : 0;
            m += put_bytes(f.buffer_bytes, ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE, o + m);
            o[m++] = f.buffer_filled;
        } else if (KIND == ZKW_PRECOMPILE_SHA256) {
            o[m++] = f.read_precompile_call ? 1 : 0;
            o[m++] = f.read_words_for_round ? 1 : 0;
            o[m++] = f.completed ? 1 : 0;
            m += put_u32s(f.sha256_inner_state, 8, o + m);
            o[m++] = f.timestamp_to_use_for_read;
            o[m++] = f.timestamp_to_use_for_write;
            o[m++] = f.input_page;
            o[m++] = f.input_offset;
            o[m++] = f.output_page;
            o[m++] = f.output_offset;
            o[m++] = f.num_rounds;
        }
        m += put_queue4(f.log_queue_state, o + m);
        return m + put_queue12(f.memory_queue_state, o + m);
    }
    __device__ static const zkw_precompile_fsm& fsm_in(const Inst& w) { return w.hidden_fsm_input; }
    __device__ static const zkw_precompile_fsm& fsm_out(const Inst& w) { return w.hidden_fsm_output; }
};
struct CfStorageApplication {
    using Inst = zkw_storage_application_instance;
    static constexpr int MAXLEN = 244, LANES = 32;
    __device__ static int input(const Inst& w, u64* o) {
        o[0] = w.shard;
        int m = 1 + put_bytes(w.initial_root_hash, 32, o + 1);
        m += put_u32s(w.initial_next_enumeration_counter, 2, o + m);
        return m + put_queue4(w.storage_application_log_state, o + m);
    }
    __device__ static int output(const Inst& w, u64* o) {
        int m = put_bytes(w.new_root_hash, 32, o);
        m += put_u32s(w.new_next_enumeration_counter, 2, o + m);
        return m + put_bytes(w.state_diffs_keccak256_hash, 32, o + m);
    }
    __device__ static int fsm(const zkw_storage_application_fsm& f, u64* o) {
        int m = put_bytes(f.current_root_hash, 32, o);
        m += put_u32s(f.next_enumeration_counter, 2, o + m);
        m += put_queue4(f.current_storage_application_log_state, o + m);
        return m + put_bytes(f.current_diffs_keccak_accumulator_state, 200, o + m);
    }
    __device__ static const zkw_storage_application_fsm& fsm_in(const Inst& w) { return w.hidden_fsm_input; }
    __device__ static const zkw_storage_application_fsm& fsm_out(const Inst& w) { return w.hidden_fsm_output; }
};
struct CfLinearHasher {
    using Inst = zkw_linear_hasher_instance;
    struct NoFsm {};
    static constexpr int MAXLEN = 32, LANES = 64;
    __device__ static int input(const Inst& w, u64* o) { return put_queue4(w.queue_state, o); }
    __device__ static int output(const Inst& w, u64* o) { return put_bytes(w.keccak256_hash, 32, o); }
    __device__ static int fsm(const NoFsm&, u64*) { return 0; }  // hidden FSM = (): nothing absorbed
    __device__ static NoFsm fsm_in(const Inst&) { return NoFsm{}; }
    __device__ static NoFsm fsm_out(const Inst&) { return NoFsm{}; }
};

// One wave per instance: DPP row p (16 lanes) = part p of the closed form (observable input, observable output, FSM input, FSM output).
// Lane 0 of a row runs the part's encoder into its LDS slice (a run-time-indexed per-lane array would live in scratch memory, DESIGN.md
// 3.14), then the row absorbs the words eight at a time through the cooperative permutation (p2::Coop: a lane per state element, linear
// layers by DPP) — commit_variable_length_encodable_item: overwrite mode from (0, .., 0, n), the last chunk zero padded, the first four
// state words. Round 5: was one LANE per part running the lane-serial permutation — 55 dependent permutations of ~13 us for a Keccak FSM,
// 0.73 ms per launch on ONE wave, and with 96 blocks in flight these single-wave kernels are what the hardware queues spend their time
// on (docs/KERNELS.md 3.14); the row form is ~3 x shorter.
template <class T> struct CfLanes { static constexpr int value = 64; };
template <class T>
static __device__ __forceinline__ void k_closed_form_commitments(const VB& vb, const typename T::Inst* __restrict__ inst, size_t n, u64* __restrict__ compact) {
    const size_t i = vb.x;
    if (i >= n) return;
    __shared__ u64 sh_buf[4][T::MAXLEN];
    __shared__ int sh_m[4];
    const u32 lane = threadIdx.x & 63, g = lane & 15, part = lane >> 4;
    u64* cf = compact + COMPACT_FORM_LEN * i;
    if (g == 0) {
        int m;
        if (part == 0) {  // the observable input is the one of the block's first instance (postprocessing/mod.rs:358-364)
            size_t j = i;
            while (j > 0 && !inst[j].start_flag) j--;
            m = T::input(inst[j], sh_buf[0]);
        } else if (part == 1) {
            cf[0] = inst[i].start_flag ? 1 : 0;
            cf[1] = inst[i].completion_flag ? 1 : 0;
            m = T::output(inst[i], sh_buf[1]);
        } else {
            m = T::fsm(part == 2 ? T::fsm_in(inst[i]) : T::fsm_out(inst[i]), sh_buf[part]);
        }
        if (m > T::MAXLEN) __builtin_trap();  // an encoder outgrew its slice: fail loudly instead of hashing a neighbour's words
        sh_m[part] = m;
    }
    __syncthreads();
    const u32 m = (u32)sh_m[part];
    u32 most = 0;
    for (int p = 0; p < 4; p++) most = max(most, ((u32)sh_m[p] + 7) / 8);
    p2::Coop co;
    co.init((int)g);
    u64 x = g == 11 ? (u64)m : 0;  // apply_length_specialization
    const u32 perms = (m + 7) / 8;
    for (u32 q = 0; q < most; q++) {  // a uniform trip count: the DPP steps are wave-wide; a row that has run out keeps its state
        const bool on = q < perms;
        u64 in = x;
        if (g < 8) in = 8 * q + g < m ? sh_buf[part][8 * q + g] : 0;  // AbsorptionModeOverwrite
        const u64 y = co.permute(on ? in : 0);
        if (on) x = y;
    }
    const int at = part == 0 ? 2 : (part == 1 ? 6 : (part == 2 ? 10 : 14));
    if (g < 4) cf[at + g] = gl::canon(x);  // (an empty encoding: no permutation, the zero state's first words)
}

static __device__ __forceinline__ void k_encode_recursion(const VB& vb, u64 circuit_type, const u64* __restrict__ pi, size_t n,
                                                         u64* __restrict__ enc) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    enc[8 * i] = circuit_type;
    for (int k = 0; k < 4; k++) enc[8 * i + 1 + k] = pi[4 * i + k];
    for (int k = 5; k < 8; k++) enc[8 * i + k] = 0;
}

}  // namespace zkw
