// precompile_kernels.cuh — keccak256 / sha256 / ecrecover round-function witness builders.
//
// Reference functions replaced:
//   keccak256_decompose_into_per_circuit_witness  src/witness/individual_circuits/keccak256_round_function.rs:23-528
//   sha256_decompose_into_per_circuit_witness     src/witness/individual_circuits/sha256_round_function.rs:23-406
//   ecrecover_decompose_into_per_circuit_witness  src/witness/individual_circuits/ecrecover.rs:12-262
//
// The reference walks the rounds of all requests one after another, cutting an instance every `capacity` rounds.
// Requests are independent: rounds and memory queries per request follow from its ABI, so a prefix sum places
// every request in the global round / query sequence (k_precompile_counts), one lane per request then replays
// its own rounds and leaves a snapshot of the FSM at each instance boundary it crosses (k_precompile_walk), the
// memory queue is one Poseidon2 chain over the given queries (k_chain_full), and one lane per instance joins
// snapshots and queue states into the instance record (k_precompile_instances).
#pragma once
#include "decommitter_kernels.cuh"
#include "scan_kernels.cuh"

namespace zkw {

struct PrecompileAbi {  // PrecompileCallABI::from_u256(query.key), zkevm_opcode_defs (absent crate, see zkw_types.h)
    u32 input_memory_offset, input_memory_length, output_memory_offset, output_memory_length;
    u32 memory_page_to_read, memory_page_to_write;
    u64 precompile_interpreted_data;
};
__device__ __forceinline__ PrecompileAbi precompile_abi_in_log(const zkw_log_query& q) {
    PrecompileAbi a;
    a.input_memory_offset = q.key[0]; a.input_memory_length = q.key[1];
    a.output_memory_offset = q.key[2]; a.output_memory_length = q.key[3];
    a.memory_page_to_read = q.key[4]; a.memory_page_to_write = q.key[5];
    a.precompile_interpreted_data = (u64)q.key[6] | ((u64)q.key[7] << 32);
    return a;
}

// rounds / memory queries / reads one request contributes
__device__ __forceinline__ void precompile_request_shape(int kind, const zkw_log_query& q, u64& rounds, u64& queries, u64& reads) {
    const PrecompileAbi a = precompile_abi_in_log(q);
    if (kind == ZKW_PRECOMPILE_SHA256) {
        rounds = a.precompile_interpreted_data; reads = 2 * rounds; queries = reads + 1;
    } else if (kind == ZKW_PRECOMPILE_ECRECOVER) {
        rounds = 1; reads = 4; queries = 6;
    } else {
        const u64 off = a.input_memory_offset, len = a.input_memory_length;
        rounds = (len + 135) / 136 + (len % 136 == 0 ? 1 : 0);
        reads = len ? (off + len - 1) / 32 - off / 32 + 1 : 0;  // every touched word is read exactly once
        queries = reads + 1;
    }
}

// rounds / queries / reads of request i, for the tiled prefix sums (sum_prefix<3>, scan_kernels.cuh) that place every request in the
// global round / query / read sequences; *err |= 1 for a request without rounds
struct PrecompileShape {
    int kind;
    const zkw_log_query* requests;
    u32* err;
    __device__ void operator()(size_t i, u64 v[3]) const {
        precompile_request_shape(kind, requests[i], v[0], v[1], v[2]);
        if (v[0] == 0 || v[0] > (1ull << 32)) atomicOr(err, 1u);
    }
};

// the internal part of the FSM at an instance boundary + how far the global sequences have advanced
struct PrecompileSnap {
    zkw_precompile_fsm fsm;  // queue states are filled in by k_precompile_instances
    u64 popped, queries_done, reads_done;
};

struct PrecompileJob {
    int kind;
    const zkw_log_query* requests;
    const zkw_mem_query* mem_q;
    const u64 *round_off, *query_off, *read_off;
    PrecompileSnap* snaps;  // [n_instances]
    u32* violations;
    u64 n_requests, total_rounds;
    u32 capacity;
    zkw_keccak_round_record* keccak_rounds;  // keccak256 only, may be null: one record per round in the global round order
    zkw_sha256_round_record* sha256_rounds;  // sha256 only, may be null
    RoundOps* round_ops;                     // may be null: [total_rounds]
};

__device__ __forceinline__ void word_be_bytes(const u32* limbs, uint8_t out[32]) {  // U256::to_big_endian
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 l = limbs[7 - k];
        out[4 * k] = (uint8_t)(l >> 24); out[4 * k + 1] = (uint8_t)(l >> 16); out[4 * k + 2] = (uint8_t)(l >> 8); out[4 * k + 3] = (uint8_t)l;
    }
}

// one lane per request
static __device__ __forceinline__ void k_precompile_walk(const VB& vb, PrecompileJob job) {
    const u64 r = (u64)vb.x * blockDim.x + threadIdx.x;
    if (r >= job.n_requests) return;
    const zkw_log_query request = job.requests[r];
    PrecompileAbi abi = precompile_abi_in_log(request);
    const u64 g0 = job.round_off[r], num_rounds = job.round_off[r + 1] - g0;
    u64 qpos = job.query_off[r], reads = job.read_off[r];
    const u64 qend = job.query_off[r + 1];
    const bool is_last_request = r + 1 == job.n_requests;
    const int kind = job.kind;
    u32 sha[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    u64 kst[25];
    for (int i = 0; i < 25; i++) kst[i] = 0;
    // Per-lane arrays with run-time indices would live in scratch memory, which every HSA queue that ran the kernel then
    // holds for every wave slot of the chip: the byte buffer is a slice of LDS, the FSM snapshot is
    // written in place, bytes of a memory word are picked out of its limbs.
    __shared__ uint8_t sh_buf[64][ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE];
    uint8_t* buf = sh_buf[threadIdx.x];
    for (int i = 0; i < ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE; i++) buf[i] = 0;
    u32 filled = 0;
    const u32 padding_space = abi.input_memory_length % 136;
    const bool needs_extra_padding_round = kind == ZKW_PRECOMPILE_KECCAK256 && padding_space == 0;
    u64 rounds_left = num_rounds;
    int state = 1;  // 0 GetRequestFromQueue, 1 RunRoundFunction, 2 RunPaddingRound, 3 Finished
    if (kind == ZKW_PRECOMPILE_KECCAK256 && abi.input_memory_length == 0 && num_rounds == 1) state = 2;
    u32 bad = 0;
    for (u64 round = 0; round < num_rounds; round++) {
        const bool is_last_round = round + 1 == num_rounds;
        if (kind == ZKW_PRECOMPILE_SHA256) {
            u32 w[16];
            for (int k = 0; k < 2; k++) {
                const zkw_mem_query* q = job.mem_q + qpos;
                bad |= q->rw_flag;
#pragma unroll
                for (int j = 0; j < 8; j++) w[8 * k + j] = q->value[7 - j];
                qpos++; reads++;
                abi.input_memory_offset += 1;
            }
            // the cycle of the Sha256RoundFunction circuit: block as hashed (big-endian words), reset, state after
            u64* rec = job.sha256_rounds ? reinterpret_cast<u64*>(job.sha256_rounds + g0 + round) : nullptr;  // 104 = 8 * 13 bytes
            if (rec) {
#pragma unroll
                for (int m = 0; m < 8; m++) rec[m] = (u64)__builtin_bswap32(w[2 * m]) | ((u64)__builtin_bswap32(w[2 * m + 1]) << 32);
            }
            if (job.round_ops) job.round_ops[g0 + round] = RoundOps{(u32)r, (u32)(qpos - 2), is_last_round ? 3u : 2u, (round == 0 ? 1u : 0u) | (is_last_round ? 2u : 0u) | ((u32)(rounds_left - 1) << 8)};  // (bits 8..: rounds left after this one)
            sha256_compress(sha, w);  // expands the schedule in place
            if (rec) {
                rec[8] = (u64)(round == 0 ? 1u : 0u) | ((u64)sha[0] << 32);
                rec[9] = (u64)sha[1] | ((u64)sha[2] << 32);
                rec[10] = (u64)sha[3] | ((u64)sha[4] << 32);
                rec[11] = (u64)sha[5] | ((u64)sha[6] << 32);
                rec[12] = (u64)sha[7];
            }
            rounds_left--;
        } else if (kind == ZKW_PRECOMPILE_ECRECOVER) {
            for (int k = 0; k < 4; k++) bad |= job.mem_q[qpos + k].rw_flag;
            for (int k = 4; k < 6; k++) bad |= !job.mem_q[qpos + k].rw_flag;
            if (job.round_ops) job.round_ops[g0 + round] = RoundOps{(u32)r, (u32)qpos, 6u, 1u};  // the request's one round pops it and pushes its six queries
            qpos += 6; reads += 4;
        } else {
            const bool paddings_round = needs_extra_padding_round && is_last_round;
            const u64 qpos_before = qpos;
            for (int slot = 0; slot < ZKW_KECCAK_MEMORY_READS_PER_CYCLE; slot++) {
                const u32 memory_index = abi.input_memory_offset / 32, unalignment = abi.input_memory_offset % 32;
                const u32 at_most = 32 - unalignment;
                const u32 meaningful = abi.input_memory_length >= at_most ? at_most : abi.input_memory_length;
                const bool should_read = meaningful != 0 && filled + meaningful <= ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE;
                if (!should_read) continue;
                if (paddings_round || qpos + 1 >= qend) { bad = 1; break; }  // the last query of a request is its write
                const zkw_mem_query* q = job.mem_q + qpos;
                bad |= q->rw_flag | (q->index != memory_index);
                abi.input_memory_offset += meaningful;
                abi.input_memory_length -= meaningful;
                qpos++; reads++;
                for (u32 j = 0; j < meaningful; j++) {  // byte x of U256::to_big_endian
                    const u32 x = unalignment + j;
                    buf[filled + j] = (uint8_t)(q->value[7 - (x >> 2)] >> (8 * (3 - (x & 3))));
                }
                filled += meaningful;
            }
            if (job.round_ops) job.round_ops[g0 + round] = RoundOps{(u32)r, (u32)qpos_before, (u32)(qpos - qpos_before) + (is_last_round ? 1u : 0u), (round == 0 ? 1u : 0u) | (is_last_round ? 2u : 0u)};
            // consume::<136>, padding applied to the copy
            u64 lanes[17];
#pragma unroll
            for (int k = 0; k < 17; k++) {
                u64 l = 0;
                for (int b = 0; b < 8; b++) l |= (u64)buf[8 * k + b] << (8 * b);
                lanes[k] = l;
            }
            filled = filled < 136 ? 0 : filled - 136;
            for (int i = 0; i < ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE; i++)
                buf[i] = i + 136 < ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE ? buf[i + 136] : 0;
            if (is_last_round) {
                const u32 at = needs_extra_padding_round ? 0 : padding_space;
                // block[at] = 0x01 (or 0x81 when at == 135), block[135] = 0x80: the bytes being replaced are zero
                // because the buffer is zero beyond `filled`
#pragma unroll
                for (int k = 0; k < 17; k++) lanes[k] |= k == (int)(at >> 3) ? (u64)0x01 << (8 * (at & 7)) : 0;
                lanes[16] |= (u64)0x80 << 56;
            }
#pragma unroll
            for (int k = 0; k < 17; k++) kst[k] ^= lanes[k];
            keccak_f1600(kst);
            if (job.keccak_rounds) {  // the cycle of the Keccak256RoundFunction circuit: padded block, reset, state after
                zkw_keccak_round_record* rec = job.keccak_rounds + g0 + round;
                u64* blk64 = reinterpret_cast<u64*>(rec->block);  // records are 344 = 8 * 43 bytes: 8-byte aligned
#pragma unroll
                for (int k = 0; k < 17; k++) blk64[k] = lanes[k];
                blk64[17] = round == 0 ? 1 : 0;                    // reset + 7 bytes of padding
                u64* st64 = reinterpret_cast<u64*>(rec->state_after);
#pragma unroll
                for (int k = 0; k < 25; k++) st64[k] = kst[k];
            }
            if (state == 1 && needs_extra_padding_round && round + 2 == num_rounds) state = 2;
        }
        if (is_last_round) {
            if (kind != ZKW_PRECOMPILE_ECRECOVER) { bad |= !job.mem_q[qpos].rw_flag; qpos++; }
            state = is_last_request ? 3 : 0;
        }
        const u64 g = g0 + round;
        const bool cut = (g + 1) % job.capacity == 0 || g + 1 == job.total_rounds;
        if (!cut) continue;
        const bool early_termination = (g + 1) % job.capacity != 0;
        PrecompileSnap* sn = job.snaps + g / job.capacity;
        zkw_precompile_fsm& f = sn->fsm;
        memset(&f, 0, sizeof f);
        if (kind != ZKW_PRECOMPILE_ECRECOVER) {
            f.completed = state == 3; f.read_words_for_round = state == 1; f.read_precompile_call = state == 0;
            f.padding_round = state == 2;
            f.timestamp_to_use_for_read = request.timestamp;
            f.timestamp_to_use_for_write = request.timestamp + 1;
            f.input_page = abi.memory_page_to_read; f.input_offset = abi.input_memory_offset;
            f.output_page = abi.memory_page_to_write; f.output_offset = abi.output_memory_offset;
            if (kind == ZKW_PRECOMPILE_SHA256) {
                f.num_rounds = (u32)rounds_left;
                if (early_termination) {  // Sha256 over one zero block, sha256_round_function.rs:283-296
                    u32 e[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
                    u32 z[16];
                    for (int j = 0; j < 16; j++) z[j] = 0;
                    sha256_compress(e, z);
                    for (int j = 0; j < 8; j++) f.sha256_inner_state[j] = e[j];
                } else {
                    for (int j = 0; j < 8; j++) f.sha256_inner_state[j] = sha[j];
                }
            } else {
                f.input_length = abi.input_memory_length;
                f.needs_full_padding_round = needs_extra_padding_round;
                f.buffer_filled = filled;
                u64 e[25];
                if (early_termination) {  // keccak256_round_function.rs:376-394
                    for (int i = 0; i < 25; i++) e[i] = 0;
                    keccak_f1600(e);
                } else {
                    for (int i = 0; i < 25; i++) e[i] = kst[i];
                    for (int i = 0; i < ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE; i++) f.buffer_bytes[i] = buf[i];
                }
                for (int idx = 0; idx < 25; idx++) {  // encode_kecca256_inner_state :530-541
                    const int i = idx % 5, j = idx / 5;
                    for (int b = 0; b < 8; b++) f.keccak_internal_state[(i * 5 + j) * 8 + b] = (uint8_t)(e[idx] >> (8 * b));
                }
            }
        }
        sn->popped = r + 1;
        sn->queries_done = qpos;
        sn->reads_done = reads;
    }
    if (bad || qpos != qend) atomicAdd(job.violations, 1u);
}

struct PrecompileBlock {
    int kind;
    const PrecompileSnap* snaps;
    const u64* req_tails;  // [n_requests][4]
    const u64* mem_tails;  // [n_queries][12]
    zkw_precompile_instance* instances;
    zkw_queue_state12 mem_in;
    u64 n_requests, total_rounds, n_instances;
    u32 capacity;
};

static __device__ __forceinline__ void k_precompile_instances(const VB& vb, const PrecompileBlock* __restrict__ blk) {
    const PrecompileBlock& b = *blk;
    const u64 idx = (u64)vb.x * blockDim.x + threadIdx.x;
    if (idx >= b.n_instances) return;
    const u64* req_final = b.n_requests ? b.req_tails + 4 * (b.n_requests - 1) : nullptr;
    auto queues = [&](zkw_precompile_fsm& f, u64 popped, u64 queries_done) {
        qs4(f.log_queue_state, popped ? b.req_tails + 4 * (popped - 1) : nullptr, req_final, (u32)(b.n_requests - popped));
        qs12(f.memory_queue_state, b.mem_in.head, queries_done ? b.mem_tails + 12 * (queries_done - 1) : b.mem_in.tail,
             b.mem_in.length + (u32)queries_done);
    };
    zkw_precompile_instance& w = b.instances[idx];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    if (b.n_requests == 0) {  // the dummy instance (keccak :88-157, sha256 :82-150, ecrecover :60-103)
        w.start_flag = w.completion_flag = 1;
        w.initial_memory_queue_state = b.mem_in;
        w.final_memory_state = b.mem_in;
        queues(w.hidden_fsm_input, 0, 0);
        queues(w.hidden_fsm_output, 0, 0);
        if (b.kind != ZKW_PRECOMPILE_ECRECOVER) {
            w.hidden_fsm_input.read_precompile_call = 1;
            w.hidden_fsm_output.completed = 1;
            if (b.kind == ZKW_PRECOMPILE_SHA256) {
                u32 e[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
                u32 z[16];
                for (int j = 0; j < 16; j++) z[j] = 0;
                sha256_compress(e, z);
                for (int j = 0; j < 8; j++) w.hidden_fsm_output.sha256_inner_state[j] = e[j];
            } else {
                u64 e[25];
                for (int i = 0; i < 25; i++) e[i] = 0;
                keccak_f1600(e);
                for (int id = 0; id < 25; id++)
                    for (int by = 0; by < 8; by++)
                        w.hidden_fsm_output.keccak_internal_state[((id % 5) * 5 + id / 5) * 8 + by] = (uint8_t)(e[id] >> (8 * by));
            }
        }
        return;
    }
    const PrecompileSnap& out = b.snaps[idx];
    u64 p0 = 0, q0 = 0, r0 = 0;
    w.start_flag = idx == 0;
    if (idx == 0) {
        if (b.kind != ZKW_PRECOMPILE_ECRECOVER) w.hidden_fsm_input.read_precompile_call = 1;
        qs4(w.initial_log_queue_state, nullptr, req_final, (u32)b.n_requests);
        w.initial_memory_queue_state = b.mem_in;
    } else {
        const PrecompileSnap& in = b.snaps[idx - 1];
        w.hidden_fsm_input = in.fsm;
        p0 = in.popped; q0 = in.queries_done; r0 = in.reads_done;
    }
    queues(w.hidden_fsm_input, p0, q0);
    w.hidden_fsm_output = out.fsm;
    queues(w.hidden_fsm_output, out.popped, out.queries_done);
    w.first_request = p0; w.num_requests = out.popped - p0;
    w.first_read = r0; w.num_reads = out.reads_done - r0;
    w.first_round = idx * b.capacity;
    w.num_rounds = (idx + 1 == b.n_instances ? b.total_rounds : (idx + 1) * b.capacity) - w.first_round;
    if (idx + 1 == b.n_instances) {
        w.completion_flag = 1;
        w.final_memory_state = w.hidden_fsm_output.memory_queue_state;
    }
}

}  // namespace zkw
