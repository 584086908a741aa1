// events_kernels.cuh — Events / L1-messages sorter witness builder on gfx950.
// Reference: compute_events_dedup_and_sort + sort_and_dedup_events_log,
//            src/witness/individual_circuits/events_sort_dedup.rs:16-580.
// The reference replays the circuit item by item (a stack for the dedup, an iterator for the pushes); here
// "kept" (forward event not followed by its rollback) is a local predicate on neighbours, the result queue
// is the compaction of the kept items, and the number of pushes a chunk has made is a prefix count.
#pragma once
#include "scan_kernels.cuh"
#include "log_kernels.cuh"

namespace zkw {

static __device__ __forceinline__ void k_events_sort_keys(const VB& vb, const zkw_log_query* __restrict__ q, size_t n, u64* __restrict__ key,
                                   u32* __restrict__ iota) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = ((u64)q[i].timestamp << 1) | (q[i].rollback ? 1 : 0);  // rollback sorts after its forward twin
    iota[i] = (u32)i;
}

__device__ __forceinline__ void load_log(const zkw_log_query* src, zkw_log_query& dst) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(&dst);
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = s[k];
}
__device__ __forceinline__ void store_log(zkw_log_query* dst, const zkw_log_query& src) {
    const uint4* s = reinterpret_cast<const uint4*>(&src);
    uint4* d = reinterpret_cast<uint4*>(dst);
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = s[k];
}
__device__ __forceinline__ void store_enc20(u64* dst, const u64 e[20]) {
    ulonglong2* o = reinterpret_cast<ulonglong2*>(dst);
#pragma unroll
    for (int k = 0; k < 10; k++) o[k] = make_ulonglong2(e[2 * k], e[2 * k + 1]);
}

static __device__ __forceinline__ void k_log_gather_encode(const VB& vb, const zkw_log_query* __restrict__ q, const u32* __restrict__ perm,
                                                           size_t n, zkw_log_query* __restrict__ sorted_q,
                                                           u64* __restrict__ sorted_enc) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zkw_log_query m;
    load_log(q + perm[i], m);
    store_log(sorted_q + i, m);
    u64 e[20];
    encode_log_query(m, false, 0, e);
    store_enc20(sorted_enc + 20 * i, e);
}

__device__ __forceinline__ bool same_words(const u32* a, const u32* b, int n) {
    bool eq = true;
    for (int k = 0; k < n; k++) eq &= a[k] == b[k];
    return eq;
}

// kept = a forward record that no rollback follows (:541-553): the flag of flag_prefix (scan_kernels.cuh)
struct EventsKeptFlag {
    const zkw_log_query* sorted_q;
    size_t n;
    __device__ u32 operator()(size_t i) const { return (!sorted_q[i].rollback && (i + 1 == n || sorted_q[i + 1].timestamp != sorted_q[i].timestamp)) ? 1u : 0u; }
};

// every record on its own, given the tiled prefix count of the kept flags (prefix[k] = kept among [0, k)): the reference's asserts
// (:344-356, :512-533), the inclusive count, and the compaction of the kept items into normalised result records (:541-553) with
// their encodings. totals[1] (violations) is zeroed by the caller; the last record's thread writes totals[0] = n_result.
static __device__ __forceinline__ void k_events_dedup(const VB& vb, const zkw_log_query* __restrict__ sorted_q, size_t n, const u32* __restrict__ prefix,
                                                      u32* __restrict__ kept_count /* [n] inclusive */,
                                                      zkw_log_query* __restrict__ result_q, u64* __restrict__ result_enc,
                                                      u32* __restrict__ totals /* [2]: n_result, violations */) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zkw_log_query me;
    load_log(sorted_q + i, me);
    bool bad = me.shard_id != 0;
    if (i == 0) {
        bad |= me.rollback != 0;
    } else {
        const zkw_log_query* p = sorted_q + i - 1;
        bad |= me.rw_flag == 0;
        if (p->timestamp == me.timestamp) {
            bad |= me.rollback == 0 || p->rollback != 0 || p->rw_flag == 0 ||
                   p->tx_number_in_block != me.tx_number_in_block || p->is_service != me.is_service ||
                   !same_words(p->address, me.address, 5) || !same_words(p->key, me.key, 8) ||
                   !same_words(p->written_value, me.written_value, 8);
        } else {
            bad |= me.rollback != 0;
        }
    }
    if (bad) atomicAdd(&totals[1], 1u);
    const u32 cnt = prefix[i + 1];
    kept_count[i] = cnt;
    if (i + 1 == n) totals[0] = cnt;
    if (cnt != prefix[i]) {  // kept
        zkw_log_query r;
        memset(&r, 0, sizeof r);
        r.tx_number_in_block = me.tx_number_in_block;
        r.shard_id = me.shard_id;
        for (int k = 0; k < 5; k++) r.address[k] = me.address[k];
        for (int k = 0; k < 8; k++) { r.key[k] = me.key[k]; r.written_value[k] = me.written_value[k]; }
        r.is_service = me.is_service;
        store_log(result_q + (cnt - 1), r);
        u64 e[20];
        encode_log_query(r, false, 0, e);
        store_enc20(result_enc + 20 * (size_t)(cnt - 1), e);
    }
}

struct EventsBlock {
    const zkw_log_query* sorted_q;
    const u64* unsorted_new_tails;  // [n][4]
    const u64* sorted_new_tails;
    const u64* result_new_tails;    // [n_result][4]
    const u64* lhs_z;               // [2][n]
    const u64* rhs_z;
    const u32* kept_count;          // inclusive
    zkw_events_sorter_instance* instances;
    zkw_queue_state4 result_in;
    u64 n;
    u32 capacity;
};

__device__ __forceinline__ void qs4(zkw_queue_state4& s, const u64* head, const u64* tail, u32 len) {
    for (int k = 0; k < 4; k++) { s.head[k] = head ? head[k] : 0; s.tail[k] = tail ? tail[k] : 0; }
    s.length = len;
    s._pad = 0;
}

static __device__ __forceinline__ void k_events_instances(const VB& vb, const EventsBlock* __restrict__ blk) {
    const EventsBlock& b = *blk;
    const u64 n = b.n, n_inst = (n + b.capacity - 1) / b.capacity;
    const u64 idx = (u64)vb.x * blockDim.x + threadIdx.x;
    if (idx >= n_inst) return;
    zkw_events_sorter_instance& w = b.instances[idx];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    const u64 lo = idx * b.capacity, hi = lo + b.capacity < n ? lo + b.capacity : n;
    w.start_flag = idx == 0;
    w.completion_flag = idx == n_inst - 1;
    w.first_item = lo;
    w.num_items = hi - lo;
    const u64* u_final = b.unsorted_new_tails + 4 * (n - 1);
    const u64* s_final = b.sorted_new_tails + 4 * (n - 1);
    qs4(w.initial_log_queue_state, nullptr, u_final, (u32)n);
    qs4(w.intermediate_sorted_queue_state, nullptr, s_final, (u32)n);
    // result-queue state after the chunks covering items [0, end) have been processed: a kept item is pushed
    // when its successor is processed, the last item ever when it is processed itself
    auto result_at = [&](u64 end, zkw_queue_state4& s) {
        const u32 c = end == n ? b.kept_count[n - 1] : (end >= 2 ? b.kept_count[end - 2] : 0);
        qs4(s, b.result_in.head, c ? b.result_new_tails + 4 * (size_t)(c - 1) : b.result_in.tail, b.result_in.length + c);
    };
    auto fill = [&](zkw_events_sorter_fsm& f, u64 end /* > 0 */) {
        const u64 l = end - 1;
        for (int r = 0; r < 2; r++) { f.lhs_accumulator[r] = b.lhs_z[r * n + l]; f.rhs_accumulator[r] = b.rhs_z[r * n + l]; }
        qs4(f.initial_unsorted_queue_state, b.unsorted_new_tails + 4 * l, u_final, (u32)(n - end));
        qs4(f.intermediate_sorted_queue_state, b.sorted_new_tails + 4 * l, s_final, (u32)(n - end));
        result_at(end, f.final_result_queue_state);
        f.previous_key = b.sorted_q[l].timestamp;
        f.previous_item = b.sorted_q[l];
    };
    if (idx == 0) {
        for (int r = 0; r < 2; r++) { w.hidden_fsm_input.lhs_accumulator[r] = 1; w.hidden_fsm_input.rhs_accumulator[r] = 1; }
        w.hidden_fsm_input.final_result_queue_state = b.result_in;
    } else {
        fill(w.hidden_fsm_input, lo);
    }
    fill(w.hidden_fsm_output, hi);
    if ((hi - lo) % b.capacity != 0) {  // padding reset, :469-479
        w.hidden_fsm_output.previous_key = 0;
        memset(&w.hidden_fsm_output.previous_item, 0, sizeof(zkw_log_query));
    }
    if (idx == n_inst - 1) result_at(n, w.final_queue_state);
}

}  // namespace zkw
