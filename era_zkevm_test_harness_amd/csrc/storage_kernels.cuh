// storage_kernels.cuh — StorageSorter witness builder on gfx950.
// Reference: sort_storage_access_queries (circuit_sequencer_api/src/sort_storage_access.rs:19-260) and
//            compute_storage_dedup_and_sort (src/witness/individual_circuits/storage_sort_dedup.rs:12-703).
// The reference keeps a per-cell history (stack of pending writes) while walking the sorted log; here the
// per-cell registers are differences of global prefix sums taken at the cell's first item:
//   depth(t)  = D(t) - D(start-1),   D = prefix sum of (+1 forward write, -1 rollback, 0 read)
//   has_read_at_depth_zero(t) = R(t) - R(start-1) > 0,   R = prefix count of reads seen at depth 0
//   base value = read_value of the cell's first item, current value = a function of item t alone
// so every sorted position — and therefore every chunk boundary — is computed independently.
#pragma once
#include "demux_kernels.cuh"

namespace zkw {

static __device__ __forceinline__ void k_storage_sort_keys(const VB& vb, const zkw_log_query* __restrict__ q, size_t n, u64* __restrict__ k0, u64* __restrict__ k1,
                                    u64* __restrict__ k2, u64* __restrict__ k3, u64* __restrict__ a0, u64* __restrict__ a1,
                                    u32* __restrict__ a2, u32* __restrict__ iota) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const zkw_log_query* d = q + i;
    k0[i] = ((u64)d->key[1] << 32) | d->key[0];
    k1[i] = ((u64)d->key[3] << 32) | d->key[2];
    k2[i] = ((u64)d->key[5] << 32) | d->key[4];
    k3[i] = ((u64)d->key[7] << 32) | d->key[6];
    a0[i] = ((u64)d->address[1] << 32) | d->address[0];
    a1[i] = ((u64)d->address[3] << 32) | d->address[2];
    a2[i] = d->address[4];
    iota[i] = (u32)i;
}

// sorted_q[i] = q[perm[i]], encoded with extended_timestamp = perm[i] (its position in the unsorted queue)
static __device__ __forceinline__ void k_storage_gather_encode(const VB& vb, const zkw_log_query* __restrict__ q, const u32* __restrict__ perm,
                                                               size_t n, zkw_log_query* __restrict__ sorted_q,
                                                               u32* __restrict__ sorted_ext, u64* __restrict__ sorted_enc) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 src = perm[i];
    zkw_log_query m;
    load_log(q + src, m);
    store_log(sorted_q + i, m);
    sorted_ext[i] = src;
    u64 e[20];
    encode_log_query(m, true, src, e);
    store_enc20(sorted_enc + 20 * i, e);
}

struct StorageScan {
    int* D;      // [n] inclusive prefix of depth deltas
    u32* S;      // [n] index of the first item of the cell item t belongs to
    u32* R;      // [n] inclusive prefix count of reads seen at depth 0
    u32* E;      // [n] inclusive prefix count of emitting cells (counted at their last item)
};

__device__ __forceinline__ bool same_cell_dev(const zkw_log_query* a, const zkw_log_query* b) {
    return a->shard_id == b->shard_id && same_words(a->address, b->address, 5) && same_words(a->key, b->key, 8);
}

__device__ __forceinline__ void cell_current_value(const zkw_log_query& m, u32 out[8]) {
    // forward write -> written value; rollback -> the value before the write; read -> the value read
    const bool fwd_write = m.rw_flag && !m.rollback;
    for (int k = 0; k < 8; k++) out[k] = fwd_write ? m.written_value[k] : m.read_value[k];
}

// ---- the per-cell registers and the deduplicated queue (sort_storage_access.rs:64-253) as TILED prefix passes (scan_kernels.cuh) —
// the first version was one workgroup sweeping the queue three times. D = inclusive sum of the depth deltas (sum_prefix), S = first item
// of an item's cell (flag_prefix over the cell starts numbers the cells, the starts scatter their index), R = reads seen at depth 0,
// E = emitting cells (flag_prefix each), then one lane per item emits.
struct StoDelta {
    const zkw_log_query* q;
    __device__ void operator()(size_t i, u64 v[1]) const { v[0] = q[i].rw_flag ? (q[i].rollback ? ~0ull : 1ull) : 0ull; }
};
struct StoIsStart {
    const zkw_log_query* q;
    __device__ u32 operator()(size_t i) const { return (i == 0 || !same_cell_dev(q + i, q + i - 1)) ? 1u : 0u; }
};
static __device__ __forceinline__ void k_storage_first(const VB& vb, const u32* __restrict__ start_prefix, size_t n, u32* __restrict__ first) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i < n && start_prefix[i + 1] != start_prefix[i]) first[start_prefix[i + 1] - 1] = (u32)i;
}
static __device__ __forceinline__ void k_storage_ds(const VB& vb, const zkw_log_query* __restrict__ sorted_q, size_t n, const u64* __restrict__ delta_prefix,
                                                           const u32* __restrict__ start_prefix, const u32* __restrict__ first, StorageScan sc, u32* __restrict__ viol) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const zkw_log_query* m = sorted_q + i;
    const int delta = m->rw_flag ? (m->rollback ? -1 : 1) : 0;
    sc.D[i] = (int)(long long)delta_prefix[i] + delta;
    sc.S[i] = first[start_prefix[i + 1] - 1];
    const bool is_start = start_prefix[i + 1] != start_prefix[i];
    if (is_start && m->rw_flag && m->rollback) atomicAdd(viol, 1u);  // sort_storage_access.rs:91 / storage_sort_dedup.rs:342
    if (m->shard_id != 0) atomicAdd(viol, 1u);
}
struct StoReadAtZero {  // a read at depth zero; a negative depth = a rollback without a pending write (changes_stack.pop().unwrap())
    const zkw_log_query* q;
    StorageScan sc;
    u32* viol;
    __device__ u32 operator()(size_t i) const {
        const u32 s = sc.S[i];
        const int depth = sc.D[i] - (s ? sc.D[s - 1] : 0);
        if (depth < 0) atomicAdd(viol, 1u);
        return (!q[i].rw_flag && depth == 0) ? 1u : 0u;
    }
};
__device__ __forceinline__ bool storage_emits(const StorageScan& sc, const u32* __restrict__ read_prefix, size_t n, size_t i, u32& s, int& depth) {
    s = sc.S[i];
    depth = sc.D[i] - (s ? sc.D[s - 1] : 0);
    const bool has = read_prefix[i + 1] - read_prefix[s] > 0;  // (inclusive counts: R[i] - R[s - 1])
    const bool cell_last = i + 1 == n || sc.S[i + 1] != s;
    return cell_last && (depth > 0 || has);
}
struct StoEmits {
    StorageScan sc;
    const u32* read_prefix;
    size_t n;
    __device__ u32 operator()(size_t i) const { u32 s; int depth; return storage_emits(sc, read_prefix, n, i, s, depth) ? 1u : 0u; }
};
static __device__ __forceinline__ void k_storage_emit(const VB& vb, const zkw_log_query* __restrict__ sorted_q, size_t n, StorageScan sc, const u32* __restrict__ read_prefix,
                                                             const u32* __restrict__ emit_prefix, zkw_log_query* __restrict__ result_q, u64* __restrict__ result_enc,
                                                             u32* __restrict__ totals /* [2]: n_result, violations (accumulated by the passes before) */) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i == 0) totals[0] = emit_prefix[n];
    if (i >= n) return;
    sc.R[i] = read_prefix[i + 1];
    const u32 e = emit_prefix[i + 1];
    sc.E[i] = e;
    if (e == emit_prefix[i]) return;
    u32 s;
    int depth;
    (void)storage_emits(sc, read_prefix, n, i, s, depth);
    zkw_log_query me;
    load_log(sorted_q + i, me);
    u32 cur[8];
    cell_current_value(me, cur);
    const zkw_log_query* first = sorted_q + s;
    bool eq = true;
    for (int k = 0; k < 8; k++) eq &= cur[k] == first->read_value[k];
    if (depth == 0 && !eq) atomicAdd(&totals[1], 1u);  // sort_storage_access.rs:198-203
    zkw_log_query r;
    memset(&r, 0, sizeof r);
    r.shard_id = me.shard_id;
    for (int k = 0; k < 5; k++) r.address[k] = me.address[k];
    for (int k = 0; k < 8; k++) { r.key[k] = me.key[k]; r.read_value[k] = first->read_value[k]; r.written_value[k] = cur[k]; }
    r.rw_flag = eq ? 0 : 1;
    store_log(result_q + (e - 1), r);
    u64 enc[20];
    encode_log_query(r, false, 0, enc);
    store_enc20(result_enc + 20 * (size_t)(e - 1), enc);
}

struct StorageBlock {
    const zkw_log_query* sorted_q;
    const u32* sorted_ext;
    const u64* unsorted_new_tails;
    const u64* sorted_new_tails;
    const u64* result_new_tails;
    const u64* lhs_z;
    const u64* rhs_z;
    StorageScan sc;
    zkw_storage_sorter_instance* instances;
    u64 n;
    u32 capacity;
};

static __device__ __forceinline__ void k_storage_instances(const VB& vb, const StorageBlock* __restrict__ blk) {
    const StorageBlock b = *blk;
    const u64 n = b.n, n_inst = (n + b.capacity - 1) / b.capacity;
    const u64 idx = (u64)vb.x * blockDim.x + threadIdx.x;
    if (idx >= n_inst) return;
    zkw_storage_sorter_instance& w = b.instances[idx];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    const u64 lo = idx * b.capacity, hi = lo + b.capacity < n ? lo + b.capacity : n;
    const bool is_last = idx == n_inst - 1;
    w.start_flag = idx == 0;
    w.completion_flag = is_last;
    w.first_item = lo;
    w.num_items = hi - lo;
    const u64* u_final = b.unsorted_new_tails + 4 * (n - 1);
    const u64* s_final = b.sorted_new_tails + 4 * (n - 1);
    qs4(w.unsorted_log_queue_state, nullptr, u_final, (u32)n);
    qs4(w.intermediate_sorted_queue_state, nullptr, s_final, (u32)n);
    // a cell's record is pushed when the first item of the NEXT cell is processed (the last cell's when the
    // last item ever is processed)
    auto result_at = [&](u64 end, zkw_queue_state4& s) {
        const u32 c = end == n ? b.sc.E[n - 1] : (end >= 2 ? b.sc.E[end - 2] : 0);
        qs4(s, nullptr, c ? b.result_new_tails + 4 * (size_t)(c - 1) : nullptr, c);
    };
    auto fill = [&](zkw_storage_sorter_fsm& f, u64 end /* > 0 */, u32 chunks_done) {
        const u64 l = end - 1;
        const zkw_log_query* q = b.sorted_q + l;
        for (int r = 0; r < 2; r++) { f.lhs_accumulator[r] = b.lhs_z[r * n + l]; f.rhs_accumulator[r] = b.rhs_z[r * n + l]; }
        qs4(f.current_unsorted_queue_state, b.unsorted_new_tails + 4 * l, u_final, (u32)(n - end));
        qs4(f.current_intermediate_sorted_queue_state, b.sorted_new_tails + 4 * l, s_final, (u32)(n - end));
        result_at(end, f.current_final_sorted_queue_state);
        f.cycle_idx = chunks_done * b.capacity;
        for (int k = 0; k < 8; k++) { f.previous_packed_key[k] = q->key[k]; f.previous_key[k] = q->key[k]; }
        for (int k = 0; k < 5; k++) { f.previous_packed_key[8 + k] = q->address[k]; f.previous_address[k] = q->address[k]; }
        f.previous_timestamp = b.sorted_ext[l];
        const u32 s = b.sc.S[l];
        f.this_cell_current_depth = (u32)(b.sc.D[l] - (s ? b.sc.D[s - 1] : 0));
        f.this_cell_has_explicit_read_and_rollback_depth_zero = (b.sc.R[l] - (s ? b.sc.R[s - 1] : 0)) > 0 ? 1 : 0;
        for (int k = 0; k < 8; k++) f.this_cell_base_value[k] = b.sorted_q[s].read_value[k];
        cell_current_value(*q, f.this_cell_current_value);
    };
    if (idx == 0) {
        for (int r = 0; r < 2; r++) { w.hidden_fsm_input.lhs_accumulator[r] = 1; w.hidden_fsm_input.rhs_accumulator[r] = 1; }
    } else {
        fill(w.hidden_fsm_input, lo, (u32)idx);
    }
    fill(w.hidden_fsm_output, hi, (u32)idx + 1);
    zkw_storage_sorter_fsm& fo = w.hidden_fsm_output;
    if ((hi - lo) % b.capacity != 0) {  // storage_sort_dedup.rs:613-636
        for (int k = 0; k < 13; k++) fo.previous_packed_key[k] = 0;
        for (int k = 0; k < 8; k++) fo.previous_key[k] = 0;
        for (int k = 0; k < 5; k++) fo.previous_address[k] = 0;
        fo.previous_timestamp = 0;
        fo.this_cell_has_explicit_read_and_rollback_depth_zero = 0;
    } else if (is_last) {
        fo.this_cell_has_explicit_read_and_rollback_depth_zero = 0;
    }
    if (is_last) result_at(n, w.final_sorted_queue_state);
}

}  // namespace zkw
