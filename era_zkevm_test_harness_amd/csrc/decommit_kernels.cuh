// decommit_kernels.cuh — CodeDecommittmentsSorter witness builder on gfx950.
// Reference: compute_decommitts_sorter_circuit_snapshots,
//            src/witness/individual_circuits/sort_decommit_requests.rs:20-420.
// The reference walks the sorted requests sequentially carrying counters; here every per-chunk snapshot is
// expressed through prefix counts of the `is_fresh` flag, so chunks are filled independently.
#pragma once
#include "scan_kernels.cuh"
#include "log_kernels.cuh"

namespace zkw {

// sort keys: timestamp and the four 64-bit halves of the hash (least significant first)
static __device__ __forceinline__ void k_decommit_sort_keys(const VB& vb, const zkw_decommit_query* __restrict__ q, size_t n, u32* __restrict__ ts,
                                     u64* __restrict__ h0, u64* __restrict__ h1, u64* __restrict__ h2,
                                     u64* __restrict__ h3, u32* __restrict__ iota) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const zkw_decommit_query* d = q + i;
    ts[i] = d->timestamp;
    h0[i] = ((u64)d->hash[1] << 32) | d->hash[0];
    h1[i] = ((u64)d->hash[3] << 32) | d->hash[2];
    h2[i] = ((u64)d->hash[5] << 32) | d->hash[4];
    h3[i] = ((u64)d->hash[7] << 32) | d->hash[6];
    iota[i] = (u32)i;
}

static __device__ __forceinline__ void k_decommit_gather_encode(const VB& vb, const zkw_decommit_query* __restrict__ q,
                                                                const u32* __restrict__ perm, size_t n,
                                                                zkw_decommit_query* __restrict__ sorted_q,
                                                                u64* __restrict__ sorted_enc) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* src = reinterpret_cast<const uint4*>(q + perm[i]);
    uint4 w0 = src[0], w1 = src[1], w2 = src[2];
    uint4* dq = reinterpret_cast<uint4*>(sorted_q + i);
    dq[0] = w0; dq[1] = w1; dq[2] = w2;
    zkw_decommit_query m;
    uint4* dm = reinterpret_cast<uint4*>(&m);
    dm[0] = w0; dm[1] = w1; dm[2] = w2;
    u64 e[8];
    encode_decommit_query(m, e);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(sorted_enc + 8 * i);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = make_ulonglong2(e[2 * k], e[2 * k + 1]);
}

// is_fresh of a sorted request: the flag of flag_prefix (scan_kernels.cuh)
struct DecommitFreshFlag {
    const zkw_decommit_query* sorted_q;
    __device__ u32 operator()(size_t i) const { return sorted_q[i].is_fresh ? 1u : 0u; }
};

// every request on its own, given the tiled prefix count of is_fresh (prefix[k] = fresh among [0, k)): the inclusive count, the
// reference's ordering self-check (:99-114), the compaction of the fresh requests (= the deduplicated queue, :121-140) and the
// position of every fresh request (fresh_pos[k] = index of the k-th). totals[1] is zeroed by the caller.
static __device__ __forceinline__ void k_decommit_dedup(const VB& vb, const zkw_decommit_query* __restrict__ sorted_q, const u64* __restrict__ sorted_enc, size_t n,
                                                        const u32* __restrict__ prefix, u32* __restrict__ fresh_count /* [n] inclusive */,
                                                        u32* __restrict__ fresh_pos /* [n] */, zkw_decommit_query* __restrict__ dedup_q,
                                                        u64* __restrict__ dedup_enc, u32* __restrict__ totals /* [2]: n_dedup, violations */) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const zkw_decommit_query* d = sorted_q + i;
    if (i > 0) {
        const zkw_decommit_query* p = d - 1;
        bool same = true;
        for (int k = 0; k < 8; k++) same &= d->hash[k] == p->hash[k];
        if (same && (d->memory_page != p->memory_page || !(d->timestamp > p->timestamp))) atomicAdd(&totals[1], 1u);
    }
    const u32 cnt = prefix[i + 1];
    fresh_count[i] = cnt;
    if (i + 1 == n) totals[0] = cnt;
    if (cnt != prefix[i]) {  // fresh
        const size_t dst = cnt - 1;
        fresh_pos[dst] = (u32)i;
        const uint4* s = reinterpret_cast<const uint4*>(sorted_q + i);
        uint4* o = reinterpret_cast<uint4*>(dedup_q + dst);
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        const ulonglong2* se = reinterpret_cast<const ulonglong2*>(sorted_enc + 8 * i);
        ulonglong2* de = reinterpret_cast<ulonglong2*>(dedup_enc + 8 * dst);
        de[0] = se[0]; de[1] = se[1]; de[2] = se[2]; de[3] = se[3];
    }
}
// last_fresh[i] = index of the latest fresh request at or before i (0 when there is none yet)
static __device__ __forceinline__ void k_decommit_last_fresh(const VB& vb, const u32* __restrict__ fresh_count, const u32* __restrict__ fresh_pos, size_t n, u32* __restrict__ last_fresh) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i < n) last_fresh[i] = fresh_count[i] ? fresh_pos[fresh_count[i] - 1] : 0u;
}

struct DecommitBlock {
    const zkw_decommit_query* sorted_q;
    const u64* unsorted_tails;
    const u64* sorted_tails;
    const u64* dedup_tails;   // [n_dedup][12]
    const u64* lhs_z;         // [2][n]
    const u64* rhs_z;
    const u32* fresh_count;   // inclusive
    const u32* last_fresh;
    zkw_decommit_sorter_instance* instances;
    zkw_queue_state12 dedup_in;  // state of the deduplicated queue before the call
    u64 n;
    u32 capacity;
};

__device__ __forceinline__ void qs12(zkw_queue_state12& s, const u64* head, const u64* tail, u32 len) {
    for (int k = 0; k < 12; k++) { s.head[k] = head ? head[k] : 0; s.tail[k] = tail ? tail[k] : 0; }
    s.length = len;
    s._pad = 0;
}

// deduplicated-queue state after `cnt` fresh requests have been pushed
__device__ __forceinline__ void dedup_state_at(const DecommitBlock& b, u32 cnt, zkw_queue_state12& s) {
    qs12(s, b.dedup_in.head, cnt ? b.dedup_tails + 12 * (size_t)(cnt - 1) : b.dedup_in.tail, b.dedup_in.length + cnt);
}

static __device__ __forceinline__ void k_decommit_instances(const VB& vb, const DecommitBlock* __restrict__ blk) {
    const DecommitBlock& b = *blk;
    const u64 n = b.n, n_inst = (n + b.capacity - 1) / b.capacity;
    const u64 idx = (u64)vb.x * blockDim.x + threadIdx.x;
    if (idx >= n_inst) return;
    zkw_decommit_sorter_instance& w = b.instances[idx];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    const u64 lo = idx * b.capacity, hi = lo + b.capacity < n ? lo + b.capacity : n;
    w.start_flag = idx == 0;
    w.completion_flag = idx == n_inst - 1;
    w.first_item = lo;
    w.num_items = hi - lo;
    const u64* u_final = b.unsorted_tails + 12 * (n - 1);
    const u64* s_final = b.sorted_tails + 12 * (n - 1);
    qs12(w.initial_queue_state, nullptr, u_final, (u32)n);
    qs12(w.sorted_queue_initial_state, nullptr, s_final, (u32)n);
    if (idx == n_inst - 1) dedup_state_at(b, b.fresh_count[n - 1], w.final_queue_state);
    auto fill = [&](zkw_decommit_sorter_fsm& f, u64 end /* items consumed, > 0 */) {
        const u64 l = end - 1;
        const bool full = end % b.capacity == 0;  // chunk boundary reached with counter == capacity
        const bool last = end == n;
        qs12(f.initial_queue_state, b.unsorted_tails + 12 * l, u_final, (u32)(n - end));
        qs12(f.sorted_queue_state, b.sorted_tails + 12 * l, s_final, (u32)(n - end));
        // :150-158: at a non-final boundary the snapshot EXCLUDES the most recent fresh push
        const u32 cnt = b.fresh_count[l];
        dedup_state_at(b, (full && !last && cnt > 0) ? cnt - 1 : cnt, f.final_queue_state);
        for (int r = 0; r < 2; r++) { f.lhs_accumulator[r] = b.lhs_z[r * n + l]; f.rhs_accumulator[r] = b.rhs_z[r * n + l]; }
        if (full) {
            const zkw_decommit_query* q = b.sorted_q + l;
            f.previous_packed_key[0] = q->timestamp;
            for (int k = 0; k < 8; k++) f.previous_packed_key[1 + k] = q->hash[k];
            f.previous_record = *q;
            f.previous_record.decommitted_length = 0;
            const u32 lf = b.last_fresh[l];
            f.first_encountered_timestamp = lf == 0xFFFFFFFFu ? 0 : b.sorted_q[lf].timestamp;
        }
    };
    if (idx > 0) fill(w.hidden_fsm_input, lo);
    fill(w.hidden_fsm_output, hi);
}

}  // namespace zkw
