// demux_kernels.cuh — LogDemuxer witness builder on gfx950.
// Reference: compute_logs_demux, src/witness/individual_circuits/log_demux.rs:20-388.
// The reference pushes each log into one of six queues while walking the input; here the route of an item
// is a local predicate, the six queues are stable compactions (prefix counts per route) and their hash
// chains run concurrently with the input queue's chain in one launch.
#pragma once
#include "events_kernels.cuh"
#include "scan_kernels.cuh"

namespace zkw {

__device__ __forceinline__ int demux_route(const zkw_log_query& q, const zkw_demux_params& p) {
    if (q.aux_byte == p.storage_aux_byte) return q.shard_id == 0 ? ZKW_DEMUX_STORAGE : -2;
    if (q.aux_byte == p.l1_message_aux_byte) return ZKW_DEMUX_L1_MESSAGES;
    if (q.aux_byte == p.event_aux_byte) return ZKW_DEMUX_EVENTS;
    if (q.aux_byte == p.precompile_aux_byte) {
        if (q.rollback) return -2;
        const bool high_zero = (q.address[1] | q.address[2] | q.address[3] | q.address[4]) == 0;
        if (high_zero && q.address[0] == p.keccak256_address) return ZKW_DEMUX_KECCAK256;
        if (high_zero && q.address[0] == p.sha256_address) return ZKW_DEMUX_SHA256;
        if (high_zero && q.address[0] == p.ecrecover_address) return ZKW_DEMUX_ECRECOVER;
        return -1;
    }
    return -2;
}

// the route of item i: the functor of route_prefix (scan_kernels.cuh); -2 (no circuit takes the log) counts as a violation below
struct DemuxRoute {
    const zkw_log_query* q;
    zkw_demux_params params;
    __device__ int operator()(size_t i) const {
        zkw_log_query m;
        m.aux_byte = q[i].aux_byte; m.shard_id = q[i].shard_id; m.rollback = q[i].rollback;
        for (int k = 0; k < 5; k++) m.address[k] = q[i].address[k];
        return demux_route(m, params);
    }
};

// queue offsets from the routes' totals (totals[7] = violations is zeroed by the caller and counted by k_demux_route)
static __device__ __forceinline__ void k_demux_offsets(const VB& vb, const u32* __restrict__ route_count /* [6][n] inclusive */, size_t n, u64* __restrict__ totals /* [8] */) {
    if (threadIdx.x || vb.x) return;
    u64 acc = 0;
    for (int c = 0; c < 6; c++) { totals[c] = acc; acc += n ? route_count[(size_t)c * n + n - 1] : 0; }
    totals[6] = acc;
}

// every item on its own, given the tiled inclusive counts per route: the scatter of the routed items and their encodings into
// the six queues (stable: position = offset of the queue + items of the same route before it)
static __device__ __forceinline__ void k_demux_route(const VB& vb, const zkw_log_query* __restrict__ q, const u64* __restrict__ in_enc, size_t n,
                                                     zkw_demux_params params, const u32* __restrict__ route_count /* [6][n] inclusive */,
                                                     zkw_log_query* __restrict__ out_q, u64* __restrict__ out_enc,
                                                     u64* __restrict__ totals /* [8]: offsets[7], violations */) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = DemuxRoute{q, params}(i);
    if (r == -2) atomicAdd(reinterpret_cast<unsigned long long*>(&totals[7]), 1ull);
    if (r < 0) return;
    const size_t dst = totals[r] + route_count[(size_t)r * n + i] - 1;
    zkw_log_query m;
    load_log(q + i, m);
    store_log(out_q + dst, m);
    const ulonglong2* se = reinterpret_cast<const ulonglong2*>(in_enc + 20 * i);
    ulonglong2* de = reinterpret_cast<ulonglong2*>(out_enc + 20 * dst);
#pragma unroll
    for (int k = 0; k < 10; k++) de[k] = se[k];
}

struct DemuxBlock {
    const u64* in_new_tails;   // [n][4]
    const u64* out_new_tails;  // [routed][4], queues back to back
    const u32* route_count;    // [6][n] inclusive
    zkw_log_demux_instance* instances;
    u64 offsets[7];
    u64 n;
    u32 capacity;
};

static __device__ __forceinline__ void k_demux_instances(const VB& vb, const DemuxBlock* __restrict__ blk) {
    const DemuxBlock b = *blk;
    const u64 n = b.n, n_inst = (n + b.capacity - 1) / b.capacity;
    const u64 idx = (u64)vb.x * blockDim.x + threadIdx.x;
    if (idx >= n_inst) return;
    zkw_log_demux_instance& w = b.instances[idx];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    const u64 lo = idx * b.capacity, hi = lo + b.capacity < n ? lo + b.capacity : n;
    w.start_flag = idx == 0;
    w.completion_flag = idx == n_inst - 1;
    w.first_item = lo;
    w.num_items = hi - lo;
    const u64* full_tail = b.in_new_tails + 4 * (n - 1);
    qs4(w.initial_log_queue_state, nullptr, full_tail, (u32)n);
    auto fill = [&](zkw_log_demux_fsm& f, u64 end /* > 0 */) {
        qs4(f.initial_log_queue_state, b.in_new_tails + 4 * (end - 1), full_tail, (u32)(n - end));
        for (int c = 0; c < 6; c++) {
            const u32 cnt = b.route_count[(size_t)c * n + end - 1];
            qs4(f.queue_state[c], nullptr, cnt ? b.out_new_tails + 4 * (b.offsets[c] + cnt - 1) : nullptr, cnt);
        }
    };
    if (idx > 0) fill(w.hidden_fsm_input, lo);
    fill(w.hidden_fsm_output, hi);
    if (idx == n_inst - 1)
        for (int c = 0; c < 6; c++) w.output_queue_state[c] = w.hidden_fsm_output.queue_state[c];
}

}  // namespace zkw
