// ram_circuit_kernels.cuh — RAMPermutation synthesis on gfx950: materialisation of the "zkw trace v1"
// (include/zkw_ram_circuit_spec.h) and its satisfiability check.
//
// Replaces, for the RAMPermutation instance type, ZkSyncBaseLayerCircuit::synthesis
// (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:286-323; wrapper
// base_layer/ram_permutation.rs:26-135) and the `check_if_satisfied` pass of the reference's tests
// (src/tests/mod.rs:130-259).
//
// Layout: column-major u64[RC_COLS][n_rows]; the trace is region-major (row of (region r, cycle i) =
// r*capacity + i), so a wave of 64 consecutive cycles writes 512 contiguous bytes of every column it
// touches: every store is a fully-used, coalesced line and every cell of the trace is written exactly
// once (no memset pass): algorithmic bytes = RC_COLS * n_rows * 8.
#pragma once
#include "../../include/zkw_ram_circuit_spec.h"
#include "../../include/zkw_decommit_sorter_circuit_spec.h"
#include "../../include/zkw_events_sorter_circuit_spec.h"
#include "ram_kernels.cuh"
#include "public_input_kernels.cuh"
#include "closed_form_kernels.cuh"

namespace zkw {

struct SynthJob {
    const zkw_ram_instance* inst;  // device
    const zkw_mem_query* sorted_q;  // block-wide arrays (device), indexed by item
    const zkw_mem_query* unsorted_q;  // the block's queue in its original order; encodings are computed on the fly
    const u64* unsorted_caps;  // [n_block][4] capacity words of the unsorted queue tail after every item
    const u64* sorted_caps;
    const u64* u_mark;         // [12] full unsorted / sorted queue tail after this instance's last item
    const u64* s_mark;
    const u64* challenges;  // [2][9]
    const u64* lhs_z;       // [2][n_block] grand-product chains of the instance's block
    const u64* rhs_z;
    u64 n_block;            // items in the block (stride between the two repetitions)
    u64* trace;             // [RC_COLS][n_rows]
    u32* hist;              // [256] lookup-value histogram of this trace (zeroed before the fills)
    u32* nd_tiles;          // [ceil(capacity/256)] exclusive prefix of nondeterministic writes per 256-cycle tile
    const u64* public_input;  // [4] commitment of the instance's closed-form input (k_ram_commitments): not written into the trace (the
                              // closed-form section derives the PI row), kept for callers that compare
    const zkw_ram_instance* first_inst;  // the block's first instance (its observable input is every instance's: postprocessing/mod.rs:358-364)
    u32 tail_clean;  // the slot already holds this layout (same circuit, capacity, rows): every cell that is zero in EVERY trace of the layout — the padding
                     // rows below the boundary rows, the gap rows of the regions, the unused columns of each row type, rows >= 256 of the multiplicity column —
                     // is still zero: the kernels skip those stores
};

constexpr int ROW_SLOTS[RC_NUM_ROW_TYPES] = RC_ROW_NUM_SLOTS_INIT;
constexpr int ROW_LOOKUPS[RC_NUM_ROW_TYPES] = RC_ROW_NUM_LOOKUPS_INIT;
static_assert(ROW_SLOTS[RC_ROW_PU] == 133 && ROW_SLOTS[RC_ROW_PS] == 130, "Poseidon2 rows: 130 gate variables (+3 spare in PU)");
static_assert(ROW_LOOKUPS[RC_ROW_PU] == 12 && ROW_LOOKUPS[RC_ROW_PS] == 12 && ROW_LOOKUPS[RC_ROW_A] == 12 &&
              ROW_LOOKUPS[RC_ROW_B] == 12 && ROW_LOOKUPS[RC_ROW_C] == 8 && ROW_LOOKUPS[RC_ROW_D] == 0,
              "lookup cells per cycle (56) are hard-wired in the fill kernels");
constexpr int LOOKUPS_PER_CYCLE = 56;

__device__ __forceinline__ u64 inv_or_zero(u64 x) { return gl::canon(x) ? gl::inv(x) : 0; }

// effective registers at "cycle -1"
struct RegsIn {
    const u64* uh;  // [12]
    const u64* sh;
    u32 len;
};
__device__ __forceinline__ RegsIn regs_in(const zkw_ram_instance* in) {
    RegsIn r;
    const bool start = in->start_flag != 0;
    r.uh = start ? in->unsorted_queue_initial_state.head : in->hidden_fsm_input.current_unsorted_queue_state.head;
    r.sh = start ? in->sorted_queue_initial_state.head : in->hidden_fsm_input.current_sorted_queue_state.head;
    r.len = start ? in->unsorted_queue_initial_state.length : in->hidden_fsm_input.current_unsorted_queue_state.length;
    return r;
}

#define TR(col, row) trace[(size_t)(col) * n_rows + (row)]

// rows [capacity, RC_REGION_STRIDE(capacity)) of a region: the alignment gap, all general + lookup columns zero
__device__ __forceinline__ void zero_gap_row(u64* trace, size_t n_rows, size_t row) {
    for (int col = 0; col < RC_G + RC_L; col++) TR(col, row) = 0;
}
__device__ __forceinline__ void zero_gap_row_n(u64* trace, size_t n_rows, size_t row, int n_cols) {
    for (int col = 0; col < n_cols; col++) TR(col, row) = 0;
}

__device__ __forceinline__ void hist_bytes(u32* sh_hist, u32 x) {
    atomicAdd(&sh_hist[x & 0xFF], 1u);
    atomicAdd(&sh_hist[(x >> 8) & 0xFF], 1u);
    atomicAdd(&sh_hist[(x >> 16) & 0xFF], 1u);
    atomicAdd(&sh_hist[x >> 24], 1u);
}
__device__ __forceinline__ void put_bytes(u64* trace, size_t n_rows, size_t row, int col0, u32 x) {
    TR(col0, row) = x & 0xFF;
    TR(col0 + 1, row) = (x >> 8) & 0xFF;
    TR(col0 + 2, row) = (x >> 16) & 0xFF;
    TR(col0 + 3, row) = x >> 24;
}
__device__ __forceinline__ void hist_flush(u32* sh_hist, u32* g_hist) {
    __syncthreads();
    for (int t = threadIdx.x; t < 256; t += blockDim.x)
        if (sh_hist[t]) atomicAdd(&g_hist[t], sh_hist[t]);
}

// One flattened Poseidon2 gate: columns 0..129 of `row` = the 12 inputs, the state after each of the first 4 full
// rounds, the S-box output of element 0 in each of the 22 partial rounds, the state after each of the last 4 full
// rounds. s holds the (weak) output state on return.
__device__ __forceinline__ void fill_flattened_poseidon(u64* trace, size_t n_rows, size_t row, u64 s[12]) {
    int pos = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) TR(pos++, row) = gl::canon(s[k]);
    p2::external(s);
    int r = 0;
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
        p2::full_round(s, r);
#pragma unroll
        for (int j = 0; j < 12; j++) TR(pos + j, row) = gl::canon(s[j]);
        pos += 12;
    }
    for (int k = 0; k < P2_PARTIAL_ROUNDS; k++, r++) {
        s[0] = gl::pow7(gl::add_canon(s[0], p2::rc_at(12 * r)));
        TR(pos++, row) = gl::canon(s[0]);
        p2::internal(s);
    }
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
        p2::full_round(s, r);
#pragma unroll
        for (int j = 0; j < 12; j++) TR(pos + j, row) = gl::canon(s[j]);
        pos += 12;
    }
}

// ------------------------------------------------------------------------------------------------
// Poseidon2 rows (regions PU and PS): one lane per cycle runs the permutation and stores all 130
// flattened-gate variables as it goes. SIDE 0 = unsorted queue, 1 = sorted queue.
template <int SIDE>
static __device__ __forceinline__ void k_ram_fill_poseidon(const VB& vb, const SynthJob* __restrict__ jobs, u32 capacity,
                                                          size_t n_rows) {
    __shared__ u32 sh_hist[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) sh_hist[t] = 0;
    __syncthreads();
    const SynthJob job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    if (i < capacity) {
        u64* trace = job.trace;
        const zkw_ram_instance* in = job.inst;
        const size_t first = in->first_item, m = in->num_items;
        const bool can_pop = i < m;
        const size_t row = (size_t)(SIDE == 0 ? RC_ROW_PU : RC_ROW_PS) * RC_REGION_STRIDE(capacity) + i;
        const zkw_mem_query* qs = SIDE == 0 ? job.unsorted_q : job.sorted_q;
        const u64* caps = SIDE == 0 ? job.unsorted_caps : job.sorted_caps;
        u64 s[12];
        if (can_pop) {
            encode_raw_query(load_raw_query(qs + first + i), s);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = 0;
        }
        const RegsIn ri = regs_in(in);
        // capacity part of the queue head entering this cycle
        const u64* prev_cap = i == 0 ? (SIDE == 0 ? ri.uh : ri.sh) + 8 : caps + 4 * (first + (i - 1 < m ? i - 1 : m - 1));
#pragma unroll
        for (int k = 0; k < 4; k++) s[8 + k] = prev_cap[k];
        fill_flattened_poseidon(trace, n_rows, row, s);
        // spare general slots + lookup cells of this row type
        zkw_mem_query q;
        memset(&q, 0, sizeof q);
        if (can_pop) q = job.sorted_q[first + i];
        if (SIDE == 0) {
            TR(RC_PU_idx, row) = q.index; TR(RC_PU_v0, row) = q.value[0]; TR(RC_PU_v1, row) = q.value[1];
            put_bytes(trace, n_rows, row, RC_PU_idx_b0, q.index);
            put_bytes(trace, n_rows, row, RC_PU_v0_b0, q.value[0]);
            put_bytes(trace, n_rows, row, RC_PU_v1_b0, q.value[1]);
            hist_bytes(sh_hist, q.index); hist_bytes(sh_hist, q.value[0]); hist_bytes(sh_hist, q.value[1]);
        } else {
            if (!job.tail_clean) for (int c = 130; c < RC_G; c++) TR(c, row) = 0;
            put_bytes(trace, n_rows, row, RC_PS_ts_b0, q.timestamp);
            put_bytes(trace, n_rows, row, RC_PS_page_b0, q.page);
            put_bytes(trace, n_rows, row, RC_PS_v4_b0, q.value[4]);
            hist_bytes(sh_hist, q.timestamp); hist_bytes(sh_hist, q.page); hist_bytes(sh_hist, q.value[4]);
        }
        if (!job.tail_clean) for (int c = RC_G + 12; c < RC_G + RC_L; c++) TR(c, row) = 0;
    } else if (i < RC_REGION_STRIDE(capacity)) {
        if (!job.tail_clean) zero_gap_row(job.trace, n_rows, (size_t)(SIDE == 0 ? RC_ROW_PU : RC_ROW_PS) * RC_REGION_STRIDE(capacity) + i);
    }
    hist_flush(sh_hist, job.hist);
}

// ------------------------------------------------------------------------------------------------
// General rows A..D: one lane per cycle; every value is a function of item i, item i-1 and the
// instance record (no carried state), so all cycles are independent.
struct CycleCtx {
    bool can_pop;
    zkw_mem_query q, pq;   // this item, previous item (zeros when padding / FSM-in for cycle 0)
    u64 eu[8], es[8];
    u64 p_val[5];          // es3..es6, v4 of the previous item
    u32 p_ptr;
};

__device__ __forceinline__ void load_query(const zkw_mem_query* src, zkw_mem_query& dst) {
    const uint4* s = reinterpret_cast<const uint4*>(src);
    uint4* d = reinterpret_cast<uint4*>(&dst);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

__device__ __forceinline__ void cycle_ctx(const SynthJob& job, u32 i, CycleCtx& c, bool want_eu) {
    const zkw_ram_instance* in = job.inst;
    const size_t first = in->first_item, m = in->num_items;
    c.can_pop = i < m;
    memset(&c.q, 0, sizeof c.q);
    memset(&c.pq, 0, sizeof c.pq);
#pragma unroll
    for (int k = 0; k < 8; k++) { c.eu[k] = 0; c.es[k] = 0; }
    if (c.can_pop) {
        load_query(job.sorted_q + first + i, c.q);
        encode_mem_query(c.q, c.es);
        if (want_eu) encode_raw_query(load_raw_query(job.unsorted_q + first + i), c.eu);
    }
    if (i == 0) {
        const zkw_ram_fsm& f = in->hidden_fsm_input;
        c.pq.timestamp = f.previous_sorting_key[0]; c.pq.index = f.previous_sorting_key[1]; c.pq.page = f.previous_sorting_key[2];
        for (int k = 0; k < 8; k++) c.pq.value[k] = f.previous_value[k];
        c.pq.value_is_pointer = f.previous_is_ptr ? 1 : 0;
    } else if (i - 1 < m) {
        load_query(job.sorted_q + first + i - 1, c.pq);
    }
    u64 pe[8];
    encode_mem_query(c.pq, pe);
#pragma unroll
    for (int k = 0; k < 5; k++) c.p_val[k] = pe[3 + k];
    c.p_ptr = c.pq.value_is_pointer ? 1 : 0;
}

// accumulator entering cycle i: FSM input for i = 0, else the chain value at the last popped item
__device__ __forceinline__ u64 acc_before(const u64* z, size_t first, size_t m, u32 i, u64 fsm_in) {
    return i == 0 ? fsm_in : z[first + (i - 1 < m ? i - 1 : m - 1)];
}

static __device__ __forceinline__ void k_ram_fill_A(const VB& vb, const SynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const SynthJob job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const u64* lhs_z_all = job.lhs_z;
    const u64* rhs_z_all = job.rhs_z;
    const size_t n_total = job.n_block;
    if (i < capacity) {
        u64* trace = job.trace;
        const zkw_ram_instance* in = job.inst;
        const size_t first = in->first_item, m = in->num_items, row = (size_t)RC_ROW_A * RC_REGION_STRIDE(capacity) + i;
        CycleCtx c;
        cycle_ctx(job, i, c, true);
        const u64 rw = c.q.rw_flag ? 1 : 0, ptr = c.q.value_is_pointer ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { TR(RC_A_eu0 + k, row) = c.eu[k]; TR(RC_A_ts + k, row) = c.es[k]; }
        for (int r = 0; r < 2; r++) {
            const u64* ch = job.challenges + 9 * r;
            u64 lc = ch[8], rc = ch[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                lc = gl::add(lc, gl::mul(c.eu[k], ch[k]));
                rc = gl::add(rc, gl::mul(c.es[k], ch[k]));
            }
            const u64 pl = acc_before(lhs_z_all + (size_t)r * n_total, first, m, i, in->hidden_fsm_input.lhs_accumulator[r]);
            const u64 pr = acc_before(rhs_z_all + (size_t)r * n_total, first, m, i, in->hidden_fsm_input.rhs_accumulator[r]);
            const u64 nl = gl::canon(gl::mul(pl, lc)), nr = gl::canon(gl::mul(pr, rc));
            const int o = r * (RC_A_lc1 - RC_A_lc0);
#pragma unroll
            for (int k = 1; k < 9; k++) TR(RC_A_G_c0_1 + (k - 1) + r * (RC_A_G_c1_1 - RC_A_G_c0_1), row) = ch[k];
            TR(RC_A_lc0 + o, row) = gl::canon(lc); TR(RC_A_P_lhs0 + o, row) = pl; TR(RC_A_nl0 + o, row) = nl;
            TR(RC_A_lhs0 + o, row) = c.can_pop ? nl : pl;
            TR(RC_A_rc0 + o, row) = gl::canon(rc); TR(RC_A_P_rhs0 + o, row) = pr; TR(RC_A_nr0 + o, row) = nr;
            TR(RC_A_rhs0 + o, row) = c.can_pop ? nr : pr;
        }
        TR(RC_A_rw, row) = rw; TR(RC_A_ptr, row) = ptr; TR(RC_A_idx, row) = c.q.index;
        TR(RC_A_v2, row) = c.q.value[2]; TR(RC_A_v3, row) = c.q.value[3]; TR(RC_A_v0, row) = c.q.value[0];
        TR(RC_A_v5_b3c, row) = c.q.value[5] >> 24; TR(RC_A_can_pop, row) = c.can_pop ? 1 : 0;
        put_bytes(trace, n_rows, row, RC_A_v2_b0, c.q.value[2]);
        put_bytes(trace, n_rows, row, RC_A_v3_b0, c.q.value[3]);
        put_bytes(trace, n_rows, row, RC_A_v5_b0, c.q.value[5]);
        hist_bytes(sh_hist, c.q.value[2]); hist_bytes(sh_hist, c.q.value[3]); hist_bytes(sh_hist, c.q.value[5]);
        constexpr int NA = ROW_SLOTS[RC_ROW_A];  // general slots used by row type A
        if (!job.tail_clean) {
            for (int col = NA; col < RC_G; col++) TR(col, row) = 0;
            for (int col = RC_G + 12; col < RC_G + RC_L; col++) TR(col, row) = 0;
        }
    } else if (i < RC_REGION_STRIDE(capacity)) {
        if (!job.tail_clean) zero_gap_row(job.trace, n_rows, (size_t)RC_ROW_A * RC_REGION_STRIDE(capacity) + i);
    }
    hist_flush(sh_hist, job.hist);
}

static __device__ __forceinline__ void k_ram_fill_B(const VB& vb, const SynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const SynthJob job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    if (i < capacity) {
        u64* trace = job.trace;
        const size_t row = (size_t)RC_ROW_B * RC_REGION_STRIDE(capacity) + i;
        CycleCtx c;
        cycle_ctx(job, i, c, false);
        put_bytes(trace, n_rows, row, RC_B_v6_b0, c.q.value[6]);
        put_bytes(trace, n_rows, row, RC_B_v7_b0, c.q.value[7]);
        TR(RC_B_es4, row) = c.es[4]; TR(RC_B_v1, row) = c.q.value[1]; TR(RC_B_v5_b3c, row) = c.q.value[5] >> 24;
        TR(RC_B_es5, row) = c.es[5]; TR(RC_B_v2, row) = c.q.value[2];
        TR(RC_B_es6, row) = c.es[6]; TR(RC_B_v3, row) = c.q.value[3];
        const u32 d0 = c.q.timestamp - c.pq.timestamp;
        TR(RC_B_d0, row) = d0; put_bytes(trace, n_rows, row, RC_B_d0_b0, d0);
        TR(RC_B_bw0, row) = c.q.timestamp < c.pq.timestamp ? 1 : 0;
        TR(RC_B_ts, row) = c.q.timestamp; TR(RC_B_P_ts, row) = c.pq.timestamp;
        hist_bytes(sh_hist, c.q.value[6]); hist_bytes(sh_hist, c.q.value[7]); hist_bytes(sh_hist, d0);
        constexpr int NB = ROW_SLOTS[RC_ROW_B];
        if (!job.tail_clean) {
            for (int col = NB; col < RC_G; col++) TR(col, row) = 0;
            for (int col = RC_G + 12; col < RC_G + RC_L; col++) TR(col, row) = 0;
        }
    } else if (i < RC_REGION_STRIDE(capacity)) {
        if (!job.tail_clean) zero_gap_row(job.trace, n_rows, (size_t)RC_ROW_B * RC_REGION_STRIDE(capacity) + i);
    }
    hist_flush(sh_hist, job.hist);
}

// nondeterministic-write flag of a cycle (circuit definition: popped & ts == 0 & heap page & write & !ptr)
__device__ __forceinline__ bool nd_flag(bool can_pop, const zkw_mem_query& q) {
    return can_pop && q.timestamp == 0 && q.page == RC_HEAP_PAGE && q.rw_flag && !q.value_is_pointer;
}

// per 256-cycle tile: number of nondeterministic writes; then an exclusive scan per instance
static __device__ __forceinline__ void k_ram_nd_tiles(const VB& vb, const SynthJob* __restrict__ jobs, u32 capacity) {
    __shared__ u32 sh[4];
    const SynthJob job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const zkw_ram_instance* in = job.inst;
    bool f = false;
    if (i < capacity && i < in->num_items) {
        zkw_mem_query q;
        load_query(job.sorted_q + in->first_item + i, q);
        f = nd_flag(true, q);
    }
    u32 cnt = __popcll(__ballot(f));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) job.nd_tiles[vb.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
static __device__ __forceinline__ void k_ram_nd_scan(const VB& vb, const SynthJob* __restrict__ jobs, int n_jobs, u32 n_tiles) {
    // one wave per job: exclusive prefix over the tile counts, 64 tiles at a time
    u32* t = jobs[vb.x].nd_tiles;
    const int lane = threadIdx.x;
    u32 acc = 0;
    for (u32 base = 0; base < n_tiles; base += 64) {
        const u32 k = base + lane;
        const u32 v = k < n_tiles ? t[k] : 0;
        u32 incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (k < n_tiles) t[k] = acc + incl - v;
        acc += __shfl(incl, 63);
    }
}

static __device__ __forceinline__ void k_ram_fill_C(const VB& vb, const SynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    __shared__ u32 sh_wave[4];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const SynthJob job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const bool live = i < capacity;
    CycleCtx c;
    c.can_pop = false;
    bool nd = false;
    if (live) {
        cycle_ctx(job, i, c, false);
        nd = nd_flag(c.can_pop, c.q);
    }
    // exclusive count of nd flags before this lane inside the block (tile)
    const unsigned long long bal = __ballot(nd);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh_wave[wave] = __popcll(bal);
    __syncthreads();
    u32 before = __popcll(bal & ((1ull << lane) - 1));
    for (int w = 0; w < wave; w++) before += sh_wave[w];
    if (live) {
        u64* trace = job.trace;
        const zkw_ram_instance* in = job.inst;
        const size_t row = (size_t)RC_ROW_C * RC_REGION_STRIDE(capacity) + i;
        const u64 rw = c.q.rw_flag ? 1 : 0, ptr = c.q.value_is_pointer ? 1 : 0;
        const u64 bw0 = c.q.timestamp < c.pq.timestamp ? 1 : 0;
        const u64 t1 = (u64)c.pq.index + bw0, bw1 = (u64)c.q.index < t1 ? 1 : 0;
        const u32 d1 = (u32)((u64)c.q.index - t1);
        const u64 t2 = (u64)c.pq.page + bw1, bw2 = (u64)c.q.page < t2 ? 1 : 0;
        const u32 d2 = (u32)((u64)c.q.page - t2);
        TR(RC_C_d1, row) = d1; put_bytes(trace, n_rows, row, RC_C_d1_b0, d1);
        TR(RC_C_d2, row) = d2; put_bytes(trace, n_rows, row, RC_C_d2_b0, d2);
        hist_bytes(sh_hist, d1); hist_bytes(sh_hist, d2);
        TR(RC_C_bw1, row) = bw1; TR(RC_C_bw2, row) = bw2; TR(RC_C_bw0, row) = bw0;
        TR(RC_C_idx, row) = c.q.index; TR(RC_C_P_idx, row) = c.pq.index;
        TR(RC_C_page, row) = c.q.page; TR(RC_C_P_page, row) = c.pq.page;
        TR(RC_C_can_pop, row) = c.can_pop ? 1 : 0;
        // 14 "inverse or zero" witnesses with one field inversion (Montgomery batch trick)
        u64 x[14];
        x[0] = gl::canon(gl::sub(c.q.index, c.pq.index));
        x[1] = gl::canon(gl::sub(c.q.page, c.pq.page));
        const u64 val[5] = {c.es[3], c.es[4], c.es[5], c.es[6], c.es[7]};
#pragma unroll
        for (int k = 0; k < 5; k++) { x[2 + k] = gl::canon(gl::sub(val[k], c.p_val[k])); x[7 + k] = val[k]; }
        x[12] = c.q.timestamp;
        x[13] = gl::canon(gl::sub(c.q.page, RC_HEAP_PAGE));
        u64 pre[14], acc = 1;
#pragma unroll
        for (int k = 0; k < 14; k++) { pre[k] = acc; acc = gl::mul(acc, x[k] ? x[k] : 1); }
        u64 inv = gl::inv(acc), w[14];
#pragma unroll
        for (int k = 13; k >= 0; k--) {
            w[k] = x[k] ? gl::canon(gl::mul(inv, pre[k])) : 0;
            inv = gl::mul(inv, x[k] ? x[k] : 1);
        }
        const u64 z_idx = x[0] == 0, z_page = x[1] == 0, same = z_idx & z_page;
        TR(RC_C_w_idx, row) = w[0]; TR(RC_C_z_idx, row) = z_idx;
        TR(RC_C_w_page, row) = w[1]; TR(RC_C_z_page, row) = z_page; TR(RC_C_same, row) = same;
        const int C_VAL[5] = {RC_C_es3, RC_C_es4, RC_C_es5, RC_C_es6, RC_C_v4};
        const int C_PVAL[5] = {RC_C_P_es3, RC_C_P_es4, RC_C_P_es5, RC_C_P_es6, RC_C_P_v4};
        const int C_WEQ[5] = {RC_C_w_eq0, RC_C_w_eq1, RC_C_w_eq2, RC_C_w_eq3, RC_C_w_eq4};
        const int C_ZEQ[5] = {RC_C_z_eq0, RC_C_z_eq1, RC_C_z_eq2, RC_C_z_eq3, RC_C_z_eq4};
        const int C_WZ[5] = {RC_C_w_z0, RC_C_w_z1, RC_C_w_z2, RC_C_w_z3, RC_C_w_z4};
        const int C_ZZ[5] = {RC_C_z_z0, RC_C_z_z1, RC_C_z_z2, RC_C_z_z3, RC_C_z_z4};
        u64 zeq[5], zz[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            zeq[k] = x[2 + k] == 0; zz[k] = x[7 + k] == 0;
            TR(C_VAL[k], row) = val[k]; TR(C_PVAL[k], row) = c.p_val[k];
            TR(C_WEQ[k], row) = w[2 + k]; TR(C_ZEQ[k], row) = zeq[k];
            TR(C_WZ[k], row) = w[7 + k]; TR(C_ZZ[k], row) = zz[k];
        }
        const u64 peq = ptr == c.p_ptr, eq_a = zeq[0] & zeq[1] & zeq[2], value_equal = eq_a & zeq[3] & zeq[4] & peq;
        const u64 zz_a = zz[0] & zz[1] & zz[2], all_zero = zz_a & zz[3] & zz[4] & (1 - ptr);
        TR(RC_C_ptr, row) = ptr; TR(RC_C_P_ptr, row) = c.p_ptr; TR(RC_C_peq, row) = peq;
        TR(RC_C_eq_a, row) = eq_a; TR(RC_C_value_equal, row) = value_equal;
        TR(RC_C_zz_a, row) = zz_a; TR(RC_C_all_zero, row) = all_zero; TR(RC_C_rw, row) = rw;
        TR(RC_C_ts, row) = c.q.timestamp; TR(RC_C_w_ts, row) = w[12]; TR(RC_C_z_ts, row) = x[12] == 0;
        TR(RC_C_w_heap, row) = w[13]; TR(RC_C_z_heap, row) = x[13] == 0; TR(RC_C_nd, row) = nd ? 1 : 0;
        const u64 p_cnt = (u64)in->hidden_fsm_input.num_nondeterministic_writes + job.nd_tiles[vb.x] + before;
        TR(RC_C_P_cnt, row) = p_cnt; TR(RC_C_cnt, row) = p_cnt + (nd ? 1 : 0);
        constexpr int NC = ROW_SLOTS[RC_ROW_C];
        if (!job.tail_clean) {
            for (int col = NC; col < RC_G; col++) TR(col, row) = 0;
            for (int col = RC_G + 8; col < RC_G + RC_L; col++) TR(col, row) = 0;
        }
    } else if (i < RC_REGION_STRIDE(capacity)) {
        if (!job.tail_clean) zero_gap_row(job.trace, n_rows, (size_t)RC_ROW_C * RC_REGION_STRIDE(capacity) + i);
    }
    if (vb.x == 0 && threadIdx.x == 0)  // the closed-form section's lookup cells (VIN / VOUT: bytes of limbs 5..7 of the previous value)
        for (int l = 5; l < 8; l++) { hist_bytes(sh_hist, job.inst->hidden_fsm_input.previous_value[l]); hist_bytes(sh_hist, job.inst->hidden_fsm_output.previous_value[l]); }
    hist_flush(sh_hist, job.hist);
}

// 1-D grid: the first n_jobs blocks fill the boundary rows of one trace each — the register rows and the closed-form section, a chain of a
// dozen dependent permutations (ram_boundary_block below) — dispatched first and at raised priority so that the row-D blocks' work hides
// it (it reads the last cycle's row C, an earlier kernel; nothing of row D); then n_tiles blocks of 256 cycles per trace.
__device__ __forceinline__ void ram_boundary_block(const SynthJob& job, u32 capacity, size_t n_rows);
constexpr int RC_D_TILES = 4;  // tiles of 256 cycles per row-D block
static __device__ __forceinline__ void k_ram_fill_D(const VB& vb, const SynthJob* __restrict__ jobs, u32 n_jobs, u32 n_tiles, u32 capacity, size_t n_rows) {
    if (vb.x < n_jobs) {
        __builtin_amdgcn_s_setprio(3);
        ram_boundary_block(jobs[vb.x], capacity, n_rows);
        return;
    }
    // a row-D block covers RC_D_TILES consecutive tiles of 256 cycles, a lane one cycle of each: the row's one expensive witness is the inverse of
    // the queue length, and the lane's RC_D_TILES inverses come from ONE field inversion (Montgomery's trick: 3 (K - 1) multiplications more)
    const SynthJob& job = jobs[(vb.x - n_jobs) / n_tiles];
    const u32 i0 = ((vb.x - n_jobs) % n_tiles) * (RC_D_TILES * 256) + threadIdx.x;
    const size_t rs = RC_REGION_STRIDE(capacity);
    u64* trace = job.trace;
    const zkw_ram_instance* in = job.inst;
    const size_t m = in->num_items;
    const RegsIn ri = regs_in(in);
    u64 len[RC_D_TILES], pre[RC_D_TILES], w[RC_D_TILES];
    u64 acc = 1;
#pragma unroll
    for (int k = 0; k < RC_D_TILES; k++) {
        const u32 i = i0 + 256 * k;
        len[k] = (u64)ri.len - (i < m ? i : m);
        pre[k] = acc;                                  // product of the lengths before this one (zero lengths count as 1)
        acc = gl::mul(acc, len[k] ? len[k] : 1);
    }
    acc = gl::inv(acc);
#pragma unroll
    for (int k = RC_D_TILES - 1; k >= 0; k--) {
        w[k] = len[k] ? gl::canon(gl::mul(acc, pre[k])) : 0;
        acc = gl::mul(acc, len[k] ? len[k] : 1);
    }
#pragma unroll
    for (int k = 0; k < RC_D_TILES; k++) {
        const u32 i = i0 + 256 * k;
        if (i >= capacity) {
            if (i < rs && !job.tail_clean) zero_gap_row(job.trace, n_rows, (size_t)RC_ROW_D * rs + i);
            continue;
        }
        const size_t row = (size_t)RC_ROW_D * rs + i;
        const size_t rPU = (size_t)RC_ROW_PU * rs + i, rPS = (size_t)RC_ROW_PS * rs + i;
        const bool can_pop = i < m;
        const u64 p_len = len[k];
        const u64 w_len = w[k], z_len = p_len == 0;
        TR(RC_D_P_len_u, row) = p_len; TR(RC_D_w_lu, row) = w_len; TR(RC_D_z_lu, row) = z_len;
        TR(RC_D_P_len_s, row) = p_len; TR(RC_D_w_ls, row) = w_len; TR(RC_D_z_ls, row) = z_len;
        TR(RC_D_can_pop, row) = can_pop ? 1 : 0;
        TR(RC_D_len_u, row) = p_len - (can_pop ? 1 : 0); TR(RC_D_len_s, row) = p_len - (can_pop ? 1 : 0);
        // The Poseidon2 rows (written earlier on this stream by k_ram_fill_poseidon) hold the queue tails: the output of
        // a popped cycle IS the tail after that item. The head entering cycle i is the output of the last popped cycle
        // before it (the FSM input for i = 0).
        const size_t p = i - 1 < m ? i - 1 : m - 1, rPUp = (size_t)RC_ROW_PU * rs + p, rPSp = (size_t)RC_ROW_PS * rs + p;
#pragma unroll
        for (int e = 0; e < 12; e++) {
            const u64 uo = TR(RC_PU_uo0 + e, rPU);
            const u64 so = TR(RC_PS_so0 + e, rPS);
            const u64 a = i == 0 ? ri.uh[e] : TR(RC_PU_uo0 + e, rPUp), b = i == 0 ? ri.sh[e] : TR(RC_PS_so0 + e, rPSp);
            const int o = 3 * e;
            TR(RC_D_uo0 + o, row) = uo; TR(RC_D_P_uh0 + o, row) = a; TR(RC_D_uh0 + o, row) = can_pop ? uo : a;
            TR(RC_D_so0 + o, row) = so; TR(RC_D_P_sh0 + o, row) = b; TR(RC_D_sh0 + o, row) = can_pop ? so : b;
        }
        constexpr int ND = ROW_SLOTS[RC_ROW_D];
        if (!job.tail_clean) for (int col = ND; col < RC_G + RC_L; col++) TR(col, row) = 0;
    }
}

// the zero padding from the first boundary row down (all general + lookup columns) and the multiplicity column.
// blockIdx.x = (column, chunk): every block streams one contiguous chunk of one column with 16-byte stores, so
// at any time the blocks in flight write ~all columns at once (a sweep of all blocks over one column at a time
// runs at 3.9 TB/s, this at 5.6, tools/ubench_fill.hip). The last TAIL_CHUNKS blocks of a job do the multiplicity
// column. grid.y = job.
constexpr int TAIL_CHUNKS = 8;
constexpr int RC_BOUNDARY_ROWS = RC_NUM_ROW_TYPES - RC_ROWS_PER_CYCLE;  // register rows, PI, the closed-form section
constexpr int RC_CF_LOOKUP_CELLS = 24;                                   // VIN / VOUT byte cells (counted in job.hist by k_ram_fill_C)
static_assert(RC_BOUNDARY_ROWS % 2 == 0, "the tail's 16-byte stores start below the boundary rows");
static __device__ __forceinline__ void k_ram_fill_tail(const VB& vb, const SynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    // 1-D grid: (RC_G + RC_L + 1) * TAIL_CHUNKS blocks per trace (the boundary rows above the zero padding are k_ram_fill_D's first blocks)
    constexpr u32 PER_JOB = (RC_G + RC_L + 1) * TAIL_CHUNKS;
    const u32 bid = vb.x % PER_JOB;
    const SynthJob& job = jobs[vb.x / PER_JOB];
    u64* trace = job.trace;
    const int col = bid / TAIL_CHUNKS, ch = bid % TAIL_CHUNKS;
    if (col < RC_G + RC_L) {
        if (job.tail_clean) return;
        const size_t bnd = (size_t)RC_BOUNDARY_ROW(capacity) + RC_BOUNDARY_ROWS;  // a multiple of 2 rows: 16-byte aligned
        const size_t n_pairs = (n_rows - bnd) / 2;                               // n_rows is even (power of two)
        const size_t per = (n_pairs + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
        ulonglong2* c2 = reinterpret_cast<ulonglong2*>(trace + (size_t)col * n_rows + bnd);
        const ulonglong2 z = make_ulonglong2(0, 0);
        for (size_t k = lo + threadIdx.x; k < hi; k += 256) c2[k] = z;
        return;
    }
    // multiplicities: histogram of the used lookup cells + every unused lookup cell counts as value 0
    u64* m = trace + (size_t)RC_MULT_COL * n_rows;
    const size_t per = (n_rows + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_rows ? lo + per : n_rows;
    for (size_t r = lo + threadIdx.x; r < (job.tail_clean && hi > 256 ? (lo < 256 ? 256 : lo) : hi); r += 256) {  // (a clean slot: rows >= 256 of the column are still zero)
        u64 v = 0;
        if (r < 256) {
            v = job.hist[r];
            if (r == 0) v += (u64)RC_L * n_rows - (u64)LOOKUPS_PER_CYCLE * capacity - RC_CF_LOOKUP_CELLS;
        }
        m[r] = v;
    }
}

static __constant__ rc_link c_links[RC_NUM_LINKS] = RC_LINKS_INIT;
static __constant__ uint8_t c_is_poseidon[RC_NUM_ROW_TYPES] = RC_ROW_IS_POSEIDON_INIT;
ZKW_CF_TABLES(RC, rc)

// the register rows BND_IN / BND_OUT (one lane)
__device__ __forceinline__ void ram_fill_register_rows(const SynthJob& job, u32 capacity, size_t n_rows) {
    const u64* lhs_z_all = job.lhs_z;
    const u64* rhs_z_all = job.rhs_z;
    const size_t n_total = job.n_block;
    u64* trace = job.trace;
    const zkw_ram_instance* in = job.inst;
    const zkw_ram_fsm& fi = in->hidden_fsm_input;
    const size_t first = in->first_item, m = in->num_items;
    const size_t bin = (size_t)RC_BOUNDARY_ROW(capacity) + RC_ROWOFF_BND_IN, bout = bin - RC_ROWOFF_BND_IN + RC_ROWOFF_BND_OUT;
    const RegsIn ri = regs_in(in);
    for (int k = 0; k < 12; k++) { TR(RC_BND_IN_uh0 + k, bin) = ri.uh[k]; TR(RC_BND_IN_sh0 + k, bin) = ri.sh[k]; }
    TR(RC_BND_IN_len_u, bin) = ri.len; TR(RC_BND_IN_len_s, bin) = ri.len;
    for (int r = 0; r < 2; r++) { TR(RC_BND_IN_lhs0 + r, bin) = fi.lhs_accumulator[r]; TR(RC_BND_IN_rhs0 + r, bin) = fi.rhs_accumulator[r]; }
    zkw_mem_query pq;
    memset(&pq, 0, sizeof pq);
    for (int k = 0; k < 8; k++) pq.value[k] = fi.previous_value[k];
    u64 pe[8];
    encode_mem_query(pq, pe);
    TR(RC_BND_IN_ts, bin) = fi.previous_sorting_key[0]; TR(RC_BND_IN_idx, bin) = fi.previous_sorting_key[1];
    TR(RC_BND_IN_page, bin) = fi.previous_sorting_key[2];
    TR(RC_BND_IN_es3, bin) = pe[3]; TR(RC_BND_IN_es4, bin) = pe[4]; TR(RC_BND_IN_es5, bin) = pe[5];
    TR(RC_BND_IN_es6, bin) = pe[6]; TR(RC_BND_IN_v4, bin) = pe[7];
    TR(RC_BND_IN_ptr, bin) = fi.previous_is_ptr ? 1 : 0; TR(RC_BND_IN_cnt, bin) = fi.num_nondeterministic_writes;
    for (int r = 0; r < 2; r++)
        for (int k = 1; k < 9; k++) TR(RC_BND_IN_G_c0_1 + 8 * r + (k - 1), bin) = job.challenges[9 * r + k];
    // BND_OUT = registers after the last cycle
    const size_t last = first + m - 1;
    const bool padded = m < capacity;
    const u64 len_out = (u64)ri.len - m;
    for (int k = 0; k < 12; k++) {
        TR(RC_BND_OUT_uh0 + k, bout) = job.u_mark[k];
        TR(RC_BND_OUT_sh0 + k, bout) = job.s_mark[k];
        TR(RC_BND_OUT_tail_u0 + k, bout) = in->unsorted_queue_initial_state.tail[k];
        TR(RC_BND_OUT_tail_s0 + k, bout) = in->sorted_queue_initial_state.tail[k];
    }
    TR(RC_BND_OUT_len_u, bout) = len_out; TR(RC_BND_OUT_len_s, bout) = len_out;
    for (int r = 0; r < 2; r++) {
        TR(RC_BND_OUT_lhs0 + r, bout) = lhs_z_all[(size_t)r * n_total + last];
        TR(RC_BND_OUT_rhs0 + r, bout) = rhs_z_all[(size_t)r * n_total + last];
    }
    zkw_mem_query lq;
    memset(&lq, 0, sizeof lq);
    if (!padded) lq = job.sorted_q[last];  // padding cycles reset the "previous" registers to zero
    u64 le[8];
    encode_mem_query(lq, le);
    TR(RC_BND_OUT_ts, bout) = lq.timestamp; TR(RC_BND_OUT_idx, bout) = lq.index; TR(RC_BND_OUT_page, bout) = lq.page;
    TR(RC_BND_OUT_es3, bout) = le[3]; TR(RC_BND_OUT_es4, bout) = le[4]; TR(RC_BND_OUT_es5, bout) = le[5];
    TR(RC_BND_OUT_es6, bout) = le[6]; TR(RC_BND_OUT_v4, bout) = le[7];
    TR(RC_BND_OUT_ptr, bout) = lq.value_is_pointer ? 1 : 0;
    // cnt after the last cycle = cnt of the last cycle's row C
    TR(RC_BND_OUT_cnt, bout) = TR(RC_C_cnt, (size_t)RC_ROW_C * RC_REGION_STRIDE(capacity) + capacity - 1);
    TR(RC_BND_OUT_completion, bout) = in->completion_flag ? 1 : 0;
    TR(RC_BND_OUT_w_end, bout) = inv_or_zero(len_out); TR(RC_BND_OUT_z_end, bout) = len_out == 0;
}

// a value row of the closed-form section: the bytes of limbs 5..7 into the lookup cells; VIN also the encoding elements es3..es6 of the limbs (memory_query.rs:60-110)
__device__ __forceinline__ void ram_value_row(u64* trace, size_t n_rows, size_t row, int v0, int b0, int e3) {
    zkw_mem_query q;
    memset(&q, 0, sizeof q);
    for (int k = 0; k < 8; k++) q.value[k] = (u32)TR(v0 + k, row);
    for (int l = 5; l < 8; l++)
        for (int k = 0; k < 4; k++) {
            const u64 b = (q.value[l] >> (8 * k)) & 0xFF;
            TR(b0 + 4 * (l - 5) + k, row) = b;  // (counted in the histogram by k_ram_fill_C)
        }
    if (e3 < 0) return;
    u64 e[8];
    encode_mem_query(q, e);
    for (int k = 0; k < 4; k++) TR(e3 + k, row) = e[3 + k];
}

// the first blocks of k_ram_fill_D (the tail kernel zeroes the rows BELOW the boundary rows): the boundary rows' cells zeroed, the
// register rows, then the closed-form section (closed_form_kernels.cuh) down to the PI row. Reads the last cycle's row C (an earlier kernel).
__device__ __forceinline__ void ram_boundary_block(const SynthJob& job, u32 capacity, size_t n_rows) {
    {
        u64* trace = job.trace;
        const size_t bnd = (size_t)RC_BOUNDARY_ROW(capacity);
        for (int k = threadIdx.x; k < (RC_G + RC_L) * RC_BOUNDARY_ROWS; k += CF_THREADS) TR(k / RC_BOUNDARY_ROWS, bnd + k % RC_BOUNDARY_ROWS) = 0;
    }
    __syncthreads();
    __shared__ u64 sh_oi[RAM_INPUT_ENC_LEN], sh_fi[RAM_FSM_ENC_LEN], sh_fo[RAM_FSM_ENC_LEN], sh_flags[2];
    if (threadIdx.x == 0) ram_fill_register_rows(job, capacity, n_rows);
    if (threadIdx.x == 64) {  // (one lane of each of the other three waves: the encoders run side by side)
        int m = put_queue12(job.first_inst->unsorted_queue_initial_state, sh_oi);
        m += put_queue12(job.first_inst->sorted_queue_initial_state, sh_oi + m);
        sh_oi[m] = job.first_inst->non_deterministic_bootloader_memory_snapshot_length;
    }
    if (threadIdx.x == 128) ram_encode_fsm(job.inst->hidden_fsm_input, sh_fi);
    if (threadIdx.x == 192) {
        ram_encode_fsm(job.inst->hidden_fsm_output, sh_fo);
        sh_flags[0] = job.inst->start_flag ? 1 : 0;
        sh_flags[1] = job.inst->completion_flag ? 1 : 0;
    }
    __syncthreads();
    const CfSpec S = ZKW_CF_SPEC(RC, rc, c_is_poseidon);
    const CfSources src = {sh_oi, sh_fi, sh_fo, sh_flags, nullptr};
    u64* trace = job.trace;
    cf_fill_block(S, trace, n_rows, (size_t)RC_BOUNDARY_ROW(capacity), src, [&](int rt, size_t row) {
        if (threadIdx.x != 0) return;
        if (rt == RC_ROW_VIN) ram_value_row(trace, n_rows, row, RC_VIN_VIN_v0, RC_VIN_VIN_v5_b0, RC_VIN_VIN_e3);
        if (rt == RC_ROW_VOUT) ram_value_row(trace, n_rows, row, RC_VOUT_VOUT_v0, RC_VOUT_VOUT_v5_b0, -1);
    });
}

// ------------------------------------------------------------------------------------------------
// Satisfiability check, generic over a spec (include/zkw_*_circuit_spec.h): the tables live in constant memory; a
// block stages 64 consecutive rows of one region (all 148 general + lookup columns) in LDS, then each lane
// interprets its row's constraints; Poseidon2 rows are recomputed from their 12 inputs.
static __constant__ rc_term c_terms[RC_NUM_TERMS] = RC_TERMS_INIT;
static __constant__ rc_constraint c_cons[RC_NUM_CONSTRAINTS] = RC_CONSTRAINTS_INIT;
static __constant__ uint16_t c_row_first[RC_NUM_ROW_TYPES + 1] = RC_ROW_FIRST_CONSTRAINT_INIT;
static __constant__ rc_term c_ds_terms[DS_NUM_TERMS] = DS_TERMS_INIT;
static __constant__ rc_constraint c_ds_cons[DS_NUM_CONSTRAINTS] = DS_CONSTRAINTS_INIT;
static __constant__ uint16_t c_ds_row_first[DS_NUM_ROW_TYPES + 1] = DS_ROW_FIRST_CONSTRAINT_INIT;
static __constant__ uint8_t c_ds_is_poseidon[DS_NUM_ROW_TYPES] = DS_ROW_IS_POSEIDON_INIT;
static __constant__ rc_link c_ds_links[DS_NUM_LINKS] = DS_LINKS_INIT;

struct SpecRam {  // RAMPermutation, circuit type 8
    static constexpr int G = RC_G, L = RC_L, ROWS_PER_CYCLE = RC_ROWS_PER_CYCLE, NUM_ROW_TYPES = RC_NUM_ROW_TYPES, NUM_LINKS = RC_NUM_LINKS;
    static constexpr int OFF_BIN = RC_ROWOFF_BND_IN, OFF_BOUT = RC_ROWOFF_BND_OUT;
    __device__ static const rc_term* terms() { return c_terms; }
    __device__ static const rc_constraint* cons() { return c_cons; }
    __device__ static const uint16_t* row_first() { return c_row_first; }
    __device__ static const uint8_t* is_poseidon() { return c_is_poseidon; }
    __device__ static const rc_link* links() { return c_links; }
};
ZKW_CF_TABLES(DS, ds)
struct SpecDecommitSorter {  // CodeDecommittmentsSorter, circuit type 2
    static constexpr int G = DS_G, L = DS_L, ROWS_PER_CYCLE = DS_ROWS_PER_CYCLE, NUM_ROW_TYPES = DS_NUM_ROW_TYPES, NUM_LINKS = DS_NUM_LINKS;
    static constexpr int OFF_BIN = DS_ROWOFF_BND_IN, OFF_BOUT = DS_ROWOFF_BND_OUT;
    __device__ static const rc_term* terms() { return c_ds_terms; }
    __device__ static const rc_constraint* cons() { return c_ds_cons; }
    __device__ static const uint16_t* row_first() { return c_ds_row_first; }
    __device__ static const uint8_t* is_poseidon() { return c_ds_is_poseidon; }
    __device__ static const rc_link* links() { return c_ds_links; }
    ZKW_CF_SPEC_MEMBERS(DS, ds)
};
static __constant__ rc_term c_es_terms[ES_NUM_TERMS] = ES_TERMS_INIT;
static __constant__ rc_constraint c_es_cons[ES_NUM_CONSTRAINTS] = ES_CONSTRAINTS_INIT;
static __constant__ uint16_t c_es_row_first[ES_NUM_ROW_TYPES + 1] = ES_ROW_FIRST_CONSTRAINT_INIT;
static __constant__ uint8_t c_es_is_poseidon[ES_NUM_ROW_TYPES] = ES_ROW_IS_POSEIDON_INIT;
static __constant__ rc_link c_es_links[ES_NUM_LINKS] = ES_LINKS_INIT;
ZKW_CF_TABLES(ES, es)
struct SpecEventsSorter {  // EventsSorter / L1MessagesSorter, circuit types 11 and 12
    static constexpr int G = ES_G, L = ES_L, ROWS_PER_CYCLE = ES_ROWS_PER_CYCLE, NUM_ROW_TYPES = ES_NUM_ROW_TYPES, NUM_LINKS = ES_NUM_LINKS;
    static constexpr int OFF_BIN = ES_ROWOFF_BND_IN, OFF_BOUT = ES_ROWOFF_BND_OUT;
    __device__ static const rc_term* terms() { return c_es_terms; }
    __device__ static const rc_constraint* cons() { return c_es_cons; }
    __device__ static const uint16_t* row_first() { return c_es_row_first; }
    __device__ static const uint8_t* is_poseidon() { return c_es_is_poseidon; }
    __device__ static const rc_link* links() { return c_es_links; }
    ZKW_CF_SPEC_MEMBERS(ES, es)
};

struct CheckResult {
    unsigned long long violations;
    unsigned long long first_bad;  // (kind << 56) | (index << 32) | row ; smallest code wins
};

__device__ __forceinline__ void flag_bad(CheckResult* res, u64 kind, u64 idx, u64 row) {
    atomicAdd(&res->violations, 1ull);
    atomicMin(&res->first_bad, (kind << 56) | (idx << 32) | row);
}

constexpr int CHK_ROWS = 64;

template <class S>
__device__ __forceinline__ size_t spec_row(int rt, u32 capacity, u32 i) {
    const size_t rs = RC_REGION_STRIDE(capacity);
    return rt < S::ROWS_PER_CYCLE ? (size_t)rt * rs + i : (size_t)S::ROWS_PER_CYCLE * rs + (rt - S::ROWS_PER_CYCLE);
}

template <class S>
static __global__ __launch_bounds__(64) void k_check_rows(const u64* __restrict__ trace, u32 capacity, size_t n_rows, CheckResult* res) {
    extern __shared__ __attribute__((aligned(16))) u64 tile[];  // [G + L][CHK_ROWS]
    constexpr int CHK_COLS = S::G + S::L;
    const int rt = blockIdx.y;  // row type; boundary row types are handled by block x == 0 only
    const bool per_cycle = rt < S::ROWS_PER_CYCLE;
    const u32 n_in_region = per_cycle ? capacity : 1;
    const u32 i = blockIdx.x * CHK_ROWS + threadIdx.x;
    if (blockIdx.x * CHK_ROWS >= n_in_region) return;
    const bool live = i < n_in_region;
    const size_t row = spec_row<S>(rt, capacity, per_cycle ? i : 0);
    for (int c = 0; c < CHK_COLS; c++) {
        u64 v = live ? TR(c, row) : 0;
        tile[c * CHK_ROWS + threadIdx.x] = v;
        if (live && (v >= gl::P || (c >= S::G && v > 255))) flag_bad(res, 3, c, row);
    }
    __syncthreads();
    if (!live) return;
#define CELLV(c) tile[(c) * CHK_ROWS + threadIdx.x]
    for (int k = S::row_first()[rt]; k < S::row_first()[rt + 1]; k++) {
        const rc_constraint cn = S::cons()[k];
        u64 acc = 0;
        for (int t = 0; t < cn.n_terms; t++) {
            const rc_term* tm = &S::terms()[cn.first_term + t];  // read in place: a local copy indexed by f would live in scratch
            u64 v = tm->coef;
            for (int f = 0; f < tm->nf; f++) v = gl::mul(v, CELLV(tm->f[f]));
            acc = gl::add(acc, v);
        }
        if (gl::canon(acc) != 0) flag_bad(res, 1, k, row);
    }
    if (S::is_poseidon()[rt]) {
        u64 s[12];
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = CELLV(k);
        int pos = 12, r = 0;
        bool ok = true;
        p2::external(s);
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
            p2::full_round(s, r);
#pragma unroll
            for (int j = 0; j < 12; j++) ok &= gl::canon(s[j]) == CELLV(pos + j);
            pos += 12;
        }
        for (int k = 0; k < P2_PARTIAL_ROUNDS; k++, r++) {
            s[0] = gl::pow7(gl::add_canon(s[0], p2::rc_at(12 * r)));
            ok &= gl::canon(s[0]) == CELLV(pos);
            pos++;
            p2::internal(s);
        }
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
            p2::full_round(s, r);
#pragma unroll
            for (int j = 0; j < 12; j++) ok &= gl::canon(s[j]) == CELLV(pos + j);
            pos += 12;
        }
        if (!ok) flag_bad(res, 2, 0, row);
    }
#undef CELLV
}

// copy links: one lane per cycle walks the link table (both cells are read coalesced across lanes)
template <class S>
static __global__ __launch_bounds__(256) void k_check_links(const u64* __restrict__ trace, u32 capacity, size_t n_rows, CheckResult* res) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= capacity) return;
    const size_t rs = RC_REGION_STRIDE(capacity), bnd = (size_t)S::ROWS_PER_CYCLE * rs;
    for (int l = 0; l < S::NUM_LINKS; l++) {
        const rc_link k = S::links()[l];
        if (k.kind == 3) {
            if (i == capacity - 1 && TR(k.col_a, bnd + S::OFF_BOUT) != TR(k.col_b, (size_t)k.row_b * rs + i))
                flag_bad(res, 4, l, bnd + S::OFF_BOUT);
            continue;
        }
        if (k.kind == 4) {  // a boundary row's cell equals a BND_OUT cell
            if (i == 0 && TR(k.col_a, spec_row<S>(k.row_a, capacity, 0)) != TR(k.col_b, bnd + S::OFF_BOUT))
                flag_bad(res, 4, l, spec_row<S>(k.row_a, capacity, 0));
            continue;
        }
        if (k.kind == 5) {  // a boundary row's cell equals a cell of another boundary row
            if (i == 0 && TR(k.col_a, spec_row<S>(k.row_a, capacity, 0)) != TR(k.col_b, spec_row<S>(k.row_b, capacity, 0)))
                flag_bad(res, 4, l, spec_row<S>(k.row_a, capacity, 0));
            continue;
        }
        const size_t ra = (size_t)k.row_a * rs + i;
        const u64 a = TR(k.col_a, ra);
        u64 b;
        if (k.kind == 0) b = TR(k.col_b, (size_t)k.row_b * rs + i);
        else if (k.kind == 1) b = i ? TR(k.col_b, (size_t)k.row_b * rs + i - 1) : TR(k.bin_col, bnd + S::OFF_BIN);
        else b = TR(k.col_b, bnd + S::OFF_BIN);
        if (a != b) flag_bad(res, 4, l, ra);
    }
}

// lookup columns: histogram of every cell (all n_rows), padding rows must be zero in every column
template <class S>
static __global__ __launch_bounds__(256) void k_check_lookups(const u64* __restrict__ trace, u32 capacity, size_t n_rows,
                                                       u32* __restrict__ hist, CheckResult* res) {
    __shared__ u32 sh_hist[256];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t rs = RC_REGION_STRIDE(capacity), pad0 = (size_t)S::ROWS_PER_CYCLE * rs + (S::NUM_ROW_TYPES - S::ROWS_PER_CYCLE);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        for (int c = S::G; c < S::G + S::L; c++) {
            const u64 v = TR(c, r);
            if (v > 255) flag_bad(res, 3, c, r); else atomicAdd(&sh_hist[v], 1u);
        }
        if (r >= pad0 || (r < (size_t)S::ROWS_PER_CYCLE * rs && r % rs >= capacity))  // tail padding and the region gaps
            for (int c = 0; c < S::G; c++)
                if (TR(c, r) != 0) { flag_bad(res, 6, c, r); break; }
    }
    __syncthreads();
    if (sh_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh_hist[threadIdx.x]);
}
static __global__ void k_check_mult(const u64* __restrict__ trace, size_t n_rows, int mult_col, const u32* __restrict__ hist, CheckResult* res) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const u64 want = r < 256 ? hist[r] : 0;
        if (TR(mult_col, r) != want) flag_bad(res, 5, 0, r);
    }
}

#undef TR
}  // namespace zkw
