// decommit_sorter_circuit_kernels.cuh — synthesis of the CodeDecommittmentsSorter trace ("zkw trace v2", circuit
// type 2, include/zkw_decommit_sorter_circuit_spec.h) on gfx950.
//
// Counterpart of ZkSyncBaseLayerCircuit::synthesis for that instance type (circuit_definitions/src/circuit_definitions/
// base_layer/mod.rs:286-323, wrapper base_layer/sort_code_decommits.rs:28-39); the witness it materialises is the
// output of compute_decommitts_sorter_circuit_snapshots (src/witness/individual_circuits/sort_decommit_requests.rs:20-420).
//
// Same plan as the RAM circuit: one lane per cycle, one kernel per row type, region-major rows, every value a pure
// function of item i, item i-1, the instance record and block-wide arrays — the registers a circuit body would carry
// (queue heads, accumulators, the open hash group, the deduplicated queue) are read back: the deduplicated queue after
// p pushes is dedup_tails[p-1], the open group's request is dedup_enc[f-1] with f = fresh requests among sorted[0, idx)
// (one prefix array), and the number of pushes before a cycle is max(f - 1, 0) because a group is pushed when the
// next one opens. The cells of a row are scattered through the generated DS_FILL_<row> lists.
#pragma once
#include "scan_kernels.cuh"
#include "decommit_kernels.cuh"
#include "ram_circuit_kernels.cuh"

namespace zkw {

struct DsSynthJob {
    const zkw_decommit_sorter_instance* inst;
    const zkw_decommit_query* sorted_q;  // block-wide arrays of the builder, indexed by item
    const u64 *unsorted_enc, *sorted_enc;      // [n][8]
    const u64 *unsorted_tails, *sorted_tails;  // [n][12]
    const u64 *dedup_enc, *dedup_tails;        // [n_dedup][8], [n_dedup][12]
    const u32* fresh_prefix;                   // [n + 1]: fresh requests among sorted[0, k)
    const u64* challenges;                     // [2][9]
    const u64 *lhs_z, *rhs_z;                  // [2][n]
    u64 n_block;
    u64 rq_tail_in[12];                        // deduplicated queue before the block
    u32 rq_len_in;
    u64* trace;
    u32* hist;  // [256]
    const u64* public_input;  // [4] commitment of the instance's closed-form input (k_ds_commitments): not written, the closed-form section derives it
    const zkw_decommit_sorter_instance* first_inst;  // the block's first instance (the shared observable input)
    u32 tail_clean;  // the slot already holds this layout (same circuit, capacity, rows): every cell that is zero in EVERY trace of the layout (the padding rows, the gap rows of a region, the columns a row type does not use, multiplicity rows >= 256) is still zero: the fills skip those stores
};

struct DsVars {
#define X(n) u64 n;
    DS_VARS(X)
#undef X
};

#define TR(col, row) trace[(size_t)(col) * n_rows + (row)]
#define DS_COLS8(ROW, v) {DS_##ROW##_##v##0, DS_##ROW##_##v##1, DS_##ROW##_##v##2, DS_##ROW##_##v##3, DS_##ROW##_##v##4, DS_##ROW##_##v##5, \
                          DS_##ROW##_##v##6, DS_##ROW##_##v##7}
#define DS_COLS12(ROW, v) {DS_##ROW##_##v##0, DS_##ROW##_##v##1, DS_##ROW##_##v##2, DS_##ROW##_##v##3, DS_##ROW##_##v##4, DS_##ROW##_##v##5, \
                           DS_##ROW##_##v##6, DS_##ROW##_##v##7, DS_##ROW##_##v##8, DS_##ROW##_##v##9, DS_##ROW##_##v##10, DS_##ROW##_##v##11}
#define DS_SET8(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; \
    dst.pfx##4 = (src)[4]; dst.pfx##5 = (src)[5]; dst.pfx##6 = (src)[6]; dst.pfx##7 = (src)[7]; } while (0)
#define DS_SET12(dst, pfx, src) do { DS_SET8(dst, pfx, src); dst.pfx##8 = (src)[8]; dst.pfx##9 = (src)[9]; dst.pfx##10 = (src)[10]; \
    dst.pfx##11 = (src)[11]; } while (0)
#define DS_BYTES(dst, pfx, x) do { const u32 _x = (u32)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; dst.pfx##_b3 = _x >> 24; } while (0)
#define DS_ISZ(dst, w, z, a, b) do { const u64 _d = gl::canon(gl::sub((a), (b))); dst.z = _d == 0; dst.w = _d ? gl::inv(_d) : 0; } while (0)

template <class Q>
__device__ __forceinline__ void ds_encode(const Q& q, u64 e[8]) {  // decommittment_request.rs:9-74
    e[0] = (u64)q.hash[0] | ((u64)(q.memory_page & 0xFFFFFF) << 32);
    e[1] = (u64)q.hash[1] | ((u64)(q.memory_page >> 24) << 32) | ((u64)(q.timestamp & 0xFFFF) << 40);
    e[2] = (u64)q.hash[2] | ((u64)(q.timestamp >> 16) << 32) | ((u64)(q.is_fresh ? 1 : 0) << 48);
#pragma unroll
    for (int k = 3; k < 8; k++) e[k] = q.hash[k];
}

// the fields of a request the circuit looks at (kept in registers; zkw_decommit_query itself carries padding)
struct DsReq {
    u32 hash[8];
    u32 timestamp, memory_page, is_fresh;
};
__device__ __forceinline__ DsReq ds_req_zero() {
    DsReq r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.hash[k] = 0;
    r.timestamp = 0; r.memory_page = 0; r.is_fresh = 0;
    return r;
}
__device__ __forceinline__ DsReq ds_req_load(const zkw_decommit_query* q) {
    const uint4* s4 = reinterpret_cast<const uint4*>(q);
    const uint4 a = s4[0], b = s4[1], c = s4[2];
    DsReq r;
    r.hash[0] = a.x; r.hash[1] = a.y; r.hash[2] = a.z; r.hash[3] = a.w; r.hash[4] = b.x; r.hash[5] = b.y; r.hash[6] = b.z; r.hash[7] = b.w;
    r.timestamp = c.x; r.memory_page = c.y; r.is_fresh = (c.z >> 16) & 0xFF;  // decommitted_length u16, is_fresh u8
    return r;
}

// the registers entering cycle 0 of an instance = its hidden FSM input (or the observable input on the first one)
struct DsRegsIn {
    const u64 *uh, *sh;  // [12]
    u64 rh[12], ge[8], lhs[2], rhs[2];
    u32 len, len_r, gvalid;
    DsReq pq;  // previous request's key fields (zeros on the first instance)
};
__device__ __forceinline__ void ds_regs_in(const DsSynthJob& job, DsRegsIn& r) {
    const zkw_decommit_sorter_instance* in = job.inst;
    const zkw_decommit_sorter_fsm& f = in->hidden_fsm_input;
    const bool start = in->start_flag != 0;
    r.uh = start ? in->initial_queue_state.head : f.initial_queue_state.head;
    r.sh = start ? in->sorted_queue_initial_state.head : f.sorted_queue_state.head;
    r.len = start ? in->initial_queue_state.length : f.initial_queue_state.length;
#pragma unroll
    for (int k = 0; k < 12; k++) r.rh[k] = start ? job.rq_tail_in[k] : f.final_queue_state.tail[k];
    r.len_r = start ? job.rq_len_in : f.final_queue_state.length;
#pragma unroll
    for (int k = 0; k < 2; k++) { r.lhs[k] = start ? 1 : f.lhs_accumulator[k]; r.rhs[k] = start ? 1 : f.rhs_accumulator[k]; }
    r.gvalid = start ? 0 : 1;
    r.pq = ds_req_zero();
    r.pq.timestamp = f.previous_packed_key[0];
#pragma unroll
    for (int k = 0; k < 8; k++) r.pq.hash[k] = f.previous_packed_key[1 + k];
    r.pq.memory_page = f.previous_record.memory_page;
#pragma unroll
    for (int k = 0; k < 8; k++) r.ge[k] = 0;
    if (!start) {  // the open group's first request: (hash, page, first_encountered_timestamp, fresh)
        DsReq g = r.pq;  // hash and page of the previous record
        g.timestamp = f.first_encountered_timestamp;
        g.is_fresh = 1;
        ds_encode(g, r.ge);
    }
}

// what a cycle needs from its neighbourhood: this request, the previous one, group / queue positions
struct DsCycle {
    bool can_pop;
    DsReq q, pq;
    u32 p_gvalid, fresh_before;  // fresh requests among sorted[0, idx)
    u64 pushes_before;           // pushes into the deduplicated queue before this cycle (from the block's start)
    size_t last_popped;          // index of the last item popped before this cycle (valid when i > 0)
};
__device__ __forceinline__ void ds_cycle(const DsSynthJob& job, const DsRegsIn& ri, u32 i, DsCycle& c) {
    const zkw_decommit_sorter_instance* in = job.inst;
    const size_t first = in->first_item, m = in->num_items;
    c.can_pop = i < m;
    c.q = ds_req_zero();
    c.pq = ds_req_zero();
    if (c.can_pop) c.q = ds_req_load(job.sorted_q + first + i);
    if (i == 0) c.pq = ri.pq;
    else if (i - 1 < m) c.pq = ds_req_load(job.sorted_q + first + i - 1);
    c.p_gvalid = i == 0 ? ri.gvalid : 1;
    c.fresh_before = job.fresh_prefix[first + (i < m ? i : m)];
    c.pushes_before = c.fresh_before ? c.fresh_before - 1 : 0;
    c.last_popped = first + (i - 1 < m ? i - 1 : m - 1);
}
__device__ __forceinline__ void ds_prev_result_queue(const DsSynthJob& job, const DsRegsIn& ri, const DsCycle& c, u64 rh[12], u64 ge[8]) {
    // before any request of the block the registers are the instance's inputs (only the first instance can see that)
    if (c.pushes_before) for (int k = 0; k < 12; k++) rh[k] = job.dedup_tails[12 * (c.pushes_before - 1) + k];
    else for (int k = 0; k < 12; k++) rh[k] = job.rq_tail_in[k];
    if (c.fresh_before) for (int k = 0; k < 8; k++) ge[k] = job.dedup_enc[8 * (size_t)(c.fresh_before - 1) + k];
    else for (int k = 0; k < 8; k++) ge[k] = ri.ge[k];
}

// ------------------------------------------------------------------------------------------------
// Poseidon2 rows: WHICH 0 = PU (pop unsorted), 1 = PS (pop sorted), 2 = PR (push into the deduplicated queue)
template <int WHICH>
static __device__ __forceinline__ void k_ds_fill_poseidon(const VB& vb, const DsSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) sh_hist[t] = 0;
    __syncthreads();
    const DsSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = DS_REGION_STRIDE(capacity);
    constexpr int ROW = WHICH == 0 ? DS_ROW_PU : (WHICH == 1 ? DS_ROW_PS : DS_ROW_PR);
    u64* trace = job.trace;
    if (i < capacity) {
        const size_t row = (size_t)ROW * rs + i;
        DsRegsIn ri;
        ds_regs_in(job, ri);
        DsCycle c;
        ds_cycle(job, ri, i, c);
        const size_t first = job.inst->first_item;
        u64 s[12];
        if (WHICH == 2) {
            u64 rh[12], ge[8];
            ds_prev_result_queue(job, ri, c, rh, ge);
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = ge[k];
#pragma unroll
            for (int k = 0; k < 4; k++) s[8 + k] = rh[8 + k];
        } else {
            const u64* enc = WHICH == 0 ? job.unsorted_enc : job.sorted_enc;
            const u64* tails = WHICH == 0 ? job.unsorted_tails : job.sorted_tails;
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = c.can_pop ? enc[8 * (first + i) + k] : 0;
            const u64* ph = i == 0 ? (WHICH == 0 ? ri.uh : ri.sh) : tails + 12 * c.last_popped;
#pragma unroll
            for (int k = 0; k < 4; k++) s[8 + k] = ph[8 + k];
        }
        fill_flattened_poseidon(trace, n_rows, row, s);
        if (WHICH == 1) {  // the range checks of hash limbs 3..6 ride here
#pragma unroll
            for (int k = 3; k < 7; k++) {
                put_bytes(trace, n_rows, row, DS_PS_h3_b0 + 4 * (k - 3), c.q.hash[k]);
                hist_bytes(sh_hist, c.q.hash[k]);
            }
            if (!job.tail_clean) for (int col = DS_G + 16; col < DS_G + DS_L; col++) TR(col, row) = 0;
        } else {
            if (!job.tail_clean) for (int col = DS_G; col < DS_G + DS_L; col++) TR(col, row) = 0;
        }
    } else if (i < rs) {
        if (!job.tail_clean) zero_gap_row(trace, n_rows, (size_t)ROW * rs + i);
    }
    if (WHICH == 1 && vb.x == 0 && threadIdx.x == 0) {  // the closed-form section's lookup cells (GIN / GOUT: bytes of the FSM records' page and first-encountered timestamp)
        hist_bytes(sh_hist, job.inst->hidden_fsm_input.previous_record.memory_page);
        hist_bytes(sh_hist, job.inst->hidden_fsm_input.first_encountered_timestamp);
        hist_bytes(sh_hist, job.inst->hidden_fsm_output.previous_record.memory_page);  // GOUT: the handed-over group's page and first timestamp
        hist_bytes(sh_hist, job.inst->hidden_fsm_output.first_encountered_timestamp);
    }
    hist_flush(sh_hist, job.hist);
}

// ------------------------------------------------------------------------------------------------
// General rows A..D. ROW selects which row's cells are stored; the values a row does not hold are dead code.
#define DS_XC(col, v) TR(col, row) = cur.v;
#define DS_XP(col, v) TR(col, row) = prev.v;
#define DS_XG(col, v) TR(col, row) = glob.v;

template <int ROW>
static __device__ __forceinline__ void k_ds_fill_row(const VB& vb, const DsSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const DsSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = DS_REGION_STRIDE(capacity);
    u64* trace = job.trace;
    if (i < capacity) {
        const size_t row = (size_t)ROW * rs + i, n = job.n_block;
        DsRegsIn ri;
        ds_regs_in(job, ri);
        DsCycle c;
        ds_cycle(job, ri, i, c);
        const size_t first = job.inst->first_item, m = job.inst->num_items, idx = first + i;
        DsVars cur, prev, glob;
        const u64 can_pop = c.can_pop ? 1 : 0;
        cur.can_pop = can_pop;
        u64 eu[8], es[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { eu[k] = c.can_pop ? job.unsorted_enc[8 * idx + k] : 0; es[k] = c.can_pop ? job.sorted_enc[8 * idx + k] : 0; }
        DS_SET8(cur, eu, eu);
        DS_SET8(cur, es, es);
        // this request and the previous one
        const DsReq& q = c.q;
        const DsReq& pq = c.pq;
        cur.h0 = q.hash[0]; cur.h1 = q.hash[1]; cur.h2 = q.hash[2]; cur.page = q.memory_page; cur.ts = q.timestamp;
        cur.fresh = q.is_fresh ? 1 : 0;
        prev.h0 = pq.hash[0]; prev.h1 = pq.hash[1]; prev.h2 = pq.hash[2]; prev.es3 = pq.hash[3]; prev.es4 = pq.hash[4];
        prev.es5 = pq.hash[5]; prev.es6 = pq.hash[6]; prev.es7 = pq.hash[7]; prev.page = pq.memory_page; prev.ts = pq.timestamp;
        prev.gvalid = c.p_gvalid;
        cur.gvalid = c.p_gvalid | (u32)can_pop;
        DS_BYTES(cur, h0, q.hash[0]); DS_BYTES(cur, h1, q.hash[1]); DS_BYTES(cur, h2, q.hash[2]); DS_BYTES(cur, h7, q.hash[7]);
        DS_BYTES(cur, page, q.memory_page); DS_BYTES(cur, ts, q.timestamp);
        // key - previous key, nine u32 limbs from the least significant: ts, h0..h7
        {
            const u32 c9[9] = {q.timestamp, q.hash[0], q.hash[1], q.hash[2], q.hash[3], q.hash[4], q.hash[5], q.hash[6], q.hash[7]};
            const u32 p9[9] = {pq.timestamp, pq.hash[0], pq.hash[1], pq.hash[2], pq.hash[3], pq.hash[4], pq.hash[5], pq.hash[6], pq.hash[7]};
            u32 d[9], bw[9], borrow = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const u64 t = (u64)c9[k] - (u64)p9[k] - borrow;  // wraps below zero
                bw[k] = (u32)(t >> 63);
                d[k] = (u32)t;
                borrow = bw[k];
            }
            cur.d0 = d[0]; cur.d1 = d[1]; cur.d2 = d[2]; cur.d3 = d[3]; cur.d4 = d[4]; cur.d5 = d[5]; cur.d6 = d[6]; cur.d7 = d[7]; cur.d8 = d[8];
            cur.bw0 = bw[0]; cur.bw1 = bw[1]; cur.bw2 = bw[2]; cur.bw3 = bw[3]; cur.bw4 = bw[4]; cur.bw5 = bw[5]; cur.bw6 = bw[6];
            cur.bw7 = bw[7]; cur.bw8 = bw[8];
            DS_BYTES(cur, d0, d[0]); DS_BYTES(cur, d1, d[1]); DS_BYTES(cur, d2, d[2]); DS_BYTES(cur, d3, d[3]); DS_BYTES(cur, d4, d[4]);
            DS_BYTES(cur, d5, d[5]); DS_BYTES(cur, d6, d[6]); DS_BYTES(cur, d7, d[7]); DS_BYTES(cur, d8, d[8]);
        }
        // hash equality with the previous request and the group logic
        bool same = true;
#pragma unroll
        for (int k = 0; k < 8; k++) same &= q.hash[k] == pq.hash[k];
        const bool new_group = c.can_pop && !(same && c.p_gvalid), push = new_group && c.p_gvalid;
        cur.new_group = new_group; cur.push = push;
        if (ROW == DS_ROW_C) {
            // eight "inverse or zero" witnesses from ONE field inversion (Montgomery's trick; zeros are replaced by 1)
            u64 dlt[8], pre[8], inv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) dlt[k] = gl::canon(gl::sub((u64)q.hash[k], (u64)pq.hash[k]));
            u64 acc = 1;
#pragma unroll
            for (int k = 0; k < 8; k++) { pre[k] = acc; acc = gl::mul(acc, dlt[k] ? dlt[k] : 1); }
            u64 ia = gl::inv(acc);
#pragma unroll
            for (int k = 7; k >= 0; k--) { inv[k] = dlt[k] ? gl::canon(gl::mul(ia, pre[k])) : 0; ia = gl::mul(ia, dlt[k] ? dlt[k] : 1); }
            cur.w_e0 = inv[0]; cur.w_e1 = inv[1]; cur.w_e2 = inv[2]; cur.w_e3 = inv[3]; cur.w_e4 = inv[4]; cur.w_e5 = inv[5]; cur.w_e6 = inv[6]; cur.w_e7 = inv[7];
            cur.z_e0 = dlt[0] == 0; cur.z_e1 = dlt[1] == 0; cur.z_e2 = dlt[2] == 0; cur.z_e3 = dlt[3] == 0; cur.z_e4 = dlt[4] == 0;
            cur.z_e5 = dlt[5] == 0; cur.z_e6 = dlt[6] == 0; cur.z_e7 = dlt[7] == 0;
            cur.same_a = cur.z_e0 & cur.z_e1 & cur.z_e2 & cur.z_e3;
            cur.same_hash = same;
            // deduplicated queue: the PR row (written earlier on this stream) holds the pushed state
            u64 rh[12], ge[8], ro[12], o[12];
            ds_prev_result_queue(job, ri, c, rh, ge);
            const size_t rPR = (size_t)DS_ROW_PR * rs + i;
            constexpr int RO[12] = DS_COLS12(PR, ro);
#pragma unroll
            for (int k = 0; k < 12; k++) { ro[k] = TR(RO[k], rPR); o[k] = push ? ro[k] : rh[k]; }
            DS_SET12(cur, ro, ro); DS_SET12(prev, rh, rh); DS_SET12(cur, rh, o);
        }
        if (ROW == DS_ROW_A) {
            glob.c0_1 = job.challenges[1]; glob.c0_2 = job.challenges[2]; glob.c0_3 = job.challenges[3]; glob.c0_4 = job.challenges[4];
            glob.c0_5 = job.challenges[5]; glob.c0_6 = job.challenges[6]; glob.c0_7 = job.challenges[7]; glob.c0_8 = job.challenges[8];
            glob.c1_1 = job.challenges[10]; glob.c1_2 = job.challenges[11]; glob.c1_3 = job.challenges[12]; glob.c1_4 = job.challenges[13];
            glob.c1_5 = job.challenges[14]; glob.c1_6 = job.challenges[15]; glob.c1_7 = job.challenges[16]; glob.c1_8 = job.challenges[17];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const u64* ch = job.challenges + 9 * r;
                u64 lc = gl::add(ch[8], eu[0]), rc = gl::add(ch[8], es[0]);
#pragma unroll
                for (int k = 1; k < 8; k++) { lc = gl::add(lc, gl::mul(eu[k], ch[k])); rc = gl::add(rc, gl::mul(es[k], ch[k])); }
                const u64 pl = i == 0 ? ri.lhs[r] : job.lhs_z[(size_t)r * n + c.last_popped];
                const u64 pr = i == 0 ? ri.rhs[r] : job.rhs_z[(size_t)r * n + c.last_popped];
                const u64 nl = gl::canon(gl::mul(pl, lc)), nr = gl::canon(gl::mul(pr, rc));
                lc = gl::canon(lc); rc = gl::canon(rc);
                if (r == 0) { cur.lc0 = lc; cur.rc0 = rc; cur.nl0 = nl; cur.nr0 = nr; prev.lhs0 = pl; prev.rhs0 = pr; cur.lhs0 = can_pop ? nl : pl; cur.rhs0 = can_pop ? nr : pr; }
                else { cur.lc1 = lc; cur.rc1 = rc; cur.nl1 = nl; cur.nr1 = nr; prev.lhs1 = pl; prev.rhs1 = pr; cur.lhs1 = can_pop ? nl : pl; cur.rhs1 = can_pop ? nr : pr; }
            }
        }
        if (ROW == DS_ROW_D) {
            const u64 p_len = (u64)ri.len - (i < m ? i : m);
            prev.len_u = p_len; prev.len_s = p_len;
            cur.w_lu = p_len ? gl::inv(p_len) : 0; cur.z_lu = p_len == 0; cur.w_ls = cur.w_lu; cur.z_ls = cur.z_lu;
            cur.len_u = p_len - can_pop; cur.len_s = p_len - can_pop;
            prev.len_r = (u64)job.rq_len_in + c.pushes_before;
            cur.len_r = prev.len_r + (push ? 1 : 0);
            u64 rh[12], ge[8], o8[8];
            ds_prev_result_queue(job, ri, c, rh, ge);
#pragma unroll
            for (int k = 0; k < 8; k++) o8[k] = new_group ? es[k] : ge[k];
            DS_SET8(prev, ge, ge); DS_SET8(cur, ge, o8);
            // queue heads: the Poseidon2 rows hold the popped states
            const size_t rPU = (size_t)DS_ROW_PU * rs + i, rPS = (size_t)DS_ROW_PS * rs + i;
            u64 uo[12], so[12], pu[12], ps[12], ou[12], os[12];
            constexpr int UO[12] = DS_COLS12(PU, uo), SO[12] = DS_COLS12(PS, so);
#pragma unroll
            for (int k = 0; k < 12; k++) {
                uo[k] = TR(UO[k], rPU); so[k] = TR(SO[k], rPS);
                pu[k] = i == 0 ? ri.uh[k] : job.unsorted_tails[12 * c.last_popped + k];
                ps[k] = i == 0 ? ri.sh[k] : job.sorted_tails[12 * c.last_popped + k];
                ou[k] = c.can_pop ? uo[k] : pu[k]; os[k] = c.can_pop ? so[k] : ps[k];
            }
            DS_SET12(cur, uo, uo); DS_SET12(cur, so, so); DS_SET12(prev, uh, pu); DS_SET12(prev, sh, ps); DS_SET12(cur, uh, ou); DS_SET12(cur, sh, os);
        }
        // scatter this row's cells; unused general slots and lookup cells are zero
        if (ROW == DS_ROW_A) { DS_FILL_A(DS_XC, DS_XP, DS_XG, DS_XC) }
        if (ROW == DS_ROW_B) { DS_FILL_B(DS_XC, DS_XP, DS_XG, DS_XC) }
        if (ROW == DS_ROW_C) { DS_FILL_C(DS_XC, DS_XP, DS_XG, DS_XC) }
        if (ROW == DS_ROW_D) { DS_FILL_D(DS_XC, DS_XP, DS_XG, DS_XC) }
        constexpr int NS = ROW == DS_ROW_A ? DS_NSLOTS_A : (ROW == DS_ROW_B ? DS_NSLOTS_B : (ROW == DS_ROW_C ? DS_NSLOTS_C : DS_NSLOTS_D));
        constexpr int NL = ROW == DS_ROW_A ? DS_NLOOK_A : (ROW == DS_ROW_B ? DS_NLOOK_B : (ROW == DS_ROW_C ? DS_NLOOK_C : DS_NLOOK_D));
        if (!job.tail_clean) for (int col = NS; col < DS_G; col++) TR(col, row) = 0;
        if (!job.tail_clean) for (int col = DS_G + NL; col < DS_G + DS_L; col++) TR(col, row) = 0;
        for (int col = DS_G; col < DS_G + NL; col++) atomicAdd(&sh_hist[(u32)TR(col, row) & 0xFF], 1u);
    } else if (i < rs) {
        if (!job.tail_clean) zero_gap_row(trace, n_rows, (size_t)ROW * rs + i);
    }
    hist_flush(sh_hist, job.hist);
}

// the zero padding below the boundary rows and the multiplicity column (see k_ram_fill_tail)
constexpr int DS_BOUNDARY_ROWS = (DS_NUM_ROW_TYPES - DS_ROWS_PER_CYCLE + 1) & ~1;  // register rows, PI, flush rows, the closed-form section (rounded up to even: 16-byte stores below)
__device__ __forceinline__ void ds_boundary_block(const DsSynthJob& job, u32 capacity, size_t n_rows);
static __device__ __forceinline__ void k_ds_fill_tail(const VB& vb, const DsSynthJob* __restrict__ jobs, u32 n_jobs, u32 capacity, size_t n_rows) {
    // 1-D grid: the first n_jobs blocks fill the boundary rows of one trace each (dispatched first and at raised priority: a chain of a dozen
    // dependent permutations that the other blocks' stores hide), then (DS_G + DS_L + 1) * TAIL_CHUNKS blocks per trace
    if (vb.x < n_jobs) {
        __builtin_amdgcn_s_setprio(3);
        ds_boundary_block(jobs[vb.x], capacity, n_rows);
        return;
    }
    constexpr u32 PER_JOB = (DS_G + DS_L + 1) * TAIL_CHUNKS;
    const u32 bid = (vb.x - n_jobs) % PER_JOB;
    const DsSynthJob& job = jobs[(vb.x - n_jobs) / PER_JOB];
    u64* trace = job.trace;
    const int col = bid / TAIL_CHUNKS, ch = bid % TAIL_CHUNKS;
    if (col < DS_G + DS_L) {
        if (job.tail_clean) return;
        const size_t bnd = (size_t)DS_BOUNDARY_ROW(capacity) + DS_BOUNDARY_ROWS;
        const size_t n_pairs = (n_rows - bnd) / 2;
        const size_t per = (n_pairs + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
        ulonglong2* c2 = reinterpret_cast<ulonglong2*>(trace + (size_t)col * n_rows + bnd);
        const ulonglong2 z = make_ulonglong2(0, 0);
        for (size_t k = lo + threadIdx.x; k < hi; k += 256) c2[k] = z;
        return;
    }
    u64* mlt = trace + (size_t)DS_MULT_COL * n_rows;
    const size_t per = (n_rows + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_rows ? lo + per : n_rows;
    for (size_t r = lo + threadIdx.x; r < (job.tail_clean && hi > 256 ? (lo < 256 ? 256 : lo) : hi); r += 256) {  // (a clean slot: rows >= 256 of the column are still zero)
        u64 v = 0;
        if (r < 256) {
            v = job.hist[r];
            if (r == 0) v += (u64)DS_L * n_rows - (u64)DS_LOOKUPS_PER_CYCLE * capacity - 8 - 4 * DS_CF_NUM_BYTES;  // (GIN's 8 byte cells and GOUT's, counted in job.hist by k_ds_fill_poseidon<1>)
        }
        mlt[r] = v;
    }
}

// runs after everything else (same stream): BND_IN, BND_OUT, the flush permutation PF, PI
__device__ __forceinline__ void ds_fill_register_rows(const DsSynthJob& job, u32 capacity, size_t n_rows) {
    u64* trace = job.trace;
    const zkw_decommit_sorter_instance* in = job.inst;
    const size_t rs = DS_REGION_STRIDE(capacity), bnd = (size_t)DS_BOUNDARY_ROW(capacity);
    DsRegsIn ri;
    ds_regs_in(job, ri);
    DsVars cur, glob;
    glob.c0_1 = job.challenges[1]; glob.c0_2 = job.challenges[2]; glob.c0_3 = job.challenges[3]; glob.c0_4 = job.challenges[4];
    glob.c0_5 = job.challenges[5]; glob.c0_6 = job.challenges[6]; glob.c0_7 = job.challenges[7]; glob.c0_8 = job.challenges[8];
    glob.c1_1 = job.challenges[10]; glob.c1_2 = job.challenges[11]; glob.c1_3 = job.challenges[12]; glob.c1_4 = job.challenges[13];
    glob.c1_5 = job.challenges[14]; glob.c1_6 = job.challenges[15]; glob.c1_7 = job.challenges[16]; glob.c1_8 = job.challenges[17];
    {  // BND_IN: the registers at cycle -1
        const size_t row = bnd + DS_ROWOFF_BND_IN;
        DS_SET12(cur, uh, ri.uh); DS_SET12(cur, sh, ri.sh); DS_SET12(cur, rh, ri.rh);
        cur.len_u = ri.len; cur.len_s = ri.len; cur.len_r = ri.len_r;
        cur.lhs0 = ri.lhs[0]; cur.lhs1 = ri.lhs[1]; cur.rhs0 = ri.rhs[0]; cur.rhs1 = ri.rhs[1];
        cur.ts = ri.pq.timestamp; cur.page = ri.pq.memory_page; cur.h0 = ri.pq.hash[0]; cur.h1 = ri.pq.hash[1]; cur.h2 = ri.pq.hash[2];
        cur.es3 = ri.pq.hash[3]; cur.es4 = ri.pq.hash[4]; cur.es5 = ri.pq.hash[5]; cur.es6 = ri.pq.hash[6]; cur.es7 = ri.pq.hash[7];
        cur.gvalid = ri.gvalid;
        DS_SET8(cur, ge, ri.ge);
        DS_FILL_BND_IN(DS_XC, DS_XP, DS_XG, DS_XC)
        for (int col = DS_NSLOTS_BND_IN; col < DS_G + DS_L; col++) TR(col, row) = 0;
    }
    {  // BND_OUT: the registers after the last cycle = the cells of the last cycle's rows (written earlier on the stream)
        const size_t row = bnd + DS_ROWOFF_BND_OUT, lc = capacity - 1;
        const size_t rA = (size_t)DS_ROW_A * rs + lc, rB = (size_t)DS_ROW_B * rs + lc, rC = (size_t)DS_ROW_C * rs + lc, rD = (size_t)DS_ROW_D * rs + lc;
        const size_t rPS = (size_t)DS_ROW_PS * rs + lc;
        u64 t12[12], t8[8];
        constexpr int UH[12] = DS_COLS12(D, uh), SH[12] = DS_COLS12(D, sh), RH[12] = DS_COLS12(C, rh), GE[8] = DS_COLS8(D, ge);
#pragma unroll
        for (int k = 0; k < 12; k++) t12[k] = TR(UH[k], rD);
        DS_SET12(cur, uh, t12);
#pragma unroll
        for (int k = 0; k < 12; k++) t12[k] = TR(SH[k], rD);
        DS_SET12(cur, sh, t12);
        u64 rh[12];
#pragma unroll
        for (int k = 0; k < 12; k++) rh[k] = TR(RH[k], rC);
        DS_SET12(cur, rh, rh);
        cur.len_u = TR(DS_D_len_u, rD); cur.len_s = TR(DS_D_len_s, rD); cur.len_r = TR(DS_D_len_r, rD);
        cur.lhs0 = TR(DS_A_lhs0, rA); cur.lhs1 = TR(DS_A_lhs1, rA); cur.rhs0 = TR(DS_A_rhs0, rA); cur.rhs1 = TR(DS_A_rhs1, rA);
        cur.ts = TR(DS_B_ts, rB); cur.page = TR(DS_B_page, rB); cur.h0 = TR(DS_B_h0, rB); cur.h1 = TR(DS_B_h1, rB); cur.h2 = TR(DS_B_h2, rB);
        cur.es3 = TR(DS_PS_es3, rPS); cur.es4 = TR(DS_PS_es4, rPS); cur.es5 = TR(DS_PS_es5, rPS); cur.es6 = TR(DS_PS_es6, rPS);
        cur.es7 = TR(DS_PS_es7, rPS);
        cur.gvalid = TR(DS_C_gvalid, rC);
#pragma unroll
        for (int k = 0; k < 8; k++) t8[k] = TR(GE[k], rD);
        DS_SET8(cur, ge, t8);
        DS_SET12(cur, tail_u, in->initial_queue_state.tail);
        DS_SET12(cur, tail_s, in->sorted_queue_initial_state.tail);
        cur.completion = in->completion_flag ? 1 : 0;
        cur.w_end = gl::canon(cur.len_u) ? gl::inv(cur.len_u) : 0; cur.z_end = cur.len_u == 0;
        cur.flush = cur.completion & cur.gvalid;
        u64 s[12];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = t8[k];
#pragma unroll
        for (int k = 0; k < 4; k++) s[8 + k] = rh[8 + k];
        const size_t rPF = bnd + DS_ROWOFF_PF;
        fill_flattened_poseidon(trace, n_rows, rPF, s);
        for (int col = DS_G; col < DS_G + DS_L; col++) TR(col, rPF) = 0;
        u64 fo[12], fr[12];
#pragma unroll
        for (int k = 0; k < 12; k++) { fo[k] = gl::canon(s[k]); fr[k] = cur.flush ? fo[k] : rh[k]; }
        DS_SET12(cur, fo, fo); DS_SET12(cur, final_rh, fr);
        cur.final_len_r = cur.len_r + cur.flush;
        DS_FILL_BND_OUT(DS_XC, DS_XP, DS_XG, DS_XC)
        for (int col = DS_NSLOTS_BND_OUT; col < DS_G + DS_L; col++) TR(col, row) = 0;
    }
}

// runs after everything else (same stream): BND_IN, BND_OUT, the flush permutation PF (one lane), then the closed-form section
// (closed_form_kernels.cuh) down to the PI row
// (the extra block of k_ds_fill_tail, whose other blocks zero the rows BELOW the boundary rows: the boundary rows' cells are zeroed here first)
__device__ __forceinline__ void ds_boundary_block(const DsSynthJob& job, u32 capacity, size_t n_rows) {
    {
        u64* trace = job.trace;
        const size_t bnd = (size_t)DS_BOUNDARY_ROW(capacity);
        for (int k = threadIdx.x; k < (DS_G + DS_L) * DS_BOUNDARY_ROWS; k += CF_THREADS) TR(k / DS_BOUNDARY_ROWS, bnd + k % DS_BOUNDARY_ROWS) = 0;
    }
    __syncthreads();
    __shared__ u64 sh_oi[50], sh_oo[25], sh_fi[DS_FSM_ENC_LEN], sh_fo[DS_FSM_ENC_LEN], sh_flags[2];
    if (threadIdx.x == 0) ds_fill_register_rows(job, capacity, n_rows);
    if (threadIdx.x == 64) {  // (one lane of each of the other three waves: the encoders run side by side)
        put_queue12(job.first_inst->initial_queue_state, sh_oi);
        put_queue12(job.first_inst->sorted_queue_initial_state, sh_oi + 25);
        put_queue12(job.inst->final_queue_state, sh_oo);
    }
    if (threadIdx.x == 128) ds_encode_fsm(job.inst->hidden_fsm_input, sh_fi);
    if (threadIdx.x == 192) {
        ds_encode_fsm(job.inst->hidden_fsm_output, sh_fo);
        sh_flags[0] = job.inst->start_flag ? 1 : 0;
        sh_flags[1] = job.inst->completion_flag ? 1 : 0;
    }
    __syncthreads();
    const CfSources src = {sh_oi, sh_fi, sh_fo, sh_flags, sh_oo};
    u64* trace = job.trace;
    cf_fill_block(SpecDecommitSorter::cf_spec(), trace, n_rows, (size_t)DS_BOUNDARY_ROW(capacity), src, [&](int rt, size_t row) {
        if (threadIdx.x != 0 || rt != DS_ROW_GIN) return;
        // the encoding of the open group's first request from the FSM words the row copied (decommit query encoding, decommit_kernels.cuh)
        zkw_decommit_query g;
        memset(&g, 0, sizeof g);
        for (int k = 0; k < 8; k++) g.hash[k] = (u32)TR(DS_GIN_gh0 + k, row);
        g.memory_page = (u32)TR(DS_GIN_gpage, row);
        g.timestamp = (u32)TR(DS_GIN_gfts, row);
        g.is_fresh = 1;
        for (int k = 0; k < 4; k++) {
            TR(DS_GIN_gpage_b0 + k, row) = (g.memory_page >> (8 * k)) & 0xFF;  // (counted in the histogram by k_ds_fill_poseidon<1>)
            TR(DS_GIN_gfts_b0 + k, row) = (g.timestamp >> (8 * k)) & 0xFF;
        }
        u64 e[8];
        encode_decommit_query(g, e);
        for (int k = 0; k < 3; k++) TR(DS_GIN_gge0 + k, row) = e[k];
    });
}

// fresh_prefix[k] = fresh requests among sorted[0, k), k = 0..n: the flag of flag_prefix (scan_kernels.cuh)
struct DsFreshFlag {
    const zkw_decommit_query* sorted_q;
    __device__ u32 operator()(size_t i) const { return sorted_q[i].is_fresh ? 1u : 0u; }
};

#undef TR
}  // namespace zkw
