// storage_sorter_circuit_kernels.cuh — synthesis of the StorageSorter trace ("zkw trace v2", circuit type 9,
// include/zkw_storage_sorter_circuit_spec.h) on gfx950.
//
// Counterpart of ZkSyncBaseLayerCircuit::synthesis for that instance type (circuit_definitions/src/circuit_definitions/
// base_layer/mod.rs:286-323, wrapper base_layer/storage_sort_dedup.rs:29-40); the witness it materialises is the output
// of compute_storage_dedup_and_sort (src/witness/individual_circuits/storage_sort_dedup.rs:12-703).
//
// One lane per cycle, region-major rows, no carried state. The cell state machine the circuit walks (base / current
// value, rollback depth, read-at-depth-zero flag) is read back from the builder's global prefix arrays (storage_kernels.cuh:
// depth = D(l) - D(start-1), has = R(l) - R(start-1) > 0, base = read value of the cell's first record, current value = a
// function of record l alone), the key registers are the riders of the previous record's encoding, the result queue
// before a cycle is result_new_tails[E(l-1) - 1]. One lane runs the three dependent permutations of a queue operation.
// Cells of the general rows are scattered through the generated SS_FILL_<row> lists.
#pragma once
#include "events_sorter_circuit_kernels.cuh"
#include "storage_kernels.cuh"
#include "../../include/zkw_storage_sorter_circuit_spec.h"

namespace zkw {

static __constant__ rc_term c_ss_terms[SS_NUM_TERMS] = SS_TERMS_INIT;
static __constant__ rc_constraint c_ss_cons[SS_NUM_CONSTRAINTS] = SS_CONSTRAINTS_INIT;
static __constant__ uint16_t c_ss_row_first[SS_NUM_ROW_TYPES + 1] = SS_ROW_FIRST_CONSTRAINT_INIT;
static __constant__ uint8_t c_ss_is_poseidon[SS_NUM_ROW_TYPES] = SS_ROW_IS_POSEIDON_INIT;
static __constant__ rc_link c_ss_links[SS_NUM_LINKS] = SS_LINKS_INIT;
ZKW_CF_TABLES(SS, ss)
struct SpecStorageSorter {  // StorageSorter, circuit type 9
    static constexpr int G = SS_G, L = SS_L, ROWS_PER_CYCLE = SS_ROWS_PER_CYCLE, NUM_ROW_TYPES = SS_NUM_ROW_TYPES, NUM_LINKS = SS_NUM_LINKS;
    static constexpr int OFF_BIN = SS_ROWOFF_BND_IN, OFF_BOUT = SS_ROWOFF_BND_OUT;
    __device__ static const rc_term* terms() { return c_ss_terms; }
    __device__ static const rc_constraint* cons() { return c_ss_cons; }
    __device__ static const uint16_t* row_first() { return c_ss_row_first; }
    __device__ static const uint8_t* is_poseidon() { return c_ss_is_poseidon; }
    __device__ static const rc_link* links() { return c_ss_links; }
    ZKW_CF_SPEC_MEMBERS(SS, ss)
};

struct SsSynthJob {
    const zkw_storage_sorter_instance* inst;
    const u64 *unsorted_enc, *sorted_enc;              // [n][20] (plain | with extended timestamp)
    const u64 *unsorted_new_tails, *sorted_new_tails;  // [n][4]
    const u64* result_new_tails;                       // [n_result][4]
    const u64* challenges;                             // [2][21]
    const u64 *lhs_z, *rhs_z;                          // [2][n]
    StorageScan sc;                                    // D, S, R, E over the sorted records
    u64 n_block;
    const u64* public_input;  // [4]: commitment of the compact closed-form input (not written: the closed-form section derives the PI row)
    const zkw_storage_sorter_instance* first_inst;  // the block's first instance (the shared observable input)
    u64* trace;
    u32* hist;
    u32 tail_clean;  // the slot already holds this layout (same circuit, capacity, rows): every cell that is zero in EVERY trace of the layout (the padding rows, the gap rows of a region, the columns a row type does not use, multiplicity rows >= 256) is still zero: the fills skip those stores
};

struct SsVars {
#define X(n) u64 n;
    SS_VARS(X)
#undef X
};

#define TR(col, row) trace[(size_t)(col) * n_rows + (row)]
#define SS_I4(M) M(0) M(1) M(2) M(3)
#define SS_I8(M) SS_I4(M) M(4) M(5) M(6) M(7)
#define SS_I16(M) SS_I8(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define SS_I17(M) SS_I16(M) M(16)
#define SS_I18(M) SS_I17(M) M(17)
#define SS_I20(M) SS_I18(M) M(18) M(19)

struct SsCycle {
    bool can_pop;
    size_t idx, pos;  // pos: number of block records popped before this cycle; l = pos - 1 is the latest one
    u64 p_len, cidx;
    u32 pushes;       // records in the result queue before this cycle
};
__device__ __forceinline__ void ss_cycle(const SsSynthJob& job, u32 i, SsCycle& c) {
    const zkw_storage_sorter_instance* in = job.inst;
    const size_t first = in->first_item, m = in->num_items, done = i < m ? i : m;
    c.can_pop = i < m;
    c.idx = first + i;
    c.pos = first + done;
    const u32 len0 = in->start_flag ? in->unsorted_log_queue_state.length : in->hidden_fsm_input.current_unsorted_queue_state.length;
    c.p_len = (u64)len0 - done;
    c.cidx = (u64)in->hidden_fsm_input.cycle_idx + i;
    c.pushes = c.pos >= 2 ? job.sc.E[c.pos - 2] : 0;
}
__device__ __forceinline__ void ss_prev_head(const SsSynthJob& job, const SsCycle& c, int which, u64 h[4]) {
    const zkw_storage_sorter_instance* in = job.inst;
    const u64* tails = which == 0 ? job.unsorted_new_tails : job.sorted_new_tails;
    const u64* src = c.pos ? tails + 4 * (c.pos - 1)
                           : (which == 0 ? in->unsorted_log_queue_state.head : in->intermediate_sorted_queue_state.head);
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = src[k];
}
__device__ __forceinline__ void ss_prev_rh(const SsSynthJob& job, const SsCycle& c, u64 rh[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) rh[k] = c.pushes ? job.result_new_tails[4 * (size_t)(c.pushes - 1) + k] : 0;
}
// key registers = riders of the latest popped record's encoding
struct SsKeys { u64 kc[18], ksh, kts; };
__device__ __forceinline__ void ss_keys_of(const u64* enc /* 20 words */, SsKeys& r) {
#pragma unroll
    for (int k = 0; k < 17; k++) r.kc[k] = enc[k] >> 32;
    const u64 e17 = enc[17];
    r.kc[17] = (e17 >> 32) & 0xFF;
    r.ksh = (e17 >> 48) & 0xFF;
    r.kts = enc[19] >> 8;
}
__device__ __forceinline__ void ss_prev_keys(const SsSynthJob& job, const SsCycle& c, SsKeys& r) {
    if (c.pos) { ss_keys_of(job.sorted_enc + 20 * (c.pos - 1), r); return; }
#pragma unroll
    for (int k = 0; k < 18; k++) r.kc[k] = 0;
    r.ksh = 0; r.kts = 0;
}
// the open cell's registers after the latest popped record
struct SsCell { u64 depth, has, base[8], cur[8]; };
__device__ __forceinline__ void ss_prev_cell(const SsSynthJob& job, const SsCycle& c, SsCell& r) {
    if (!c.pos) {
        r.depth = 0; r.has = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { r.base[k] = 0; r.cur[k] = 0; }
        return;
    }
    const size_t l = c.pos - 1;
    const u32 s = job.sc.S[l];
    r.depth = (u64)(u32)(job.sc.D[l] - (s ? job.sc.D[s - 1] : 0));
    r.has = (job.sc.R[l] - (s ? job.sc.R[s - 1] : 0)) > 0 ? 1 : 0;
    const u64 *el = job.sorted_enc + 20 * l, *eb = job.sorted_enc + 20 * (size_t)s;
    const bool fwd_write = (el[18] & 1) && !(el[19] & 1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        r.base[k] = eb[k] & 0xFFFFFFFFull;
        r.cur[k] = (fwd_write ? el[8 + k] : el[k]) & 0xFFFFFFFFull;
    }
}
// the record the open cell would emit (storage_sort_dedup.rs:394-457) and whether it does
struct SsEmit { u64 z_d, eqv, q1, em, eq[8], wsel[8], w[20]; };
__device__ __forceinline__ void ss_emit(const SsCell& cell, const SsKeys& keys, SsEmit& e) {
    e.q1 = 1; e.eqv = 1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        e.eq[k] = cell.cur[k] == cell.base[k] ? 1 : 0;
        if (k < 4) e.q1 &= e.eq[k];
        e.eqv &= e.eq[k];
    }
    e.z_d = cell.depth == 0 ? 1 : 0;
    e.em = e.z_d ? cell.has : 1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        e.wsel[k] = e.z_d ? cell.base[k] : cell.cur[k];
        e.w[k] = cell.base[k] + (keys.kc[k] << 32);
        e.w[8 + k] = e.wsel[k] + (keys.kc[8 + k] << 32);
    }
    e.w[16] = keys.kc[16] << 32;
    e.w[17] = (keys.kc[17] << 32) + (keys.ksh << 48);
    e.w[18] = (!e.z_d && !e.eqv) ? 1 : 0;
    e.w[19] = 0;
}
// same cell as the previous record: all 18 riders equal
__device__ __forceinline__ bool ss_same_key(const u64 cv[18], const SsKeys& p) {
    bool eq = true;
#pragma unroll
    for (int k = 0; k < 18; k++) eq &= cv[k] == p.kc[k];
    return eq;
}

// WHICH 0 = unsorted pop (U1..U3), 1 = sorted pop (S1..S3), 2 = result push (R1..R3)
template <int WHICH>
static __device__ __forceinline__ void k_ss_fill_queue(const VB& vb, const SsSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    const SsSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = SS_REGION_STRIDE(capacity);
    constexpr int R1 = WHICH == 0 ? SS_ROW_U1 : (WHICH == 1 ? SS_ROW_S1 : SS_ROW_R1);
    u64* trace = job.trace;
    if (i < capacity) {
        SsCycle c;
        ss_cycle(job, i, c);
        u64 enc[20], old[4], out4[4];
        if (WHICH == 2) {
            SsKeys keys;
            SsCell cell;
            SsEmit em;
            ss_prev_keys(job, c, keys);
            ss_prev_cell(job, c, cell);
            ss_emit(cell, keys, em);
#pragma unroll
            for (int k = 0; k < 20; k++) enc[k] = em.w[k];
            ss_prev_rh(job, c, old);
        } else {
            const u64* src = WHICH == 0 ? job.unsorted_enc : job.sorted_enc;
#pragma unroll
            for (int k = 0; k < 20; k++) enc[k] = c.can_pop ? src[20 * c.idx + k] : 0;
            ss_prev_head(job, c, WHICH, old);
        }
        es_queue_op(trace, n_rows, (size_t)R1 * rs + i, (size_t)(R1 + 1) * rs + i, (size_t)(R1 + 2) * rs + i, enc, old, out4);
        if (!job.tail_clean) for (int r = 0; r < 3; r++)
            for (int col = 130; col < SS_G + SS_L; col++) TR(col, (size_t)(R1 + r) * rs + i) = 0;
    } else if (i < rs) {
        if (!job.tail_clean) for (int r = 0; r < 3; r++) zero_gap_row_n(trace, n_rows, (size_t)(R1 + r) * rs + i, SS_G + SS_L);
    }
}

#define SS_XC(col, v) TR(col, row) = cur.v;
#define SS_XP(col, v) TR(col, row) = prev.v;
#define SS_XG(col, v) TR(col, row) = glob.v;
#define SS_COLS4(ROW, v) {SS_##ROW##_##v##0, SS_##ROW##_##v##1, SS_##ROW##_##v##2, SS_##ROW##_##v##3}
#define SS_SPLIT(k) cur.lo##k = lo[k]; cur.lo##k##_b0 = lo[k] & 0xFF; cur.lo##k##_b1 = (lo[k] >> 8) & 0xFF; cur.lo##k##_b2 = (lo[k] >> 16) & 0xFF; \
    cur.lo##k##_b3 = lo[k] >> 24; cur.c##k = cv[k]; cur.c##k##_b0 = cv[k] & 0xFF; cur.c##k##_b1 = (cv[k] >> 8) & 0xFF; cur.c##k##_b2 = cv[k] >> 16;
#define SS_EMIT_TO_VARS(v, e, W) do { \
    v.q1 = (e).q1; v.eqv = (e).eqv; v.z_d = (e).z_d; v.em = (e).em; \
    v.eq0 = (e).eq[0]; v.eq1 = (e).eq[1]; v.eq2 = (e).eq[2]; v.eq3 = (e).eq[3]; v.eq4 = (e).eq[4]; v.eq5 = (e).eq[5]; v.eq6 = (e).eq[6]; v.eq7 = (e).eq[7]; \
    v.wsel0 = (e).wsel[0]; v.wsel1 = (e).wsel[1]; v.wsel2 = (e).wsel[2]; v.wsel3 = (e).wsel[3]; v.wsel4 = (e).wsel[4]; v.wsel5 = (e).wsel[5]; \
    v.wsel6 = (e).wsel[6]; v.wsel7 = (e).wsel[7]; \
    v.W##0 = (e).w[0]; v.W##1 = (e).w[1]; v.W##2 = (e).w[2]; v.W##3 = (e).w[3]; v.W##4 = (e).w[4]; v.W##5 = (e).w[5]; v.W##6 = (e).w[6]; \
    v.W##7 = (e).w[7]; v.W##8 = (e).w[8]; v.W##9 = (e).w[9]; v.W##10 = (e).w[10]; v.W##11 = (e).w[11]; v.W##12 = (e).w[12]; v.W##13 = (e).w[13]; \
    v.W##14 = (e).w[14]; v.W##15 = (e).w[15]; v.W##16 = (e).w[16]; v.W##17 = (e).w[17]; v.W##18 = (e).w[18]; v.W##19 = (e).w[19]; } while (0)
#define SS_CELL_TO(v, cell) do { v.depth = (cell).depth; v.has = (cell).has; \
    v.base0 = (cell).base[0]; v.base1 = (cell).base[1]; v.base2 = (cell).base[2]; v.base3 = (cell).base[3]; v.base4 = (cell).base[4]; \
    v.base5 = (cell).base[5]; v.base6 = (cell).base[6]; v.base7 = (cell).base[7]; \
    v.cur0 = (cell).cur[0]; v.cur1 = (cell).cur[1]; v.cur2 = (cell).cur[2]; v.cur3 = (cell).cur[3]; v.cur4 = (cell).cur[4]; \
    v.cur5 = (cell).cur[5]; v.cur6 = (cell).cur[6]; v.cur7 = (cell).cur[7]; } while (0)
#define SS_KEYS_TO(v, keys) do { v.ksh = (keys).ksh; v.kts = (keys).kts; \
    v.kc0 = (keys).kc[0]; v.kc1 = (keys).kc[1]; v.kc2 = (keys).kc[2]; v.kc3 = (keys).kc[3]; v.kc4 = (keys).kc[4]; v.kc5 = (keys).kc[5]; \
    v.kc6 = (keys).kc[6]; v.kc7 = (keys).kc[7]; v.kc8 = (keys).kc[8]; v.kc9 = (keys).kc[9]; v.kc10 = (keys).kc[10]; v.kc11 = (keys).kc[11]; \
    v.kc12 = (keys).kc[12]; v.kc13 = (keys).kc[13]; v.kc14 = (keys).kc[14]; v.kc15 = (keys).kc[15]; v.kc16 = (keys).kc[16]; v.kc17 = (keys).kc[17]; } while (0)
// the eight inverse witnesses of the cur == base gadget and the one of depth == 0: one field inversion (Montgomery's trick)
#define SS_EMIT_INVERSES(v, cell) do { \
    u64 _d[9], _pre[9], _acc = 1; \
    for (int k = 0; k < 8; k++) _d[k] = gl::canon(gl::sub((cell).cur[k], (cell).base[k])); \
    _d[8] = (cell).depth; \
    for (int k = 0; k < 9; k++) { _pre[k] = _acc; if (_d[k]) _acc = gl::mul(_acc, _d[k]); } \
    u64 _ia = gl::inv(_acc), _w[9]; \
    for (int k = 8; k >= 0; k--) { _w[k] = _d[k] ? gl::canon(gl::mul(_ia, _pre[k])) : 0; if (_d[k]) _ia = gl::mul(_ia, _d[k]); } \
    v.wq0 = _w[0]; v.wq1 = _w[1]; v.wq2 = _w[2]; v.wq3 = _w[3]; v.wq4 = _w[4]; v.wq5 = _w[5]; v.wq6 = _w[6]; v.wq7 = _w[7]; v.w_d = _w[8]; } while (0)

template <int ROW>
static __device__ __forceinline__ void k_ss_fill_row(const VB& vb, const SsSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const SsSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = SS_REGION_STRIDE(capacity);
    u64* trace = job.trace;
    if (i < capacity) {
        const size_t row = (size_t)ROW * rs + i, n = job.n_block;
        SsCycle c;
        ss_cycle(job, i, c);
        SsVars cur, prev, glob;
        const u64 can_pop = c.can_pop ? 1 : 0;
        const u64 p_valid = c.pos ? 1 : 0;
        cur.can_pop = can_pop;
        prev.valid = p_valid;
        u64 es[20];
#pragma unroll
        for (int k = 0; k < 20; k++) es[k] = c.can_pop ? job.sorted_enc[20 * c.idx + k] : 0;
#define M(k) cur.es##k = es[k];
        SS_I20(M)
#undef M
        u64 lo[17], cv[18];
#pragma unroll
        for (int k = 0; k < 17; k++) { lo[k] = es[k] & 0xFFFFFFFFull; cv[k] = es[k] >> 32; }
        cv[17] = (es[17] >> 32) & 0xFF;
        const u64 shard = (es[17] >> 48) & 0xFF, rw = es[18] & 1, rb = es[19] & 1, ts = es[19] >> 8;
        cur.rw = rw; cur.rb = rb; cur.ts = ts; cur.shard = shard;
        if (ROW == SS_ROW_A) {
            u64 eu[20];
#pragma unroll
            for (int k = 0; k < 20; k++) eu[k] = c.can_pop ? job.unsorted_enc[20 * c.idx + k] : 0;
#define M(k) cur.eu##k = eu[k];
            SS_I20(M)
#undef M
            prev.cidx = c.cidx;
            u64* g = &glob.c0_1;  // c0_1..c0_20, c1_1..c1_20 are consecutive fields
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const u64* ch = job.challenges + 21 * r;
#pragma unroll
                for (int k = 1; k <= 20; k++) g[20 * r + k - 1] = ch[k];
                u64 lc = gl::add(ch[20], eu[0]), rc = gl::add(ch[20], es[0]);
#pragma unroll
                for (int k = 1; k < 20; k++) { lc = gl::add(lc, gl::mul(eu[k], ch[k])); rc = gl::add(rc, gl::mul(es[k], ch[k])); }
                if (c.can_pop) lc = gl::add(lc, gl::mul(256 * c.cidx, ch[19]));  // extended timestamp = queue position
                const u64 pl = c.pos ? job.lhs_z[(size_t)r * n + c.pos - 1] : (u64)1;  // pos == 0 only in the first instance: ONE at the start
                const u64 pr = c.pos ? job.rhs_z[(size_t)r * n + c.pos - 1] : (u64)1;
                const u64 nl = gl::canon(gl::mul(pl, lc)), nr = gl::canon(gl::mul(pr, rc));
                lc = gl::canon(lc); rc = gl::canon(rc);
                if (r == 0) { cur.lc0 = lc; cur.rc0 = rc; cur.nl0 = nl; cur.nr0 = nr; prev.lhs0 = pl; prev.rhs0 = pr; cur.lhs0 = can_pop ? nl : pl; cur.rhs0 = can_pop ? nr : pr; }
                else { cur.lc1 = lc; cur.rc1 = rc; cur.nl1 = nl; cur.nr1 = nr; prev.lhs1 = pl; prev.rhs1 = pr; cur.lhs1 = can_pop ? nl : pl; cur.rhs1 = can_pop ? nr : pr; }
            }
        }
        if (ROW >= SS_ROW_X0 && ROW <= SS_ROW_K) { SS_I17(SS_SPLIT) }
        if (ROW == SS_ROW_K || ROW == SS_ROW_C1 || ROW == SS_ROW_C2 || ROW == SS_ROW_Q) {
            SsKeys pk;
            ss_prev_keys(job, c, pk);
            SS_KEYS_TO(prev, pk);
            cur.c17 = cv[17];
            const u64 keq = ss_same_key(cv, pk) ? 1 : 0;
            cur.keq = keq;
            if (ROW == SS_ROW_K) {
                cur.ts_b0 = ts & 0xFF; cur.ts_b1 = (ts >> 8) & 0xFF; cur.ts_b2 = (ts >> 16) & 0xFF; cur.ts_b3 = ts >> 24;
                // 18 equality gadgets with one inversion; the first difference from the top
                u64 d[18], pre[18], acc = 1, w[18], ek[18], pe[17];
#pragma unroll
                for (int k = 0; k < 18; k++) { d[k] = gl::canon(gl::sub(cv[k], pk.kc[k])); ek[k] = d[k] == 0; pre[k] = acc; if (d[k]) acc = gl::mul(acc, d[k]); }
                u64 ia = gl::inv(acc);
#pragma unroll
                for (int k = 17; k >= 0; k--) { w[k] = d[k] ? gl::canon(gl::mul(ia, pre[k])) : 0; if (d[k]) ia = gl::mul(ia, d[k]); }
                pe[16] = ek[17];
#pragma unroll
                for (int k = 15; k >= 0; k--) pe[k] = pe[k + 1] & ek[k + 1];
                u64 diff = d[17];
#pragma unroll
                for (int k = 16; k >= 0; k--)
                    if (pe[k]) diff = gl::add(diff, d[k]);
                if (keq) diff = gl::add(diff, gl::sub(ts, pk.kts));
                diff = gl::canon(diff);
#define M(k) cur.wk##k = w[k]; cur.ek##k = ek[k];
                SS_I18(M)
#undef M
#define M(k) cur.pe##k = pe[k];
                SS_I16(M)
#undef M
                cur.diff = diff;
                const u64 dm = (can_pop & p_valid) ? diff - 1 : 0;  // the builder sorted the records: 1 <= diff < 2^32 + 1
                cur.d_b0 = dm & 0xFF; cur.d_b1 = (dm >> 8) & 0xFF; cur.d_b2 = (dm >> 16) & 0xFF; cur.d_b3 = (dm >> 24) & 0xFF;
            }
            if (ROW == SS_ROW_C1 || ROW == SS_ROW_C2 || ROW == SS_ROW_Q) {
                SsCell cell;
                SsEmit em;
                ss_prev_cell(job, c, cell);
                ss_emit(cell, pk, em);
                SS_CELL_TO(prev, cell);
                SS_EMIT_TO_VARS(cur, em, pw);
                const u64 nkey = can_pop & p_valid & (1 - keq);
                const u64 push = nkey & em.em;
                cur.nkey = nkey; cur.push = push;
                if (ROW == SS_ROW_C1) {
                    const u64 tx = es[17] & 0xFFFFFFFFull;
                    cur.tx_b0 = tx & 0xFF; cur.tx_b1 = (tx >> 8) & 0xFF; cur.tx_b2 = (tx >> 16) & 0xFF; cur.tx_b3 = tx >> 24;
                    cur.aux = (es[17] >> 40) & 0xFF; cur.sv = es[18] >> 1;
                    SS_EMIT_INVERSES(cur, cell);
                }
                if (ROW == SS_ROW_C2) {
                    const u64 same = p_valid & keq, sm = can_pop & same, nc = can_pop - sm, wr = rw & (1 - rb), rbk = rw & rb;
                    cur.same = same; cur.sm = sm; cur.nc = nc; cur.wr = wr; cur.rbk = rbk;
                    u64 t[8], ncur[8], nbase[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        t[k] = wr ? lo[8 + k] : lo[k];
                        ncur[k] = c.can_pop ? t[k] : cell.cur[k];
                        nbase[k] = nc ? lo[k] : cell.base[k];
                    }
#define M(k) cur.lo##k = lo[k];
                    SS_I16(M)
#undef M
                    cur.t0 = t[0]; cur.t1 = t[1]; cur.t2 = t[2]; cur.t3 = t[3]; cur.t4 = t[4]; cur.t5 = t[5]; cur.t6 = t[6]; cur.t7 = t[7];
                    cur.cur0 = ncur[0]; cur.cur1 = ncur[1]; cur.cur2 = ncur[2]; cur.cur3 = ncur[3]; cur.cur4 = ncur[4]; cur.cur5 = ncur[5]; cur.cur6 = ncur[6]; cur.cur7 = ncur[7];
                    cur.base0 = nbase[0]; cur.base1 = nbase[1]; cur.base2 = nbase[2]; cur.base3 = nbase[3]; cur.base4 = nbase[4]; cur.base5 = nbase[5];
                    cur.base6 = nbase[6]; cur.base7 = nbase[7];
                    cur.depth = nc ? rw : cell.depth + (sm ? wr : 0) - (sm ? rbk : 0);
                    cur.u = sm & (1 - rw) & em.z_d;
                    cur.has = nc ? 1 - rw : (cell.has | cur.u);
                    cur.valid = p_valid | can_pop;
                }
                if (ROW == SS_ROW_Q) {
                    prev.len_u = c.p_len; prev.len_s = c.p_len;
                    cur.w_lu = c.p_len ? gl::inv(c.p_len) : 0; cur.z_lu = c.p_len == 0; cur.w_ls = cur.w_lu; cur.z_ls = cur.z_lu;
                    cur.len_u = c.p_len - can_pop; cur.len_s = c.p_len - can_pop;
                    constexpr int U3O[4] = SS_COLS4(U3, u3o), S3O[4] = SS_COLS4(S3, s3o), R3O[4] = SS_COLS4(R3, r3o);
                    const size_t rU3 = (size_t)SS_ROW_U3 * rs + i, rS3 = (size_t)SS_ROW_S3 * rs + i, rR3 = (size_t)SS_ROW_R3 * rs + i;
                    u64 uo[4], so[4], ro[4], pu[4], ps[4], prh[4], ou[4], os[4], orh[4];
                    ss_prev_head(job, c, 0, pu);
                    ss_prev_head(job, c, 1, ps);
                    ss_prev_rh(job, c, prh);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uo[k] = TR(U3O[k], rU3); so[k] = TR(S3O[k], rS3); ro[k] = TR(R3O[k], rR3);
                        ou[k] = c.can_pop ? uo[k] : pu[k]; os[k] = c.can_pop ? so[k] : ps[k]; orh[k] = push ? ro[k] : prh[k];
                    }
#define M(k) cur.u3o##k = uo[k]; cur.s3o##k = so[k]; cur.r3o##k = ro[k]; prev.uh##k = pu[k]; prev.sh##k = ps[k]; prev.rh##k = prh[k]; \
    cur.uh##k = ou[k]; cur.sh##k = os[k]; cur.rh##k = orh[k];
                    SS_I4(M)
#undef M
                    prev.len_r = c.pushes; cur.len_r = c.pushes + push;
#define M(k) cur.c##k = cv[k]; cur.kc##k = c.can_pop ? cv[k] : pk.kc[k];
                    SS_I18(M)
#undef M
                    cur.ksh = c.can_pop ? shard : pk.ksh;
                    cur.kts = c.can_pop ? ts : pk.kts;
                    prev.cidx = c.cidx; cur.cidx = c.cidx + 1;
                }
            }
        }
#define SS_ROWCASE(R) if (ROW == SS_ROW_##R) { SS_FILL_##R(SS_XC, SS_XP, SS_XG, SS_XC) }
        SS_ROWCASE(A) SS_ROWCASE(X0) SS_ROWCASE(X1) SS_ROWCASE(X2) SS_ROWCASE(X3) SS_ROWCASE(X4) SS_ROWCASE(X5) SS_ROWCASE(X6) SS_ROWCASE(X7)
        SS_ROWCASE(K) SS_ROWCASE(C1) SS_ROWCASE(C2) SS_ROWCASE(Q)
#undef SS_ROWCASE
        constexpr int NSL[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, SS_NSLOTS_A, SS_NSLOTS_X0, SS_NSLOTS_X1, SS_NSLOTS_X2, SS_NSLOTS_X3, SS_NSLOTS_X4,
                               SS_NSLOTS_X5, SS_NSLOTS_X6, SS_NSLOTS_X7, SS_NSLOTS_K, SS_NSLOTS_C1, SS_NSLOTS_C2, SS_NSLOTS_Q};
        constexpr int NLK[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, SS_NLOOK_A, SS_NLOOK_X0, SS_NLOOK_X1, SS_NLOOK_X2, SS_NLOOK_X3, SS_NLOOK_X4,
                               SS_NLOOK_X5, SS_NLOOK_X6, SS_NLOOK_X7, SS_NLOOK_K, SS_NLOOK_C1, SS_NLOOK_C2, SS_NLOOK_Q};
        if (!job.tail_clean) for (int col = NSL[ROW]; col < SS_G; col++) TR(col, row) = 0;
        if (!job.tail_clean) for (int col = SS_G + NLK[ROW]; col < SS_G + SS_L; col++) TR(col, row) = 0;
        for (int col = SS_G; col < SS_G + NLK[ROW]; col++) atomicAdd(&sh_hist[(u32)TR(col, row) & 0xFF], 1u);
    } else if (i < rs) {
        if (!job.tail_clean) zero_gap_row_n(trace, n_rows, (size_t)ROW * rs + i, SS_G + SS_L);
    }
    if (ROW == SS_ROW_A && vb.x == 0 && threadIdx.x == 0)  // the closed-form section's lookup cells: the bytes of the FSM records' previous_packed_key (bridge rows KIB* / KOB*)
        for (int k = 0; k < ZKW_STORAGE_PACKED_KEY_LENGTH; k++) {
            hist_bytes(sh_hist, job.inst->hidden_fsm_input.previous_packed_key[k]);
            hist_bytes(sh_hist, job.inst->hidden_fsm_output.previous_packed_key[k]);
        }
    hist_flush(sh_hist, job.hist);
}

constexpr int SS_BOUNDARY_ROWS = (SS_NUM_ROW_TYPES - SS_ROWS_PER_CYCLE + 1) & ~1;  // register rows, PI, flush rows, the closed-form section (rounded up to even: 16-byte stores below)
__device__ __forceinline__ void ss_boundary_block(const SsSynthJob& job, u32 capacity, size_t n_rows);
static __device__ __forceinline__ void k_ss_fill_tail(const VB& vb, const SsSynthJob* __restrict__ jobs, u32 n_jobs, u32 capacity, size_t n_rows) {
    // 1-D grid: the first n_jobs blocks fill the boundary rows of one trace each (dispatched first and at raised priority: a chain of a dozen
    // dependent permutations that the other blocks' stores hide), then (SS_G + SS_L + 1) * TAIL_CHUNKS blocks per trace
    if (vb.x < n_jobs) {
        __builtin_amdgcn_s_setprio(3);
        ss_boundary_block(jobs[vb.x], capacity, n_rows);
        return;
    }
    constexpr u32 PER_JOB = (SS_G + SS_L + 1) * TAIL_CHUNKS;
    const u32 bid = (vb.x - n_jobs) % PER_JOB;
    const SsSynthJob& job = jobs[(vb.x - n_jobs) / PER_JOB];
    u64* trace = job.trace;
    const int col = bid / TAIL_CHUNKS, ch = bid % TAIL_CHUNKS;
    if (col < SS_G + SS_L) {
        if (job.tail_clean) return;
        const size_t bnd = (size_t)SS_BOUNDARY_ROW(capacity) + SS_BOUNDARY_ROWS;
        const size_t n_pairs = (n_rows - bnd) / 2;
        const size_t per = (n_pairs + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
        ulonglong2* c2 = reinterpret_cast<ulonglong2*>(trace + (size_t)col * n_rows + bnd);
        const ulonglong2 z = make_ulonglong2(0, 0);
        for (size_t k = lo + threadIdx.x; k < hi; k += 256) c2[k] = z;
        return;
    }
    u64* mlt = trace + (size_t)SS_MULT_COL * n_rows;
    const size_t per = (n_rows + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_rows ? lo + per : n_rows;
    for (size_t r = lo + threadIdx.x; r < (job.tail_clean && hi > 256 ? (lo < 256 ? 256 : lo) : hi); r += 256) {  // (a clean slot: rows >= 256 of the column are still zero)
        u64 v = 0;
        if (r < 256) {
            v = job.hist[r];
            if (r == 0) v += (u64)SS_L * n_rows - (u64)SS_LOOKUPS_PER_CYCLE * capacity - 4 * SS_CF_NUM_BYTES;  // (the section's byte cells: counted in job.hist by k_ss_fill_row<A>)
        }
        mlt[r] = v;
    }
}

// BND_IN, BND_OUT, the flush permutations F1..F3, PI (runs last on the stream: reads the last cycle's rows)
__device__ __forceinline__ void ss_fill_register_rows(const SsSynthJob& job, u32 capacity, size_t n_rows) {
    u64* trace = job.trace;
    const zkw_storage_sorter_instance* in = job.inst;
    const size_t rs = SS_REGION_STRIDE(capacity), bnd = (size_t)SS_BOUNDARY_ROW(capacity);
    SsVars cur, glob;
    u64* g = &glob.c0_1;
    for (int r = 0; r < 2; r++)
        for (int k = 1; k <= 20; k++) g[20 * r + k - 1] = job.challenges[21 * r + k];
#define SS_XPB(col, v)
    {
        const size_t row = bnd + SS_ROWOFF_BND_IN, n = job.n_block;
        SsCycle c;
        ss_cycle(job, 0, c);
        SsKeys pk;
        SsCell cell;
        ss_prev_keys(job, c, pk);
        ss_prev_cell(job, c, cell);
        u64 pu[4], ps[4], prh[4];
        ss_prev_head(job, c, 0, pu);
        ss_prev_head(job, c, 1, ps);
        ss_prev_rh(job, c, prh);
#define M(k) cur.uh##k = pu[k]; cur.sh##k = ps[k]; cur.rh##k = prh[k];
        SS_I4(M)
#undef M
        cur.len_u = c.p_len; cur.len_s = c.p_len; cur.len_r = c.pushes;
        cur.lhs0 = c.pos ? job.lhs_z[c.pos - 1] : (u64)1;
        cur.lhs1 = c.pos ? job.lhs_z[n + c.pos - 1] : (u64)1;
        cur.rhs0 = c.pos ? job.rhs_z[c.pos - 1] : (u64)1;
        cur.rhs1 = c.pos ? job.rhs_z[n + c.pos - 1] : (u64)1;
        SS_KEYS_TO(cur, pk);
        SS_CELL_TO(cur, cell);
        cur.cidx = c.cidx; cur.valid = c.pos ? 1 : 0;
        SS_FILL_BND_IN(SS_XC, SS_XPB, SS_XG, SS_XC)
        for (int col = SS_NSLOTS_BND_IN; col < SS_G + SS_L; col++) TR(col, row) = 0;
    }
    {
        const size_t row = bnd + SS_ROWOFF_BND_OUT, lc = capacity - 1;
        const size_t rA = (size_t)SS_ROW_A * rs + lc, rC2 = (size_t)SS_ROW_C2 * rs + lc, rQ = (size_t)SS_ROW_Q * rs + lc;
        SsKeys keys;
        SsCell cell;
        u64 rh[4];
#define M(k) cur.uh##k = TR(SS_Q_uh##k, rQ); cur.sh##k = TR(SS_Q_sh##k, rQ); rh[k] = TR(SS_Q_rh##k, rQ); cur.rh##k = rh[k];
        SS_I4(M)
#undef M
        cur.len_u = TR(SS_Q_len_u, rQ); cur.len_s = TR(SS_Q_len_s, rQ); cur.len_r = TR(SS_Q_len_r, rQ);
        cur.lhs0 = TR(SS_A_lhs0, rA); cur.lhs1 = TR(SS_A_lhs1, rA); cur.rhs0 = TR(SS_A_rhs0, rA); cur.rhs1 = TR(SS_A_rhs1, rA);
#define M(k) keys.kc[k] = TR(SS_Q_kc##k, rQ);
        SS_I18(M)
#undef M
        keys.ksh = TR(SS_Q_ksh, rQ); keys.kts = TR(SS_Q_kts, rQ);
        SS_KEYS_TO(cur, keys);
        cur.cidx = TR(SS_Q_cidx, rQ);
        cur.valid = TR(SS_C2_valid, rC2);
        cell.depth = TR(SS_C2_depth, rC2); cell.has = TR(SS_C2_has, rC2);
#define M(k) cell.base[k] = TR(SS_C2_base##k, rC2); cell.cur[k] = TR(SS_C2_cur##k, rC2);
        SS_I8(M)
#undef M
        SS_CELL_TO(cur, cell);
#define M(k) cur.tail_u##k = in->unsorted_log_queue_state.tail[k]; cur.tail_s##k = in->intermediate_sorted_queue_state.tail[k];
        SS_I4(M)
#undef M
        cur.completion = in->completion_flag ? 1 : 0;
        cur.w_end = gl::canon(cur.len_u) ? gl::inv(cur.len_u) : 0; cur.z_end = cur.len_u == 0;
        SsEmit em;
        ss_emit(cell, keys, em);
        SS_EMIT_TO_VARS(cur, em, fw);
        SS_EMIT_INVERSES(cur, cell);
        cur.flush = cur.completion & cur.valid & em.em;
        u64 o4[4];
        es_queue_op(trace, n_rows, bnd + SS_ROWOFF_F1, bnd + SS_ROWOFF_F2, bnd + SS_ROWOFF_F3, em.w, rh, o4);
        for (int r = 0; r < 3; r++)
            for (int col = 130; col < SS_G + SS_L; col++) TR(col, bnd + SS_ROWOFF_F1 + r) = 0;
#define M(k) cur.f3o##k = o4[k]; cur.final_rh##k = cur.flush ? o4[k] : rh[k];
        SS_I4(M)
#undef M
        cur.final_len_r = cur.len_r + cur.flush;
        SS_FILL_BND_OUT(SS_XC, SS_XPB, SS_XG, SS_XC)
        for (int col = SS_NSLOTS_BND_OUT; col < SS_G + SS_L; col++) TR(col, row) = 0;
    }
}

// the register rows (one lane), then the closed-form section down to the PI row (runs last on the stream: reads the last cycle's rows)
// (the extra block of k_ss_fill_tail, whose other blocks zero the rows BELOW the boundary rows: the boundary rows' cells are zeroed here first)
__device__ __forceinline__ void ss_boundary_block(const SsSynthJob& job, u32 capacity, size_t n_rows) {
    {
        u64* trace = job.trace;
        const size_t bnd = (size_t)SS_BOUNDARY_ROW(capacity);
        for (int k = threadIdx.x; k < (SS_G + SS_L) * SS_BOUNDARY_ROWS; k += CF_THREADS) TR(k / SS_BOUNDARY_ROWS, bnd + k % SS_BOUNDARY_ROWS) = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) ss_fill_register_rows(job, capacity, n_rows);
    cf_section_from_records<CfStorageSorter, SpecStorageSorter>(job.first_inst, job.inst, job.trace, n_rows, (size_t)SS_BOUNDARY_ROW(capacity), [](int, size_t) {});
}

#undef TR
}  // namespace zkw
