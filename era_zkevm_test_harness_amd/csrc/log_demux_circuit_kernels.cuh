// log_demux_circuit_kernels.cuh — synthesis of the LogDemuxer trace ("zkw trace v2", circuit type 4,
// include/zkw_log_demux_circuit_spec.h) on gfx950.
//
// Counterpart of ZkSyncBaseLayerCircuit::synthesis for that instance type (circuit_definitions/src/circuit_definitions/
// base_layer/mod.rs:286-323, wrapper base_layer/log_demux.rs:27-38); the witness it materialises is the output of
// compute_logs_demux (src/witness/individual_circuits/log_demux.rs:20-388).
//
// One lane per cycle, region-major rows, no carried state: the tail of output queue c before cycle i is
// out_new_tails[offset_c + count_c - 1] with count_c the builder's inclusive route prefix count at the previous item,
// the input queue's head is in_new_tails of the previous item. The pop and the one conditional push are three dependent
// permutations each (lib.rs:179-221): one lane runs all three and writes the three Poseidon2 rows. The route flags of
// row R are re-derived from the bytes of the record's encoding. Cells of the general rows are scattered through the
// generated LD_FILL_<row> lists.
#pragma once
#include "events_sorter_circuit_kernels.cuh"
#include "../../include/zkw_log_demux_circuit_spec.h"

namespace zkw {

static __constant__ rc_term c_ld_terms[LD_NUM_TERMS] = LD_TERMS_INIT;
static __constant__ rc_constraint c_ld_cons[LD_NUM_CONSTRAINTS] = LD_CONSTRAINTS_INIT;
static __constant__ uint16_t c_ld_row_first[LD_NUM_ROW_TYPES + 1] = LD_ROW_FIRST_CONSTRAINT_INIT;
static __constant__ uint8_t c_ld_is_poseidon[LD_NUM_ROW_TYPES] = LD_ROW_IS_POSEIDON_INIT;
static __constant__ rc_link c_ld_links[LD_NUM_LINKS] = LD_LINKS_INIT;
ZKW_CF_TABLES(LD, ld)
struct SpecLogDemux {  // LogDemuxer, circuit type 4
    static constexpr int G = LD_G, L = LD_L, ROWS_PER_CYCLE = LD_ROWS_PER_CYCLE, NUM_ROW_TYPES = LD_NUM_ROW_TYPES, NUM_LINKS = LD_NUM_LINKS;
    static constexpr int OFF_BIN = LD_ROWOFF_BND_IN, OFF_BOUT = LD_ROWOFF_BND_OUT;
    __device__ static const rc_term* terms() { return c_ld_terms; }
    __device__ static const rc_constraint* cons() { return c_ld_cons; }
    __device__ static const uint16_t* row_first() { return c_ld_row_first; }
    __device__ static const uint8_t* is_poseidon() { return c_ld_is_poseidon; }
    __device__ static const rc_link* links() { return c_ld_links; }
    ZKW_CF_SPEC_MEMBERS(LD, ld)
};

struct LdSynthJob {
    const zkw_log_demux_instance* inst;
    const u64* in_enc;         // [n][20]
    const u64* in_new_tails;   // [n][4]
    const u64* out_new_tails;  // [routed][4], the six queues back to back
    const u32* route_count;    // [6][n] inclusive prefix counts per route
    u64 offsets[7];
    u64 n_block;
    const u64* public_input;  // [4]: commitment of the compact closed-form input (not written: the closed-form section derives the PI row)
    const zkw_log_demux_instance* first_inst;  // the block's first instance (the shared observable input)
    u64* trace;
    u32* hist;
    u32 tail_clean;  // the slot already holds this layout (same circuit, capacity, rows): every cell that is zero in EVERY trace of the layout (the padding rows, the gap rows of a region, the columns a row type does not use, multiplicity rows >= 256) is still zero: the fills skip those stores
};

struct LdVars {
#define X(n) u64 n;
    LD_VARS(X)
#undef X
};

#define TR(col, row) trace[(size_t)(col) * n_rows + (row)]

struct LdCycle {
    bool can_pop;
    size_t idx, pos;  // pos: number of block items popped before this cycle
    u32 p_len;
    u32 cnt[6];       // items routed into queue c before this cycle
};
__device__ __forceinline__ void ld_cycle(const LdSynthJob& job, u32 i, LdCycle& c) {
    const zkw_log_demux_instance* in = job.inst;
    const size_t first = in->first_item, m = in->num_items, done = i < m ? i : m;
    c.can_pop = i < m;
    c.idx = first + i;
    c.pos = first + done;
    const u32 len0 = in->start_flag ? in->initial_log_queue_state.length : in->hidden_fsm_input.initial_log_queue_state.length;
    c.p_len = len0 - (u32)done;
#pragma unroll
    for (int q = 0; q < 6; q++) c.cnt[q] = c.pos ? job.route_count[(size_t)q * job.n_block + c.pos - 1] : 0;
}
__device__ __forceinline__ void ld_prev_head(const LdSynthJob& job, const LdCycle& c, u64 h[4]) {
    const zkw_log_demux_instance* in = job.inst;
    const u64* src = c.pos ? job.in_new_tails + 4 * (c.pos - 1)
                           : (in->start_flag ? in->initial_log_queue_state.head : in->hidden_fsm_input.initial_log_queue_state.head);
#pragma unroll
    for (int k = 0; k < 4; k++) h[k] = src[k];
}
__device__ __forceinline__ void ld_queue_tail(const LdSynthJob& job, const LdCycle& c, int q, u64 t[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = c.cnt[q] ? job.out_new_tails[4 * (job.offsets[q] + c.cnt[q] - 1) + k] : 0;
}
// route of this cycle's record as the builder counted it: -1 = not pushed
__device__ __forceinline__ int ld_route(const LdSynthJob& job, const LdCycle& c) {
    int r = -1;
    if (c.can_pop) {
#pragma unroll
        for (int q = 0; q < 6; q++)
            if (job.route_count[(size_t)q * job.n_block + c.idx] != c.cnt[q]) r = q;
    }
    return r;
}

// WHICH 0 = pop of the input queue (I1..I3), 1 = the conditional push (P1..P3)
template <int WHICH>
static __device__ __forceinline__ void k_ld_fill_queue(const VB& vb, const LdSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    const LdSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = LD_REGION_STRIDE(capacity);
    constexpr int R1 = WHICH == 0 ? LD_ROW_I1 : LD_ROW_P1;
    u64* trace = job.trace;
    if (i < capacity) {
        LdCycle c;
        ld_cycle(job, i, c);
        u64 enc[20], old[4], out4[4];
#pragma unroll
        for (int k = 0; k < 20; k++) enc[k] = c.can_pop ? job.in_enc[20 * c.idx + k] : 0;
        if (WHICH == 0) {
            ld_prev_head(job, c, old);
        } else {
            const int r = ld_route(job, c);
#pragma unroll
            for (int k = 0; k < 4; k++) old[k] = 0;
#pragma unroll
            for (int q = 0; q < 6; q++)
                if (r == q) ld_queue_tail(job, c, q, old);
        }
        es_queue_op(trace, n_rows, (size_t)R1 * rs + i, (size_t)(R1 + 1) * rs + i, (size_t)(R1 + 2) * rs + i, enc, old, out4);
        if (!job.tail_clean) for (int r = 0; r < 3; r++)
            for (int col = 130; col < LD_G + LD_L; col++) TR(col, (size_t)(R1 + r) * rs + i) = 0;
    } else if (i < rs) {
        if (!job.tail_clean) for (int r = 0; r < 3; r++) zero_gap_row_n(trace, n_rows, (size_t)(R1 + r) * rs + i, LD_G + LD_L);
    }
}

#define LD_XC(col, v) TR(col, row) = cur.v;
#define LD_XP(col, v) TR(col, row) = prev.v;
#define LD_XG(col, v)
#define LD_SET4(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; } while (0)
#define LD_BYTES4(dst, pfx, x) do { const u32 _x = (u32)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; dst.pfx##_b3 = _x >> 24; } while (0)
#define LD_IS_ZERO(x, w, z) do { const u64 _d = gl::canon(x); cur.z = _d == 0; cur.w = _d ? gl::inv(_d) : 0; } while (0)
#define LD_COLS4(ROW, v) {LD_##ROW##_##v##0, LD_##ROW##_##v##1, LD_##ROW##_##v##2, LD_##ROW##_##v##3}
// applies M(q, index) to the six output queues in ZKW_DEMUX_* order
#define LD_QUEUES(M) M(st, 0) M(ev, 1) M(l1, 2) M(kc, 3) M(sh, 4) M(ec, 5)

template <int ROW>
static __device__ __forceinline__ void k_ld_fill_row(const VB& vb, const LdSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const LdSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = LD_REGION_STRIDE(capacity);
    u64* trace = job.trace;
    if (i < capacity) {
        const size_t row = (size_t)ROW * rs + i;
        LdCycle c;
        ld_cycle(job, i, c);
        LdVars cur, prev;
        const u64 can_pop = c.can_pop ? 1 : 0;
        cur.can_pop = can_pop;
        cur.one = 1;
        if (ROW >= LD_ROW_X0 && ROW <= LD_ROW_R) {
            u64 es[8];  // words 10..17
#pragma unroll
            for (int k = 0; k < 8; k++) es[k] = c.can_pop ? job.in_enc[20 * c.idx + 10 + k] : 0;
            cur.es10 = es[0]; cur.es11 = es[1]; cur.es12 = es[2]; cur.es13 = es[3]; cur.es14 = es[4]; cur.es15 = es[5]; cur.es16 = es[6]; cur.es17 = es[7];
            cur.es19 = c.can_pop ? job.in_enc[20 * c.idx + 19] : 0;
            LD_BYTES4(cur, w10, es[0]); LD_BYTES4(cur, w11, es[1]); LD_BYTES4(cur, w12, es[2]); LD_BYTES4(cur, w13, es[3]);
            LD_BYTES4(cur, w14, es[4]); LD_BYTES4(cur, w15, es[5]); LD_BYTES4(cur, w16, es[6]); LD_BYTES4(cur, w17, es[7]);
            u32 ab[20];
            cur.kb30 = (es[0] >> 32) & 0xFF; cur.kb31 = (es[0] >> 40) & 0xFF;
            ab[0] = (es[0] >> 48) & 0xFF;
#pragma unroll
            for (int k = 1; k <= 6; k++)
#pragma unroll
                for (int j = 0; j < 3; j++) ab[1 + 3 * (k - 1) + j] = (es[k] >> (32 + 8 * j)) & 0xFF;
            ab[19] = (es[7] >> 32) & 0xFF;
            cur.aux = (es[7] >> 40) & 0xFF; cur.shard = (es[7] >> 48) & 0xFF;
            cur.a0 = ab[0]; cur.a1 = ab[1]; cur.a2 = ab[2]; cur.a3 = ab[3]; cur.a4 = ab[4]; cur.a5 = ab[5]; cur.a6 = ab[6]; cur.a7 = ab[7];
            cur.a8 = ab[8]; cur.a9 = ab[9]; cur.a10 = ab[10]; cur.a11 = ab[11]; cur.a12 = ab[12]; cur.a13 = ab[13]; cur.a14 = ab[14];
            cur.a15 = ab[15]; cur.a16 = ab[16]; cur.a17 = ab[17]; cur.a18 = ab[18]; cur.a19 = ab[19];
            if (ROW == LD_ROW_R) {
                LD_IS_ZERO(cur.aux, w_st, is_st);
                LD_IS_ZERO(gl::sub(cur.aux, 1), w_ev, is_ev);
                LD_IS_ZERO(gl::sub(cur.aux, 2), w_l1, is_l1);
                LD_IS_ZERO(gl::sub(cur.aux, 3), w_pre, is_pre);
                u64 hsum = 0;
#pragma unroll
                for (int k = 4; k < 20; k++) hsum += ab[k];
                LD_IS_ZERO(hsum, w_hz, hz);
                const u64 limb0 = (u64)ab[0] | (u64)ab[1] << 8 | (u64)ab[2] << 16 | (u64)ab[3] << 24;
                LD_IS_ZERO(gl::sub(limb0, 0x8010), w_akc, eq_kc);
                LD_IS_ZERO(gl::sub(limb0, 0x02), w_ash, eq_sh);
                LD_IS_ZERO(gl::sub(limb0, 0x01), w_aec, eq_ec);
                cur.r_st = can_pop & cur.is_st; cur.r_ev = can_pop & cur.is_ev; cur.r_l1 = can_pop & cur.is_l1;
                cur.pre_hz = can_pop & cur.is_pre & cur.hz;
                cur.r_kc = cur.pre_hz & cur.eq_kc; cur.r_sh = cur.pre_hz & cur.eq_sh; cur.r_ec = cur.pre_hz & cur.eq_ec;
            }
        }
        if (ROW == LD_ROW_Q) {
            const int r = ld_route(job, c);
            prev.len_i = c.p_len;
            cur.w_li = c.p_len ? gl::inv(c.p_len) : 0; cur.z_li = c.p_len == 0;
            cur.len_i = c.p_len - can_pop;
            constexpr int I3O[4] = LD_COLS4(I3, i3o), P3O[4] = LD_COLS4(P3, p3o);
            const size_t rI3 = (size_t)LD_ROW_I3 * rs + i, rP3 = (size_t)LD_ROW_P3 * rs + i;
            u64 io[4], po[4], ph[4], nh[4], sel[4] = {0, 0, 0, 0};
            ld_prev_head(job, c, ph);
#pragma unroll
            for (int k = 0; k < 4; k++) { io[k] = TR(I3O[k], rI3); po[k] = TR(P3O[k], rP3); nh[k] = c.can_pop ? io[k] : ph[k]; }
            LD_SET4(cur, i3o, io); LD_SET4(cur, p3o, po); LD_SET4(prev, ih, ph); LD_SET4(cur, ih, nh);
#define LD_Q_ONE(q, n) { u64 t[4], o[4]; ld_queue_tail(job, c, n, t); const bool hit = r == n; cur.r_##q = hit ? 1 : 0; \
            for (int k = 0; k < 4; k++) { o[k] = hit ? po[k] : t[k]; if (hit) sel[k] = t[k]; } \
            LD_SET4(prev, qt_##q, t); LD_SET4(cur, qt_##q, o); prev.ql_##q = c.cnt[n]; cur.ql_##q = c.cnt[n] + (hit ? 1 : 0); }
            LD_QUEUES(LD_Q_ONE)
#undef LD_Q_ONE
            LD_SET4(cur, sel, sel);
        }
#define LD_ROWCASE(R) if (ROW == LD_ROW_##R) { LD_FILL_##R(LD_XC, LD_XP, LD_XG, LD_XC) }
        LD_ROWCASE(X0) LD_ROWCASE(X1) LD_ROWCASE(X2) LD_ROWCASE(X3) LD_ROWCASE(R) LD_ROWCASE(Q)
#undef LD_ROWCASE
        constexpr int NSL[] = {0, 0, 0, 0, 0, 0, LD_NSLOTS_X0, LD_NSLOTS_X1, LD_NSLOTS_X2, LD_NSLOTS_X3, LD_NSLOTS_R, LD_NSLOTS_Q};
        constexpr int NLK[] = {0, 0, 0, 0, 0, 0, LD_NLOOK_X0, LD_NLOOK_X1, LD_NLOOK_X2, LD_NLOOK_X3, LD_NLOOK_R, LD_NLOOK_Q};
        if (!job.tail_clean) for (int col = NSL[ROW]; col < LD_G; col++) TR(col, row) = 0;
        if (!job.tail_clean) for (int col = LD_G + NLK[ROW]; col < LD_G + LD_L; col++) TR(col, row) = 0;
        for (int col = LD_G; col < LD_G + NLK[ROW]; col++) atomicAdd(&sh_hist[(u32)TR(col, row) & 0xFF], 1u);
    } else if (i < rs) {
        if (!job.tail_clean) zero_gap_row_n(trace, n_rows, (size_t)ROW * rs + i, LD_G + LD_L);
    }
    hist_flush(sh_hist, job.hist);
}

constexpr int LD_BOUNDARY_ROWS = (LD_NUM_ROW_TYPES - LD_ROWS_PER_CYCLE + 1) & ~1;  // register rows, PI, flush rows, the closed-form section (rounded up to even: 16-byte stores below)
__device__ __forceinline__ void ld_boundary_block(const LdSynthJob& job, u32 capacity, size_t n_rows);
static __device__ __forceinline__ void k_ld_fill_tail(const VB& vb, const LdSynthJob* __restrict__ jobs, u32 n_jobs, u32 capacity, size_t n_rows) {
    // 1-D grid: the first n_jobs blocks fill the boundary rows of one trace each (dispatched first and at raised priority: a chain of a dozen
    // dependent permutations that the other blocks' stores hide), then (LD_G + LD_L + 1) * TAIL_CHUNKS blocks per trace
    if (vb.x < n_jobs) {
        __builtin_amdgcn_s_setprio(3);
        ld_boundary_block(jobs[vb.x], capacity, n_rows);
        return;
    }
    constexpr u32 PER_JOB = (LD_G + LD_L + 1) * TAIL_CHUNKS;
    const u32 bid = (vb.x - n_jobs) % PER_JOB;
    const LdSynthJob& job = jobs[(vb.x - n_jobs) / PER_JOB];
    u64* trace = job.trace;
    const int col = bid / TAIL_CHUNKS, ch = bid % TAIL_CHUNKS;
    if (col < LD_G + LD_L) {
        if (job.tail_clean) return;
        const size_t bnd = (size_t)LD_BOUNDARY_ROW(capacity) + LD_BOUNDARY_ROWS;
        const size_t n_pairs = (n_rows - bnd) / 2;
        const size_t per = (n_pairs + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
        ulonglong2* c2 = reinterpret_cast<ulonglong2*>(trace + (size_t)col * n_rows + bnd);
        const ulonglong2 z = make_ulonglong2(0, 0);
        for (size_t k = lo + threadIdx.x; k < hi; k += 256) c2[k] = z;
        return;
    }
    u64* mlt = trace + (size_t)LD_MULT_COL * n_rows;
    const size_t per = (n_rows + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_rows ? lo + per : n_rows;
    for (size_t r = lo + threadIdx.x; r < (job.tail_clean && hi > 256 ? (lo < 256 ? 256 : lo) : hi); r += 256) {  // (a clean slot: rows >= 256 of the column are still zero)
        u64 v = 0;
        if (r < 256) {
            v = job.hist[r];
            if (r == 0) v += (u64)LD_L * n_rows - (u64)LD_LOOKUPS_PER_CYCLE * capacity;
        }
        mlt[r] = v;
    }
}

// BND_IN, BND_OUT, PI (runs last on the stream: reads the last cycle's row Q)
__device__ __forceinline__ void ld_fill_register_rows(const LdSynthJob& job, u32 capacity, size_t n_rows) {
    u64* trace = job.trace;
    const zkw_log_demux_instance* in = job.inst;
    const size_t rs = LD_REGION_STRIDE(capacity), bnd = (size_t)LD_BOUNDARY_ROW(capacity);
    LdVars cur;
    {
        const size_t row = bnd + LD_ROWOFF_BND_IN;
        LdCycle c;
        ld_cycle(job, 0, c);
        u64 h[4];
        ld_prev_head(job, c, h);
        LD_SET4(cur, ih, h);
        cur.len_i = c.p_len;
#define LD_B_ONE(q, n) { u64 t[4]; ld_queue_tail(job, c, n, t); LD_SET4(cur, qt_##q, t); cur.ql_##q = c.cnt[n]; }
        LD_QUEUES(LD_B_ONE)
#undef LD_B_ONE
#define LD_XPB(col, v)
        LD_FILL_BND_IN(LD_XC, LD_XPB, LD_XG, LD_XC)
        for (int col = LD_NSLOTS_BND_IN; col < LD_G + LD_L; col++) TR(col, row) = 0;
    }
    {
        const size_t row = bnd + LD_ROWOFF_BND_OUT, rQ = (size_t)LD_ROW_Q * rs + capacity - 1;
        constexpr int IH[4] = LD_COLS4(Q, ih);
        u64 t4[4];
        for (int k = 0; k < 4; k++) t4[k] = TR(IH[k], rQ);
        LD_SET4(cur, ih, t4);
        cur.len_i = TR(LD_Q_len_i, rQ);
#define LD_B_ONE(q, n) { constexpr int QT[4] = LD_COLS4(Q, qt_##q); for (int k = 0; k < 4; k++) t4[k] = TR(QT[k], rQ); \
        LD_SET4(cur, qt_##q, t4); cur.ql_##q = TR(LD_Q_ql_##q, rQ); }
        LD_QUEUES(LD_B_ONE)
#undef LD_B_ONE
        LD_SET4(cur, tail_i, in->initial_log_queue_state.tail);
        cur.completion = in->completion_flag ? 1 : 0;
        cur.w_end = gl::canon(cur.len_i) ? gl::inv(cur.len_i) : 0; cur.z_end = cur.len_i == 0;
        LD_FILL_BND_OUT(LD_XC, LD_XPB, LD_XG, LD_XC)
        for (int col = LD_NSLOTS_BND_OUT; col < LD_G + LD_L; col++) TR(col, row) = 0;
    }
}

// the register rows (one lane), then the closed-form section down to the PI row (runs last on the stream: reads the last cycle's rows)
// (the extra block of k_ld_fill_tail, whose other blocks zero the rows BELOW the boundary rows: the boundary rows' cells are zeroed here first)
__device__ __forceinline__ void ld_boundary_block(const LdSynthJob& job, u32 capacity, size_t n_rows) {
    {
        u64* trace = job.trace;
        const size_t bnd = (size_t)LD_BOUNDARY_ROW(capacity);
        for (int k = threadIdx.x; k < (LD_G + LD_L) * LD_BOUNDARY_ROWS; k += CF_THREADS) TR(k / LD_BOUNDARY_ROWS, bnd + k % LD_BOUNDARY_ROWS) = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) ld_fill_register_rows(job, capacity, n_rows);
    cf_section_from_records<CfLogDemux, SpecLogDemux>(job.first_inst, job.inst, job.trace, n_rows, (size_t)LD_BOUNDARY_ROW(capacity), [](int, size_t) {});
}

#undef TR
}  // namespace zkw
