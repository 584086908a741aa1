// events_sorter_circuit_kernels.cuh — synthesis of the EventsSorter / L1MessagesSorter trace ("zkw trace v2", circuit
// types 11 and 12, include/zkw_events_sorter_circuit_spec.h) on gfx950.
//
// Counterpart of ZkSyncBaseLayerCircuit::synthesis for those instance types (circuit_definitions/src/circuit_definitions/
// base_layer/mod.rs:286-323, wrapper base_layer/events_sort_dedup.rs:28-39); the witness it materialises is the output of
// compute_events_dedup_and_sort (src/witness/individual_circuits/events_sort_dedup.rs:16-580).
//
// One lane per cycle, region-major rows, no carried state: the result queue after p pushes is result_new_tails[p-1]
// with p = kept'-prefix of the previous item (a record is pushed when its successor has another timestamp and it is
// a forward record), the normalised encoding of the latest popped record is recomputed from that record. A queue
// operation of the 4-wide log queue is three dependent permutations (lib.rs:179-221): one lane runs all three and
// writes the three Poseidon2 rows. Cells of the general rows are scattered through the generated ES_FILL_<row> lists.
#pragma once
#include "scan_kernels.cuh"
#include "decommit_sorter_circuit_kernels.cuh"
#include "log_kernels.cuh"
#include "../../include/zkw_events_sorter_circuit_spec.h"

namespace zkw {

struct EsSynthJob {
    const zkw_events_sorter_instance* inst;
    const zkw_log_query* sorted_q;             // block-wide arrays of the builder, indexed by item
    const u64 *unsorted_enc, *sorted_enc;      // [n][20]
    const u64 *unsorted_new_tails, *sorted_new_tails;  // [n][4]
    const u64* result_new_tails;               // [n_result][4]
    const u32* kept_prefix;                    // [n + 1]: kept'(j) for j < k (see above)
    const u64* challenges;                     // [2][21]
    const u64 *lhs_z, *rhs_z;                  // [2][n]
    u64 n_block;
    u64 rq_tail_in[4];                         // result queue before the block
    u32 rq_len_in;
    const u64* public_input;  // [4]: commitment of the compact closed-form input (not written: the closed-form section derives the PI row)
    const zkw_events_sorter_instance* first_inst;  // the block's first instance (the shared observable input)
    u64* trace;
    u32* hist;
    u32 tail_clean;  // the slot already holds this layout (same circuit, capacity, rows): every cell that is zero in EVERY trace of the layout (the padding rows, the gap rows of a region, the columns a row type does not use, multiplicity rows >= 256) is still zero: the fills skip those stores
};

struct EsVars {
#define X(n) u64 n;
    ES_VARS(X)
#undef X
};

#define TR(col, row) trace[(size_t)(col) * n_rows + (row)]
#define ES_SET4(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; } while (0)
#define ES_SET20(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; \
    dst.pfx##4 = (src)[4]; dst.pfx##5 = (src)[5]; dst.pfx##6 = (src)[6]; dst.pfx##7 = (src)[7]; dst.pfx##8 = (src)[8]; dst.pfx##9 = (src)[9]; \
    dst.pfx##10 = (src)[10]; dst.pfx##11 = (src)[11]; dst.pfx##12 = (src)[12]; dst.pfx##13 = (src)[13]; dst.pfx##14 = (src)[14]; \
    dst.pfx##15 = (src)[15]; dst.pfx##16 = (src)[16]; dst.pfx##17 = (src)[17]; dst.pfx##18 = (src)[18]; dst.pfx##19 = (src)[19]; } while (0)
#define ES_BYTES4(dst, pfx, x) do { const u32 _x = (u32)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; dst.pfx##_b3 = _x >> 24; } while (0)
#define ES_BYTES3(dst, pfx, x) do { const u32 _x = (u32)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; } while (0)
#define ES_COLS4(ROW, v) {ES_##ROW##_##v##0, ES_##ROW##_##v##1, ES_##ROW##_##v##2, ES_##ROW##_##v##3}

// normalised encoding of an encoded record: read value, timestamp, aux byte, rw and rollback flags cleared
__device__ __forceinline__ void es_normalise(const u64 es[20], u64 cn[20]) {
#pragma unroll
    for (int k = 0; k < 8; k++) cn[k] = es[k] & 0xFFFFFFFF00000000ull;
#pragma unroll
    for (int k = 8; k < 16; k++) cn[k] = es[k];
    cn[16] = es[16] & 0xFFFFFFFF00000000ull;
    cn[17] = es[17] & ~(0xFFull << 40);
    cn[18] = es[18] & 2;
    cn[19] = 0;
}

struct EsRegsIn {
    const u64 *uh, *sh;  // [4]
    u64 rh[4], lhs[2], rhs[2];
    u32 len, len_r, valid, kts, krb;
};
__device__ __forceinline__ void es_regs_in(const EsSynthJob& job, EsRegsIn& r) {
    const zkw_events_sorter_instance* in = job.inst;
    const zkw_events_sorter_fsm& f = in->hidden_fsm_input;
    const bool start = in->start_flag != 0;
    r.uh = start ? in->initial_log_queue_state.head : f.initial_unsorted_queue_state.head;
    r.sh = start ? in->intermediate_sorted_queue_state.head : f.intermediate_sorted_queue_state.head;
    r.len = start ? in->initial_log_queue_state.length : f.initial_unsorted_queue_state.length;
#pragma unroll
    for (int k = 0; k < 4; k++) r.rh[k] = start ? job.rq_tail_in[k] : f.final_result_queue_state.tail[k];
    r.len_r = start ? job.rq_len_in : f.final_result_queue_state.length;
#pragma unroll
    for (int k = 0; k < 2; k++) { r.lhs[k] = start ? 1 : f.lhs_accumulator[k]; r.rhs[k] = start ? 1 : f.rhs_accumulator[k]; }  // ONE at the start
    r.valid = start ? 0 : 1;
    r.kts = f.previous_key;
    r.krb = f.previous_item.rollback ? 1 : 0;
}

struct EsCycle {
    bool can_pop, fresh;
    size_t idx, last_popped;  // last_popped: item popped last before this cycle (valid when i > 0)
    u32 p_valid, p_kts, p_krb;
    u64 pushes_before;
};
__device__ __forceinline__ void es_cycle(const EsSynthJob& job, const EsRegsIn& ri, u32 i, EsCycle& c) {
    const size_t first = job.inst->first_item, m = job.inst->num_items;
    c.can_pop = i < m;
    c.idx = first + i;
    c.last_popped = first + (i - 1 < m ? i - 1 : m - 1);
    c.fresh = i == 0 || m == 0;  // nothing popped in this instance yet (an empty instance never pops): registers = FSM input
    c.p_valid = c.fresh ? ri.valid : 1;
    if (c.fresh) { c.p_kts = ri.kts; c.p_krb = ri.krb; }
    else { const zkw_log_query* pq = job.sorted_q + c.last_popped; c.p_kts = pq->timestamp; c.p_krb = pq->rollback ? 1 : 0; }
    const size_t tt = first + (i < m ? i : m);
    c.pushes_before = tt ? job.kept_prefix[tt - 1] : 0;
}
// the normalised encoding of the latest popped record before this cycle (zeros before the very first record)
__device__ __forceinline__ void es_prev_ne(const EsSynthJob& job, const EsCycle& c, u32 i, u64 ne[20]) {
    u64 e[20];
    if (c.fresh) {
        if (job.inst->start_flag) {
#pragma unroll
            for (int k = 0; k < 20; k++) ne[k] = 0;
            return;
        }
        encode_log_query(job.inst->hidden_fsm_input.previous_item, false, 0, e);
    } else {
#pragma unroll
        for (int k = 0; k < 20; k++) e[k] = job.sorted_enc[20 * c.last_popped + k];
    }
    es_normalise(e, ne);
}
__device__ __forceinline__ void es_prev_rh(const EsSynthJob& job, const EsCycle& c, u64 rh[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) rh[k] = c.pushes_before ? job.result_new_tails[4 * (c.pushes_before - 1) + k] : job.rq_tail_in[k];
}

// three permutations of one 4-wide queue operation into rows r1, r2, r3; returns the new 4-word state in out4
__device__ __forceinline__ void es_queue_op(u64* trace, size_t n_rows, size_t r1, size_t r2, size_t r3, const u64 enc[20],
                                            const u64 old[4], u64 out4[4]) {
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = enc[k];
#pragma unroll
    for (int k = 8; k < 12; k++) s[k] = 0;
    fill_flattened_poseidon(trace, n_rows, r1, s);
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = enc[8 + k];
    fill_flattened_poseidon(trace, n_rows, r2, s);
#pragma unroll
    for (int k = 0; k < 4; k++) { s[k] = enc[16 + k]; s[4 + k] = old[k]; }
    fill_flattened_poseidon(trace, n_rows, r3, s);
#pragma unroll
    for (int k = 0; k < 4; k++) out4[k] = gl::canon(s[k]);
}

// WHICH 0 = unsorted pop (U1..U3), 1 = sorted pop (S1..S3), 2 = result push (R1..R3)
template <int WHICH>
static __device__ __forceinline__ void k_es_fill_queue(const VB& vb, const EsSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    const EsSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = ES_REGION_STRIDE(capacity);
    constexpr int R1 = WHICH == 0 ? ES_ROW_U1 : (WHICH == 1 ? ES_ROW_S1 : ES_ROW_R1);
    u64* trace = job.trace;
    if (i < capacity) {
        EsRegsIn ri;
        es_regs_in(job, ri);
        EsCycle c;
        es_cycle(job, ri, i, c);
        u64 enc[20], old[4], out4[4];
        if (WHICH == 2) {
            es_prev_ne(job, c, i, enc);
            es_prev_rh(job, c, old);
        } else {
            const u64* src = WHICH == 0 ? job.unsorted_enc : job.sorted_enc;
            const u64* tails = WHICH == 0 ? job.unsorted_new_tails : job.sorted_new_tails;
#pragma unroll
            for (int k = 0; k < 20; k++) enc[k] = c.can_pop ? src[20 * c.idx + k] : 0;
            const u64* ph = c.fresh ? (WHICH == 0 ? ri.uh : ri.sh) : tails + 4 * c.last_popped;
#pragma unroll
            for (int k = 0; k < 4; k++) old[k] = ph[k];
        }
        es_queue_op(trace, n_rows, (size_t)R1 * rs + i, (size_t)(R1 + 1) * rs + i, (size_t)(R1 + 2) * rs + i, enc, old, out4);
        // (the rows' lookup cells — range checks of the record's bytes — are written by k_es_fill_row<NTV>, which runs after this kernel;
        //  only the cells no lookup uses are cleared here: row R3 holds 6 of 8)
        if (!job.tail_clean && WHICH == 2)
            for (int col = ES_G + ES_NLOOK_R3; col < ES_G + ES_L; col++) TR(col, (size_t)(R1 + 2) * rs + i) = 0;
    } else if (i < rs) {
        if (!job.tail_clean) for (int r = 0; r < 3; r++) zero_gap_row_n(trace, n_rows, (size_t)(R1 + r) * rs + i, ES_G + ES_L);
    }
}

#define ES_XC(col, v) TR(col, row) = cur.v;
#define ES_XP(col, v) TR(col, row) = prev.v;
#define ES_XG(col, v) TR(col, row) = glob.v;

template <int ROW>
static __device__ __forceinline__ void k_es_fill_row(const VB& vb, const EsSynthJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    __shared__ u32 sh_hist[256];
    sh_hist[threadIdx.x] = 0;
    __syncthreads();
    const EsSynthJob& job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    const size_t rs = ES_REGION_STRIDE(capacity);
    u64* trace = job.trace;
    if (i < capacity) {
        const size_t row = (size_t)ROW * rs + i, n = job.n_block;
        EsRegsIn ri;
        es_regs_in(job, ri);
        EsCycle c;
        es_cycle(job, ri, i, c);
        const size_t m = job.inst->num_items;
        EsVars cur, prev, glob;
        const u64 can_pop = c.can_pop ? 1 : 0;
        cur.can_pop = can_pop;
        u64 es[20], cn[20];
#pragma unroll
        for (int k = 0; k < 20; k++) es[k] = c.can_pop ? job.sorted_enc[20 * c.idx + k] : 0;
        ES_SET20(cur, es, es);
        es_normalise(es, cn);
        ES_SET20(cur, cn, cn);
        // fields of this record
        const u32 ts = (u32)es[16], rb = (u32)(es[19] & 1);
        cur.ts = ts; cur.rb = rb;
        prev.kts = c.p_kts; prev.krb = c.p_krb; prev.valid = c.p_valid;
        const bool same_ts = ts == c.p_kts;
        const bool push = c.can_pop && c.p_valid && !same_ts && !c.p_krb;
        cur.push = push;
        if (ROW == ES_ROW_A) {
            u64 eu[20];
#pragma unroll
            for (int k = 0; k < 20; k++) eu[k] = c.can_pop ? job.unsorted_enc[20 * c.idx + k] : 0;
            ES_SET20(cur, eu, eu);
            u64* g = &glob.c0_1;  // c0_1..c0_20, c1_1..c1_20 are consecutive fields
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const u64* ch = job.challenges + 21 * r;
#pragma unroll
                for (int k = 1; k <= 20; k++) g[20 * r + k - 1] = ch[k];
                u64 lc = gl::add(ch[20], eu[0]), rc = gl::add(ch[20], es[0]);
#pragma unroll
                for (int k = 1; k < 20; k++) { lc = gl::add(lc, gl::mul(eu[k], ch[k])); rc = gl::add(rc, gl::mul(es[k], ch[k])); }
                const u64 pl = c.fresh ? ri.lhs[r] : job.lhs_z[(size_t)r * n + c.last_popped];
                const u64 pr = c.fresh ? ri.rhs[r] : job.rhs_z[(size_t)r * n + c.last_popped];
                const u64 nl = gl::canon(gl::mul(pl, lc)), nr = gl::canon(gl::mul(pr, rc));
                lc = gl::canon(lc); rc = gl::canon(rc);
                if (r == 0) { cur.lc0 = lc; cur.rc0 = rc; cur.nl0 = nl; cur.nr0 = nr; prev.lhs0 = pl; prev.rhs0 = pr; cur.lhs0 = can_pop ? nl : pl; cur.rhs0 = can_pop ? nr : pr; }
                else { cur.lc1 = lc; cur.rc1 = rc; cur.nl1 = nl; cur.nr1 = nr; prev.lhs1 = pl; prev.rhs1 = pr; cur.lhs1 = can_pop ? nl : pl; cur.rhs1 = can_pop ? nr : pr; }
            }
        }
        if (ROW == ES_ROW_NTV) {  // the relations that split the sorted record's encoding (the former rows N0..N7, T, V)
            const u32 rv[8] = {(u32)es[0], (u32)es[1], (u32)es[2], (u32)es[3], (u32)es[4], (u32)es[5], (u32)es[6], (u32)es[7]};
            const u32 kb[8] = {(u32)(es[0] >> 32), (u32)(es[1] >> 32), (u32)(es[2] >> 32), (u32)(es[3] >> 32), (u32)(es[4] >> 32),
                               (u32)(es[5] >> 32), (u32)(es[6] >> 32), (u32)(es[7] >> 32)};
            cur.rv0 = rv[0]; cur.rv1 = rv[1]; cur.rv2 = rv[2]; cur.rv3 = rv[3]; cur.rv4 = rv[4]; cur.rv5 = rv[5]; cur.rv6 = rv[6]; cur.rv7 = rv[7];
            ES_BYTES4(cur, rv0, rv[0]); ES_BYTES4(cur, rv1, rv[1]); ES_BYTES4(cur, rv2, rv[2]); ES_BYTES4(cur, rv3, rv[3]);
            ES_BYTES4(cur, rv4, rv[4]); ES_BYTES4(cur, rv5, rv[5]); ES_BYTES4(cur, rv6, rv[6]); ES_BYTES4(cur, rv7, rv[7]);
            ES_BYTES3(cur, kb0, kb[0]); ES_BYTES3(cur, kb1, kb[1]); ES_BYTES3(cur, kb2, kb[2]); ES_BYTES3(cur, kb3, kb[3]);
            ES_BYTES3(cur, kb4, kb[4]); ES_BYTES3(cur, kb5, kb[5]); ES_BYTES3(cur, kb6, kb[6]); ES_BYTES3(cur, kb7, kb[7]);
            ES_BYTES4(cur, ts, ts); ES_BYTES3(cur, a16, (u32)(es[16] >> 32));
            cur.tx = (u32)es[17];
            ES_BYTES4(cur, tx, (u32)es[17]);
            cur.a19 = (es[17] >> 32) & 0xFF; cur.aux = (es[17] >> 40) & 0xFF; cur.shard = (es[17] >> 48) & 0xFF;
            cur.rw = es[18] & 1; cur.sv = (es[18] >> 1) & 1;
            // the 70 bytes are range-checked in the lookup columns of the nine Poseidon2 rows (eight per row: the flattened gate leaves
            // them free); this kernel runs after the queue kernels and owns those cells
#define ES_LOOKROW(R) { const size_t row = (size_t)ES_ROW_##R * rs + i; ES_LOOK_##R(ES_XL) }
#define ES_XL(col, v) { TR(col, row) = cur.v; atomicAdd(&sh_hist[(u32)cur.v & 0xFF], 1u); }
            ES_LOOKROW(U1) ES_LOOKROW(U2) ES_LOOKROW(U3) ES_LOOKROW(S1) ES_LOOKROW(S2) ES_LOOKROW(S3) ES_LOOKROW(R1) ES_LOOKROW(R2) ES_LOOKROW(R3)
#undef ES_XL
#undef ES_LOOKROW
        }
        if (ROW == ES_ROW_W) {
            const u64 t = (u64)ts - (u64)c.p_kts;  // wraps below zero
            cur.bw = t >> 63;
            cur.dts = (u32)t;
            ES_BYTES4(cur, dts, (u32)t);
            const u64 d = gl::canon(gl::sub((u64)ts, (u64)c.p_kts));
            cur.same_ts = d == 0; cur.w_ts = d ? gl::inv(d) : 0;
            cur.valid = c.p_valid | (u32)can_pop;
            cur.kts = c.can_pop ? ts : c.p_kts;
            cur.krb = c.can_pop ? rb : c.p_krb;
            u64 rh[4], r3o[4], o[4];
            es_prev_rh(job, c, rh);
            constexpr int R3O[4] = ES_COLS4(R3, r3o);
            const size_t rR3 = (size_t)ES_ROW_R3 * rs + i;
#pragma unroll
            for (int k = 0; k < 4; k++) { r3o[k] = TR(R3O[k], rR3); o[k] = push ? r3o[k] : rh[k]; }
            ES_SET4(cur, r3o, r3o); ES_SET4(prev, rh, rh); ES_SET4(cur, rh, o);
            prev.len_r = (u64)job.rq_len_in + c.pushes_before;
            cur.len_r = prev.len_r + (push ? 1 : 0);
        }
        if (ROW == ES_ROW_Q) {
            const u64 p_len = (u64)ri.len - (i < m ? i : m);
            prev.len_u = p_len; prev.len_s = p_len;
            cur.w_lu = p_len ? gl::inv(p_len) : 0; cur.z_lu = p_len == 0; cur.w_ls = cur.w_lu; cur.z_ls = cur.z_lu;
            cur.len_u = p_len - can_pop; cur.len_s = p_len - can_pop;
            constexpr int U3O[4] = ES_COLS4(U3, u3o), S3O[4] = ES_COLS4(S3, s3o);
            const size_t rU3 = (size_t)ES_ROW_U3 * rs + i, rS3 = (size_t)ES_ROW_S3 * rs + i;
            u64 uo[4], so[4], pu[4], ps[4], ou[4], os[4], pne[20], one[20];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uo[k] = TR(U3O[k], rU3); so[k] = TR(S3O[k], rS3);
                pu[k] = c.fresh ? ri.uh[k] : job.unsorted_new_tails[4 * c.last_popped + k];
                ps[k] = c.fresh ? ri.sh[k] : job.sorted_new_tails[4 * c.last_popped + k];
                ou[k] = c.can_pop ? uo[k] : pu[k]; os[k] = c.can_pop ? so[k] : ps[k];
            }
            ES_SET4(cur, u3o, uo); ES_SET4(cur, s3o, so); ES_SET4(prev, uh, pu); ES_SET4(prev, sh, ps); ES_SET4(cur, uh, ou); ES_SET4(cur, sh, os);
            es_prev_ne(job, c, i, pne);
#pragma unroll
            for (int k = 0; k < 20; k++) one[k] = c.can_pop ? cn[k] : pne[k];
            ES_SET20(prev, ne, pne); ES_SET20(cur, ne, one);
        }
#define ES_ROWCASE(R) if (ROW == ES_ROW_##R) { ES_FILL_##R(ES_XC, ES_XP, ES_XG, ES_XC) }
        ES_ROWCASE(A) ES_ROWCASE(NTV) ES_ROWCASE(W) ES_ROWCASE(Q)
#undef ES_ROWCASE
        constexpr int NSL[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, ES_NSLOTS_A, ES_NSLOTS_NTV, ES_NSLOTS_W, ES_NSLOTS_Q};
        constexpr int NLK[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, ES_NLOOK_A, ES_NLOOK_NTV, ES_NLOOK_W, ES_NLOOK_Q};
        static_assert(ES_ROW_Q == 12 && ES_ROW_A == 9, "row order of the generated spec");
        if (!job.tail_clean) for (int col = NSL[ROW]; col < ES_G; col++) TR(col, row) = 0;
        if (!job.tail_clean) for (int col = ES_G + NLK[ROW]; col < ES_G + ES_L; col++) TR(col, row) = 0;
        for (int col = ES_G; col < ES_G + NLK[ROW]; col++) atomicAdd(&sh_hist[(u32)TR(col, row) & 0xFF], 1u);
    } else if (i < rs) {
        if (!job.tail_clean) zero_gap_row_n(trace, n_rows, (size_t)ROW * rs + i, ES_G + ES_L);
    }
    if (ROW == ES_ROW_A && vb.x == 0 && threadIdx.x == 0) {  // the closed-form section's lookup cells: the key / address bytes of the FSM records' previous_item (bridge rows NIB* / NOB*)
        const zkw_log_query& a = job.inst->hidden_fsm_input.previous_item;
        const zkw_log_query& b = job.inst->hidden_fsm_output.previous_item;
        for (int k = 0; k < 8; k++) { hist_bytes(sh_hist, a.key[k]); hist_bytes(sh_hist, b.key[k]); }
        for (int k = 0; k < 5; k++) { hist_bytes(sh_hist, a.address[k]); hist_bytes(sh_hist, b.address[k]); }
    }
    hist_flush(sh_hist, job.hist);
}

constexpr int ES_BOUNDARY_ROWS = (ES_NUM_ROW_TYPES - ES_ROWS_PER_CYCLE + 1) & ~1;  // register rows, PI, flush rows, the closed-form section (rounded up to even: 16-byte stores below)
__device__ __forceinline__ void es_boundary_block(const EsSynthJob& job, u32 capacity, size_t n_rows);
static __device__ __forceinline__ void k_es_fill_tail(const VB& vb, const EsSynthJob* __restrict__ jobs, u32 n_jobs, u32 capacity, size_t n_rows) {
    // 1-D grid: the first n_jobs blocks fill the boundary rows of one trace each (dispatched first and at raised priority: a chain of a dozen
    // dependent permutations that the other blocks' stores hide), then (ES_G + ES_L + 1) * TAIL_CHUNKS blocks per trace
    if (vb.x < n_jobs) {
        __builtin_amdgcn_s_setprio(3);
        es_boundary_block(jobs[vb.x], capacity, n_rows);
        return;
    }
    constexpr u32 PER_JOB = (ES_G + ES_L + 1) * TAIL_CHUNKS;
    const u32 bid = (vb.x - n_jobs) % PER_JOB;
    const EsSynthJob& job = jobs[(vb.x - n_jobs) / PER_JOB];
    u64* trace = job.trace;
    const int col = bid / TAIL_CHUNKS, ch = bid % TAIL_CHUNKS;
    if (col < ES_G + ES_L) {
        if (job.tail_clean) return;
        const size_t bnd = (size_t)ES_BOUNDARY_ROW(capacity) + ES_BOUNDARY_ROWS;
        const size_t n_pairs = (n_rows - bnd) / 2;
        const size_t per = (n_pairs + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
        ulonglong2* c2 = reinterpret_cast<ulonglong2*>(trace + (size_t)col * n_rows + bnd);
        const ulonglong2 z = make_ulonglong2(0, 0);
        for (size_t k = lo + threadIdx.x; k < hi; k += 256) c2[k] = z;
        return;
    }
    u64* mlt = trace + (size_t)ES_MULT_COL * n_rows;
    const size_t per = (n_rows + TAIL_CHUNKS - 1) / TAIL_CHUNKS, lo = ch * per, hi = lo + per < n_rows ? lo + per : n_rows;
    for (size_t r = lo + threadIdx.x; r < (job.tail_clean && hi > 256 ? (lo < 256 ? 256 : lo) : hi); r += 256) {  // (a clean slot: rows >= 256 of the column are still zero)
        u64 v = 0;
        if (r < 256) {
            v = job.hist[r];
            if (r == 0) v += (u64)ES_L * n_rows - (u64)ES_LOOKUPS_PER_CYCLE * capacity - 4 * ES_CF_NUM_BYTES;  // (the section's byte cells: counted in job.hist by k_es_fill_row<A>)
        }
        mlt[r] = v;
    }
}

// BND_IN, BND_OUT, the flush permutations F1..F3, PI (runs last on the stream: reads the last cycle's rows)
__device__ __forceinline__ void es_fill_register_rows(const EsSynthJob& job, u32 capacity, size_t n_rows) {
    u64* trace = job.trace;
    const zkw_events_sorter_instance* in = job.inst;
    const size_t rs = ES_REGION_STRIDE(capacity), bnd = (size_t)ES_BOUNDARY_ROW(capacity);
    EsRegsIn ri;
    es_regs_in(job, ri);
    EsVars cur, glob;
    u64* g = &glob.c0_1;
    for (int r = 0; r < 2; r++)
        for (int k = 1; k <= 20; k++) g[20 * r + k - 1] = job.challenges[21 * r + k];
    EsCycle c0;
    es_cycle(job, ri, 0, c0);
    {
        const size_t row = bnd + ES_ROWOFF_BND_IN;
        u64 ne[20];
        es_prev_ne(job, c0, 0, ne);
        ES_SET4(cur, uh, ri.uh); ES_SET4(cur, sh, ri.sh); ES_SET4(cur, rh, ri.rh);
        cur.len_u = ri.len; cur.len_s = ri.len; cur.len_r = ri.len_r;
        cur.lhs0 = ri.lhs[0]; cur.lhs1 = ri.lhs[1]; cur.rhs0 = ri.rhs[0]; cur.rhs1 = ri.rhs[1];
        cur.kts = ri.kts; cur.krb = ri.krb; cur.valid = ri.valid;
        ES_SET20(cur, ne, ne);
#define ES_XPB(col, v)
        ES_FILL_BND_IN(ES_XC, ES_XPB, ES_XG, ES_XC)
        for (int col = ES_NSLOTS_BND_IN; col < ES_G + ES_L; col++) TR(col, row) = 0;
    }
    {
        const size_t row = bnd + ES_ROWOFF_BND_OUT, lc = capacity - 1;
        const size_t rA = (size_t)ES_ROW_A * rs + lc, rW = (size_t)ES_ROW_W * rs + lc, rQ = (size_t)ES_ROW_Q * rs + lc;
        constexpr int UH[4] = ES_COLS4(Q, uh), SH[4] = ES_COLS4(Q, sh), RH[4] = ES_COLS4(W, rh);
        constexpr int NE[20] = {ES_Q_ne0, ES_Q_ne1, ES_Q_ne2, ES_Q_ne3, ES_Q_ne4, ES_Q_ne5, ES_Q_ne6, ES_Q_ne7, ES_Q_ne8, ES_Q_ne9, ES_Q_ne10,
                                ES_Q_ne11, ES_Q_ne12, ES_Q_ne13, ES_Q_ne14, ES_Q_ne15, ES_Q_ne16, ES_Q_ne17, ES_Q_ne18, ES_Q_ne19};
        u64 t4[4], rh[4], ne[20];
        for (int k = 0; k < 4; k++) t4[k] = TR(UH[k], rQ);
        ES_SET4(cur, uh, t4);
        for (int k = 0; k < 4; k++) t4[k] = TR(SH[k], rQ);
        ES_SET4(cur, sh, t4);
        for (int k = 0; k < 4; k++) rh[k] = TR(RH[k], rW);
        ES_SET4(cur, rh, rh);
        cur.len_u = TR(ES_Q_len_u, rQ); cur.len_s = TR(ES_Q_len_s, rQ); cur.len_r = TR(ES_W_len_r, rW);
        cur.lhs0 = TR(ES_A_lhs0, rA); cur.lhs1 = TR(ES_A_lhs1, rA); cur.rhs0 = TR(ES_A_rhs0, rA); cur.rhs1 = TR(ES_A_rhs1, rA);
        cur.kts = TR(ES_W_kts, rW); cur.krb = TR(ES_W_krb, rW); cur.valid = TR(ES_W_valid, rW);
        for (int k = 0; k < 20; k++) ne[k] = TR(NE[k], rQ);
        ES_SET20(cur, ne, ne);
        ES_SET4(cur, tail_u, in->initial_log_queue_state.tail);
        ES_SET4(cur, tail_s, in->intermediate_sorted_queue_state.tail);
        cur.completion = in->completion_flag ? 1 : 0;
        cur.w_end = gl::canon(cur.len_u) ? gl::inv(cur.len_u) : 0; cur.z_end = cur.len_u == 0;
        cur.flush = cur.completion & cur.valid & (1 - cur.krb);
        u64 o4[4], fr[4];
        es_queue_op(trace, n_rows, bnd + ES_ROWOFF_F1, bnd + ES_ROWOFF_F2, bnd + ES_ROWOFF_F3, ne, rh, o4);
        for (int r = 0; r < 3; r++)
            for (int col = ES_G; col < ES_G + ES_L; col++) TR(col, bnd + ES_ROWOFF_F1 + r) = 0;
        for (int k = 0; k < 4; k++) fr[k] = cur.flush ? o4[k] : rh[k];
        ES_SET4(cur, f3o, o4); ES_SET4(cur, final_rh, fr);
        cur.final_len_r = cur.len_r + cur.flush;
        ES_FILL_BND_OUT(ES_XC, ES_XPB, ES_XG, ES_XC)
        for (int col = ES_NSLOTS_BND_OUT; col < ES_G + ES_L; col++) TR(col, row) = 0;
    }
}

// BND_IN, BND_OUT, the flush permutations F1..F3 (one lane), then the closed-form section down to the PI row (runs last on the stream: reads
// the last cycle's rows)
// (the extra block of k_es_fill_tail, whose other blocks zero the rows BELOW the boundary rows: the boundary rows' cells are zeroed here first)
__device__ __forceinline__ void es_boundary_block(const EsSynthJob& job, u32 capacity, size_t n_rows) {
    {
        u64* trace = job.trace;
        const size_t bnd = (size_t)ES_BOUNDARY_ROW(capacity);
        for (int k = threadIdx.x; k < (ES_G + ES_L) * ES_BOUNDARY_ROWS; k += CF_THREADS) TR(k / ES_BOUNDARY_ROWS, bnd + k % ES_BOUNDARY_ROWS) = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) es_fill_register_rows(job, capacity, n_rows);
    cf_section_from_records<CfEventsSorter, SpecEventsSorter>(job.first_inst, job.inst, job.trace, n_rows, (size_t)ES_BOUNDARY_ROW(capacity), [](int, size_t) {});
}

// kept_prefix[k] = #{ j < k : record j is a forward record whose successor has another timestamp }, k = 0..n
// (a record without a successor is never counted: it is flushed at the very end): the flag of flag_prefix (scan_kernels.cuh)
struct EsKeptFlag {
    const zkw_log_query* sorted_q;
    size_t n;
    __device__ u32 operator()(size_t j) const { return (j + 1 < n && !sorted_q[j].rollback && sorted_q[j + 1].timestamp != sorted_q[j].timestamp) ? 1u : 0u; }
};

#undef TR
}  // namespace zkw
