/* storage_sorter_circuit.c — TEST INFRASTRUCTURE: CPU restatement of StorageSorter synthesis ("zkw trace v2",
 * include/zkw_storage_sorter_circuit_spec.h) — the counterpart of ZkSyncBaseLayerCircuit::synthesis for that instance
 * type (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:286-323, wrapper base_layer/storage_sort_dedup.rs:
 * 29-40; witness src/witness/individual_circuits/storage_sort_dedup.rs:12-703). Sequential: the registers (queue
 * heads, accumulators, key riders, the cell state machine) are carried cycle by cycle and everything is derived from
 * the two queues' ENCODINGS (not from the builder's sorted records or its per-cell scans). Cells are scattered through
 * the generated SS_FILL_<row> lists. The satisfiability check (circuit_check.c) shares no code with it. */
#include "oracle.h"
#include "../include/zkw_storage_sorter_circuit_spec.h"
#include <stdlib.h>
#include <string.h>

#define P ZKW_GOLDILOCKS_P
#define CELL(col, row) trace[(size_t)(col) * n_rows + (row)]

typedef struct {
#define X(n) uint64_t n;
    SS_VARS(X)
#undef X
} ss_vars;

/* the register file, as arrays */
typedef struct {
    uint64_t uh[4], sh[4], rh[4], len_u, len_s, len_r, lhs[2], rhs[2], kc[18], ksh, kts, cidx, valid, depth, has, base[8], cur[8];
} ss_regs;

static uint64_t inv_or_zero(uint64_t x) { return x % P ? orc_gl_inv(x) : 0; }

#define I4(M) M(0) M(1) M(2) M(3)
#define I8(M) I4(M) M(4) M(5) M(6) M(7)
#define I16(M) I8(M) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define I17(M) I16(M) M(16)
#define I18(M) I17(M) M(17)
#define I20(M) I18(M) M(18) M(19)

static void regs_to_vars(const ss_regs *r, ss_vars *v) {
#define M(k) v->uh##k = r->uh[k]; v->sh##k = r->sh[k]; v->rh##k = r->rh[k];
    I4(M)
#undef M
    v->len_u = r->len_u; v->len_s = r->len_s; v->len_r = r->len_r;
    v->lhs0 = r->lhs[0]; v->lhs1 = r->lhs[1]; v->rhs0 = r->rhs[0]; v->rhs1 = r->rhs[1];
#define M(k) v->kc##k = r->kc[k];
    I18(M)
#undef M
    v->ksh = r->ksh; v->kts = r->kts; v->cidx = r->cidx; v->valid = r->valid; v->depth = r->depth; v->has = r->has;
#define M(k) v->base##k = r->base[k]; v->cur##k = r->cur[k];
    I8(M)
#undef M
}

/* three permutations of one 4-wide queue operation (circuit_encodings/src/lib.rs:179-221) */
static void queue_op(uint64_t *trace, size_t n_rows, size_t r1, size_t r2, size_t r3, const uint64_t enc[20], const uint64_t old[4],
                     uint64_t out4[4]) {
    uint64_t in[12], slots[130];
    memcpy(in, enc, 64); memset(in + 8, 0, 32);
    orc_poseidon2_flattened(in, slots);
    for (int k = 0; k < 130; k++) CELL(k, r1) = slots[k];
    memcpy(in, enc + 8, 64); memcpy(in + 8, slots + 118 + 8, 32);
    orc_poseidon2_flattened(in, slots);
    for (int k = 0; k < 130; k++) CELL(k, r2) = slots[k];
    memcpy(in, enc + 16, 32); memcpy(in + 4, old, 32); memcpy(in + 8, slots + 118 + 8, 32);
    orc_poseidon2_flattened(in, slots);
    for (int k = 0; k < 130; k++) CELL(k, r3) = slots[k];
    memcpy(out4, slots + 118, 32);
}

/* does the open cell emit a record, and which (storage_sort_dedup.rs:394-457): fills the eq / depth gadgets and the
   20 words of the record's encoding (log_query.rs:49-72, 118-196) */
typedef struct { uint64_t wq[8], eq[8], q1, eqv, w_d, z_d, em, wsel[8], w[20]; } ss_emit;
static void emit_record(const ss_regs *r, ss_emit *e) {
    e->q1 = 1; e->eqv = 1;
    for (int k = 0; k < 8; k++) {
        const uint64_t d = orc_gl_sub(r->cur[k], r->base[k]);
        e->eq[k] = d == 0; e->wq[k] = inv_or_zero(d);
        if (k < 4) e->q1 &= e->eq[k];
        e->eqv &= e->eq[k];
    }
    e->z_d = r->depth % P == 0; e->w_d = inv_or_zero(r->depth);
    e->em = e->z_d ? r->has : 1;
    for (int k = 0; k < 8; k++) {
        e->wsel[k] = e->z_d ? r->base[k] : r->cur[k];
        e->w[k] = r->base[k] + (r->kc[k] << 32);
        e->w[8 + k] = e->wsel[k] + (r->kc[8 + k] << 32);
    }
    e->w[16] = r->kc[16] << 32;
    e->w[17] = (r->kc[17] << 32) + (r->ksh << 48);
    e->w[18] = (!e->z_d && !e->eqv) ? 1 : 0;
    e->w[19] = 0;
}
#define EMIT_TO_VARS(v, e, W) do { \
    v.q1 = (e).q1; v.eqv = (e).eqv; v.w_d = (e).w_d; v.z_d = (e).z_d; v.em = (e).em; \
    v.wq0 = (e).wq[0]; v.wq1 = (e).wq[1]; v.wq2 = (e).wq[2]; v.wq3 = (e).wq[3]; v.wq4 = (e).wq[4]; v.wq5 = (e).wq[5]; v.wq6 = (e).wq[6]; v.wq7 = (e).wq[7]; \
    v.eq0 = (e).eq[0]; v.eq1 = (e).eq[1]; v.eq2 = (e).eq[2]; v.eq3 = (e).eq[3]; v.eq4 = (e).eq[4]; v.eq5 = (e).eq[5]; v.eq6 = (e).eq[6]; v.eq7 = (e).eq[7]; \
    v.wsel0 = (e).wsel[0]; v.wsel1 = (e).wsel[1]; v.wsel2 = (e).wsel[2]; v.wsel3 = (e).wsel[3]; v.wsel4 = (e).wsel[4]; v.wsel5 = (e).wsel[5]; \
    v.wsel6 = (e).wsel[6]; v.wsel7 = (e).wsel[7]; \
    v.W##0 = (e).w[0]; v.W##1 = (e).w[1]; v.W##2 = (e).w[2]; v.W##3 = (e).w[3]; v.W##4 = (e).w[4]; v.W##5 = (e).w[5]; v.W##6 = (e).w[6]; \
    v.W##7 = (e).w[7]; v.W##8 = (e).w[8]; v.W##9 = (e).w[9]; v.W##10 = (e).w[10]; v.W##11 = (e).w[11]; v.W##12 = (e).w[12]; v.W##13 = (e).w[13]; \
    v.W##14 = (e).w[14]; v.W##15 = (e).w[15]; v.W##16 = (e).w[16]; v.W##17 = (e).w[17]; v.W##18 = (e).w[18]; v.W##19 = (e).w[19]; } while (0)

int orc_storage_sorter_synthesize(const zkw_storage_sorter_instance *inst, const uint64_t *unsorted_enc, const uint64_t *sorted_enc,
                                  const uint64_t *challenges /* [2][21] */, const uint64_t *public_input /* [4] or NULL */, uint32_t capacity, size_t n_rows, uint64_t *trace) {
    if (SS_MIN_ROWS(capacity) > n_rows) return -1;
    const size_t first = inst->first_item, m = inst->num_items;
    if (m > capacity) return -2;
    const zkw_storage_sorter_fsm *fi = &inst->hidden_fsm_input;
    const int start = inst->start_flag != 0;
    const size_t rs = (size_t)SS_REGION_STRIDE(capacity), bnd = (size_t)SS_BOUNDARY_ROW(capacity);
    ss_regs pr, nr;
    ss_vars prev, cur, glob;
    memset(&pr, 0, sizeof pr);
    memset(&glob, 0, sizeof glob);

    if (start) {
        memcpy(pr.uh, inst->unsorted_log_queue_state.head, 32); pr.len_u = inst->unsorted_log_queue_state.length;
        memcpy(pr.sh, inst->intermediate_sorted_queue_state.head, 32); pr.len_s = inst->intermediate_sorted_queue_state.length;
    } else {
        memcpy(pr.uh, fi->current_unsorted_queue_state.head, 32); pr.len_u = fi->current_unsorted_queue_state.length;
        memcpy(pr.sh, fi->current_intermediate_sorted_queue_state.head, 32); pr.len_s = fi->current_intermediate_sorted_queue_state.length;
        memcpy(pr.rh, fi->current_final_sorted_queue_state.tail, 32); pr.len_r = fi->current_final_sorted_queue_state.length;
        uint8_t kb[52]; /* comparison_key (log_query.rs:82-92): key limbs then address limbs, little-endian bytes */
        for (int k = 0; k < 52; k++) kb[k] = (uint8_t)(fi->previous_packed_key[k / 4] >> (8 * (k % 4)));
        for (int k = 0; k < 17; k++) pr.kc[k] = (uint64_t)kb[3 * k] | (uint64_t)kb[3 * k + 1] << 8 | (uint64_t)kb[3 * k + 2] << 16;
        pr.kc[17] = kb[51];
        pr.ksh = inst->shard_id_to_process;
        pr.kts = fi->previous_timestamp;
        pr.valid = 1;
        pr.depth = fi->this_cell_current_depth;
        pr.has = fi->this_cell_has_explicit_read_and_rollback_depth_zero ? 1 : 0;
        for (int k = 0; k < 8; k++) { pr.base[k] = fi->this_cell_base_value[k]; pr.cur[k] = fi->this_cell_current_value[k]; }
    }
    pr.cidx = fi->cycle_idx;
    /* the first instance starts its accumulators at ONE whatever the (placeholder) FSM input says: the empty-queue
       dummy instance has a zero FSM input and ONE in its output (storage_sort_dedup.rs:23-70) */
    for (int r = 0; r < 2; r++) { pr.lhs[r] = start ? 1 : fi->lhs_accumulator[r]; pr.rhs[r] = start ? 1 : fi->rhs_accumulator[r]; }
    {
        uint64_t *g = &glob.c0_1; /* c0_1..c0_20, c1_1..c1_20 are consecutive fields (generated in that order) */
        for (int r = 0; r < 2; r++)
            for (int k = 1; k <= 20; k++) g[20 * r + k - 1] = challenges[21 * r + k];
    }

#define XC(col, v) CELL(col, row) = cur.v;
#define XP(col, v) CELL(col, row) = prev.v;
#define XG(col, v) CELL(col, row) = glob.v;
    memset(&prev, 0, sizeof prev);
    regs_to_vars(&pr, &prev);
    {
        const size_t row = bnd + SS_ROWOFF_BND_IN;
        cur = prev;
        SS_FILL_BND_IN(XC, XP, XG, XC)
    }

    for (size_t i = 0; i < capacity; i++) {
        const size_t idx = first + i;
        const int can_pop = i < m;
        if (can_pop != (pr.len_u != 0)) return -3;
        memset(&cur, 0, sizeof cur);
        nr = pr;
        uint64_t eu[20] = {0}, es[20] = {0}, o4[4];
        if (can_pop) { memcpy(eu, unsorted_enc + 20 * idx, 160); memcpy(es, sorted_enc + 20 * idx, 160); }
        cur.can_pop = can_pop;
#define M(k) cur.eu##k = eu[k]; cur.es##k = es[k];
        I20(M)
#undef M
        /* the two pops */
        queue_op(trace, n_rows, (size_t)SS_ROW_U1 * rs + i, (size_t)SS_ROW_U2 * rs + i, (size_t)SS_ROW_U3 * rs + i, eu, pr.uh, o4);
#define M(k) cur.u3o##k = o4[k]; nr.uh[k] = can_pop ? o4[k] : pr.uh[k];
        I4(M)
#undef M
        queue_op(trace, n_rows, (size_t)SS_ROW_S1 * rs + i, (size_t)SS_ROW_S2 * rs + i, (size_t)SS_ROW_S3 * rs + i, es, pr.sh, o4);
#define M(k) cur.s3o##k = o4[k]; nr.sh[k] = can_pop ? o4[k] : pr.sh[k];
        I4(M)
#undef M
        /* grand products, W = 20; the unsorted side carries its queue position in word 19 (storage_sort_dedup.rs:128-143) */
        for (int r = 0; r < 2; r++) {
            const uint64_t *ch = challenges + 21 * r;
            uint64_t lc = orc_gl_add(ch[20], eu[0] % P), rc = orc_gl_add(ch[20], es[0] % P);
            for (int k = 1; k < 20; k++) {
                lc = orc_gl_add(lc, orc_gl_mul(eu[k] % P, ch[k]));
                rc = orc_gl_add(rc, orc_gl_mul(es[k] % P, ch[k]));
            }
            if (can_pop) lc = orc_gl_add(lc, orc_gl_mul(orc_gl_mul(256, pr.cidx % P), ch[19]));
            const uint64_t nl = orc_gl_mul(pr.lhs[r], lc), nrr = orc_gl_mul(pr.rhs[r], rc);
            nr.lhs[r] = can_pop ? nl : pr.lhs[r]; nr.rhs[r] = can_pop ? nrr : pr.rhs[r];
            if (r == 0) { cur.lc0 = lc; cur.rc0 = rc; cur.nl0 = nl; cur.nr0 = nrr; }
            else { cur.lc1 = lc; cur.rc1 = rc; cur.nl1 = nl; cur.nr1 = nrr; }
        }
        /* the split of the sorted record */
        uint64_t lo[17], cv[18];
        for (int k = 0; k < 17; k++) {
            if (es[k] >> 56) return -4; /* not an encoding */
            lo[k] = es[k] & 0xFFFFFFFFull; cv[k] = es[k] >> 32;
        }
        if (es[17] >> 56 || es[18] > 3) return -4;
        cv[17] = (es[17] >> 32) & 0xFF;
        const uint64_t tx = es[17] & 0xFFFFFFFFull, aux = (es[17] >> 40) & 0xFF, shard = (es[17] >> 48) & 0xFF;
        const uint64_t rw = es[18] & 1, sv = es[18] >> 1, rb = es[19] & 1, ts = es[19] >> 8;
        if (ts >> 32 || (es[19] & 0xFE)) return -4;
#define M(k) cur.lo##k = lo[k]; cur.lo##k##_b0 = lo[k] & 0xFF; cur.lo##k##_b1 = (lo[k] >> 8) & 0xFF; cur.lo##k##_b2 = (lo[k] >> 16) & 0xFF; \
        cur.lo##k##_b3 = lo[k] >> 24; cur.c##k##_b0 = cv[k] & 0xFF; cur.c##k##_b1 = (cv[k] >> 8) & 0xFF; cur.c##k##_b2 = cv[k] >> 16;
        I17(M)
#undef M
#define M(k) cur.c##k = cv[k];
        I18(M)
#undef M
        cur.tx_b0 = tx & 0xFF; cur.tx_b1 = (tx >> 8) & 0xFF; cur.tx_b2 = (tx >> 16) & 0xFF; cur.tx_b3 = tx >> 24;
        cur.aux = aux; cur.shard = shard; cur.rw = rw; cur.sv = sv; cur.rb = rb; cur.ts = ts;
        cur.ts_b0 = ts & 0xFF; cur.ts_b1 = (ts >> 8) & 0xFF; cur.ts_b2 = (ts >> 16) & 0xFF; cur.ts_b3 = ts >> 24;
        /* order: riders from the top, then the extended timestamp */
        uint64_t wk[18], ek[18], pe[17], keq, diff;
        for (int k = 0; k < 18; k++) { const uint64_t d = orc_gl_sub(cv[k], pr.kc[k]); ek[k] = d == 0; wk[k] = inv_or_zero(d); }
        pe[16] = ek[17];
        for (int k = 15; k >= 0; k--) pe[k] = pe[k + 1] & ek[k + 1];
        keq = pe[0] & ek[0];
        diff = orc_gl_sub(cv[17], pr.kc[17]);
        for (int k = 16; k >= 0; k--)
            if (pe[k]) diff = orc_gl_add(diff, orc_gl_sub(cv[k], pr.kc[k]));
        if (keq) diff = orc_gl_add(diff, orc_gl_sub(ts, pr.kts));
#define M(k) cur.wk##k = wk[k]; cur.ek##k = ek[k];
        I18(M)
#undef M
#define M(k) cur.pe##k = pe[k];
        I16(M)
#undef M
        cur.keq = keq; cur.diff = diff;
        if (can_pop && pr.valid) {
            const uint64_t d = orc_gl_sub(diff, 1);
            if (d >> 32) return -5; /* not sorted by (address, key, extended timestamp) */
            cur.d_b0 = d & 0xFF; cur.d_b1 = (d >> 8) & 0xFF; cur.d_b2 = (d >> 16) & 0xFF; cur.d_b3 = d >> 24;
        }
        /* does the previous cell emit a record? */
        ss_emit em;
        emit_record(&pr, &em);
        EMIT_TO_VARS(cur, em, pw);
        cur.nkey = (can_pop && pr.valid && !keq) ? 1 : 0;
        cur.push = cur.nkey & em.em;
        queue_op(trace, n_rows, (size_t)SS_ROW_R1 * rs + i, (size_t)SS_ROW_R2 * rs + i, (size_t)SS_ROW_R3 * rs + i, em.w, pr.rh, o4);
#define M(k) cur.r3o##k = o4[k]; nr.rh[k] = cur.push ? o4[k] : pr.rh[k];
        I4(M)
#undef M
        nr.len_r = pr.len_r + cur.push;
        /* the cell state machine (storage_sort_dedup.rs:339-534) */
        cur.same = pr.valid & keq; cur.sm = can_pop & cur.same; cur.nc = can_pop - cur.sm;
        cur.wr = rw & (1 - rb); cur.rbk = rw & rb;
        if (cur.nc && cur.rbk) return -6;
        if (cur.sm && cur.rbk && em.z_d) return -7;
        uint64_t t[8];
        for (int k = 0; k < 8; k++) {
            t[k] = cur.wr ? lo[8 + k] : lo[k];
            if (cur.sm && !cur.rbk && lo[k] != pr.cur[k]) return -8;
            if (cur.sm && cur.rbk && lo[8 + k] != pr.cur[k]) return -9;
            nr.cur[k] = can_pop ? t[k] : pr.cur[k];
            nr.base[k] = cur.nc ? lo[k] : pr.base[k];
        }
        cur.t0 = t[0]; cur.t1 = t[1]; cur.t2 = t[2]; cur.t3 = t[3]; cur.t4 = t[4]; cur.t5 = t[5]; cur.t6 = t[6]; cur.t7 = t[7];
        nr.depth = cur.nc ? rw : pr.depth + (cur.sm ? cur.wr : 0) - (cur.sm ? cur.rbk : 0);
        cur.u = cur.sm & (1 - rw) & em.z_d;
        nr.has = cur.nc ? 1 - rw : (pr.has | cur.u);
        nr.valid = pr.valid | (uint64_t)can_pop;
        /* queues and key registers */
        { const uint64_t d = pr.len_u % P; cur.z_lu = d == 0; cur.w_lu = inv_or_zero(d); cur.z_ls = cur.z_lu; cur.w_ls = cur.w_lu; }
        nr.len_u = pr.len_u - can_pop; nr.len_s = pr.len_s - can_pop;
        for (int k = 0; k < 18; k++) nr.kc[k] = can_pop ? cv[k] : pr.kc[k];
        nr.ksh = can_pop ? shard : pr.ksh;
        nr.kts = can_pop ? ts : pr.kts;
        nr.cidx = pr.cidx + 1;
        regs_to_vars(&nr, &cur);
#define ROWAT(R) const size_t row = (size_t)(R) * rs + i;
        { ROWAT(SS_ROW_A) SS_FILL_A(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_X0) SS_FILL_X0(XC, XP, XG, XC) } { ROWAT(SS_ROW_X1) SS_FILL_X1(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_X2) SS_FILL_X2(XC, XP, XG, XC) } { ROWAT(SS_ROW_X3) SS_FILL_X3(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_X4) SS_FILL_X4(XC, XP, XG, XC) } { ROWAT(SS_ROW_X5) SS_FILL_X5(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_X6) SS_FILL_X6(XC, XP, XG, XC) } { ROWAT(SS_ROW_X7) SS_FILL_X7(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_K) SS_FILL_K(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_C1) SS_FILL_C1(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_C2) SS_FILL_C2(XC, XP, XG, XC) }
        { ROWAT(SS_ROW_Q) SS_FILL_Q(XC, XP, XG, XC) }
        prev = cur;
        pr = nr;
    }

    {
        const size_t row = bnd + SS_ROWOFF_BND_OUT;
        memset(&cur, 0, sizeof cur);
        regs_to_vars(&pr, &cur);
#define M(k) cur.tail_u##k = inst->unsorted_log_queue_state.tail[k]; cur.tail_s##k = inst->intermediate_sorted_queue_state.tail[k];
        I4(M)
#undef M
        cur.completion = inst->completion_flag ? 1 : 0;
        { const uint64_t d = cur.len_u % P; cur.z_end = d == 0; cur.w_end = inv_or_zero(d); }
        ss_emit em;
        uint64_t o4[4];
        emit_record(&pr, &em);
        EMIT_TO_VARS(cur, em, fw);
        cur.flush = cur.completion & pr.valid & em.em;
        queue_op(trace, n_rows, bnd + SS_ROWOFF_F1, bnd + SS_ROWOFF_F2, bnd + SS_ROWOFF_F3, em.w, pr.rh, o4);
#define M(k) cur.f3o##k = o4[k]; cur.final_rh##k = cur.flush ? o4[k] : pr.rh[k];
        I4(M)
#undef M
        cur.final_len_r = pr.len_r + cur.flush;
        SS_FILL_BND_OUT(XC, XP, XG, XC)
        if (cur.completion && !cur.z_end) return -10;
        if (cur.completion && (pr.lhs[0] != pr.rhs[0] || pr.lhs[1] != pr.rhs[1])) return -11;
    }

    (void)public_input; /* the PI row is derived by the closed-form section (orc_ss_fill_closed_form, closed_form_fill.c), which runs next */
    for (int t = 0; t < 256; t++) CELL(SS_MULT_COL, t) = 0;
    for (int c = SS_G; c < SS_G + SS_L; c++)
        for (size_t r = 0; r < n_rows; r++) {
            uint64_t v = CELL(c, r);
            if (v > 255) return -12;
            CELL(SS_MULT_COL, v) += 1;
        }
    return 0;
}
