/* events_sorter_circuit.c — TEST INFRASTRUCTURE: CPU restatement of EventsSorter / L1MessagesSorter synthesis
 * ("zkw trace v2", include/zkw_events_sorter_circuit_spec.h) — the counterpart of ZkSyncBaseLayerCircuit::synthesis
 * for those two instance types (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:286-323, wrapper
 * base_layer/events_sort_dedup.rs:28-39). Sequential: the registers are carried cycle by cycle; cells are scattered
 * through the generated ES_FILL_<row> lists. The satisfiability check (circuit_check.c) shares no code with it. */
#include "oracle.h"
#include "../include/zkw_events_sorter_circuit_spec.h"
#include <stdlib.h>
#include <string.h>

#define P ZKW_GOLDILOCKS_P
#define CELL(col, row) trace[(size_t)(col) * n_rows + (row)]

typedef struct {
#define X(n) uint64_t n;
    ES_VARS(X)
#undef X
} es_vars;

static uint64_t inv_or_zero(uint64_t x) { return x % P ? orc_gl_inv(x) : 0; }

#define SET4(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; } while (0)
#define GET4(arr, src, pfx) do { (arr)[0] = src.pfx##0; (arr)[1] = src.pfx##1; (arr)[2] = src.pfx##2; (arr)[3] = src.pfx##3; } while (0)
#define SET20(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; \
    dst.pfx##4 = (src)[4]; dst.pfx##5 = (src)[5]; dst.pfx##6 = (src)[6]; dst.pfx##7 = (src)[7]; dst.pfx##8 = (src)[8]; dst.pfx##9 = (src)[9]; \
    dst.pfx##10 = (src)[10]; dst.pfx##11 = (src)[11]; dst.pfx##12 = (src)[12]; dst.pfx##13 = (src)[13]; dst.pfx##14 = (src)[14]; \
    dst.pfx##15 = (src)[15]; dst.pfx##16 = (src)[16]; dst.pfx##17 = (src)[17]; dst.pfx##18 = (src)[18]; dst.pfx##19 = (src)[19]; } while (0)
#define GET20(arr, src, pfx) do { (arr)[0] = src.pfx##0; (arr)[1] = src.pfx##1; (arr)[2] = src.pfx##2; (arr)[3] = src.pfx##3; \
    (arr)[4] = src.pfx##4; (arr)[5] = src.pfx##5; (arr)[6] = src.pfx##6; (arr)[7] = src.pfx##7; (arr)[8] = src.pfx##8; (arr)[9] = src.pfx##9; \
    (arr)[10] = src.pfx##10; (arr)[11] = src.pfx##11; (arr)[12] = src.pfx##12; (arr)[13] = src.pfx##13; (arr)[14] = src.pfx##14; \
    (arr)[15] = src.pfx##15; (arr)[16] = src.pfx##16; (arr)[17] = src.pfx##17; (arr)[18] = src.pfx##18; (arr)[19] = src.pfx##19; } while (0)
#define BYTES4(dst, pfx, x) do { uint32_t _x = (uint32_t)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; dst.pfx##_b3 = _x >> 24; } while (0)
#define BYTES3(dst, pfx, x) do { uint32_t _x = (uint32_t)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; } while (0)

/* three permutations of one 4-wide queue operation (circuit_encodings/src/lib.rs:179-221): rows r1, r2, r3 get their
   130 gate cells; returns the new 4-word state */
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

static zkw_log_query normalized(const zkw_log_query *p) { /* events_sort_dedup.rs:541-553 */
    zkw_log_query r;
    memset(&r, 0, sizeof r);
    r.tx_number_in_block = p->tx_number_in_block;
    r.shard_id = p->shard_id;
    memcpy(r.address, p->address, sizeof r.address);
    memcpy(r.key, p->key, sizeof r.key);
    memcpy(r.written_value, p->written_value, sizeof r.written_value);
    r.is_service = p->is_service;
    return r;
}

/* rq_tail_in / rq_len_in: state of the result queue before the block (NULL / 0 = empty), read for the first instance */
int orc_events_sorter_synthesize(const zkw_events_sorter_instance *inst, const zkw_log_query *sorted_q, const uint64_t *unsorted_enc,
                                 const uint64_t *sorted_enc, const uint64_t *challenges /* [2][21] */, const uint64_t *rq_tail_in,
                                 uint32_t rq_len_in, const uint64_t *public_input /* [4] or NULL */, uint32_t capacity, size_t n_rows, uint64_t *trace) {
    if (ES_MIN_ROWS(capacity) > n_rows) return -1;
    const size_t first = inst->first_item, m = inst->num_items;
    if (m > capacity) return -2;
    const zkw_events_sorter_fsm *fi = &inst->hidden_fsm_input;
    const int start = inst->start_flag != 0;
    const size_t rs = (size_t)ES_REGION_STRIDE(capacity), bnd = (size_t)ES_BOUNDARY_ROW(capacity);
    es_vars prev, cur, glob;
    memset(&prev, 0, sizeof prev);
    memset(&glob, 0, sizeof glob);
    const uint64_t zero4[4] = {0};

    SET4(prev, uh, start ? inst->initial_log_queue_state.head : fi->initial_unsorted_queue_state.head);
    SET4(prev, sh, start ? inst->intermediate_sorted_queue_state.head : fi->intermediate_sorted_queue_state.head);
    prev.len_u = start ? inst->initial_log_queue_state.length : fi->initial_unsorted_queue_state.length;
    prev.len_s = start ? inst->intermediate_sorted_queue_state.length : fi->intermediate_sorted_queue_state.length;
    SET4(prev, rh, start ? (rq_tail_in ? rq_tail_in : zero4) : fi->final_result_queue_state.tail);
    prev.len_r = start ? rq_len_in : fi->final_result_queue_state.length;
    /* the first instance starts its accumulators at ONE whatever the (placeholder) FSM input says: the empty-queue
       dummy instance has a zero FSM input and ONE in its output (events_sort_dedup.rs:27-76) */
    prev.lhs0 = start ? 1 : fi->lhs_accumulator[0]; prev.lhs1 = start ? 1 : fi->lhs_accumulator[1];
    prev.rhs0 = start ? 1 : fi->rhs_accumulator[0]; prev.rhs1 = start ? 1 : fi->rhs_accumulator[1];
    prev.kts = fi->previous_key;
    prev.krb = fi->previous_item.rollback ? 1 : 0;
    prev.valid = start ? 0 : 1;
    if (!start) {
        zkw_log_query nq = normalized(&fi->previous_item);
        uint64_t e[20];
        orc_encode_log_queries(&nq, 1, NULL, e);
        SET20(prev, ne, e);
    }
    {
        uint64_t *g = &glob.c0_1; /* c0_1..c0_20, c1_1..c1_20 are consecutive fields (generated in that order) */
        for (int r = 0; r < 2; r++)
            for (int k = 1; k <= 20; k++) g[20 * r + k - 1] = challenges[21 * r + k];
    }

#define XC(col, v) CELL(col, row) = cur.v;
#define XP(col, v) CELL(col, row) = prev.v;
#define XG(col, v) CELL(col, row) = glob.v;
#define XSKIP(col, v)
#define XC_L(col, v) if ((col) >= ES_G) CELL(col, row) = cur.v;
    {
        const size_t row = bnd + ES_ROWOFF_BND_IN;
        cur = prev;
        ES_FILL_BND_IN(XC, XP, XG, XC)
    }

    for (size_t i = 0; i < capacity; i++) {
        const size_t idx = first + i;
        const int can_pop = i < m;
        memset(&cur, 0, sizeof cur);
        zkw_log_query q;
        memset(&q, 0, sizeof q);
        uint64_t eu[20] = {0}, es[20] = {0}, old[4], o4[4], pne[20];
        if (can_pop) { q = sorted_q[idx]; memcpy(eu, unsorted_enc + 20 * idx, 160); memcpy(es, sorted_enc + 20 * idx, 160); }
        cur.can_pop = can_pop;
        SET20(cur, eu, eu);
        SET20(cur, es, es);
        /* the three queue operations */
        GET4(old, prev, uh);
        queue_op(trace, n_rows, (size_t)ES_ROW_U1 * rs + i, (size_t)ES_ROW_U2 * rs + i, (size_t)ES_ROW_U3 * rs + i, eu, old, o4);
        SET4(cur, u3o, o4);
        GET4(old, prev, sh);
        queue_op(trace, n_rows, (size_t)ES_ROW_S1 * rs + i, (size_t)ES_ROW_S2 * rs + i, (size_t)ES_ROW_S3 * rs + i, es, old, o4);
        SET4(cur, s3o, o4);
        GET20(pne, prev, ne);
        GET4(old, prev, rh);
        queue_op(trace, n_rows, (size_t)ES_ROW_R1 * rs + i, (size_t)ES_ROW_R2 * rs + i, (size_t)ES_ROW_R3 * rs + i, pne, old, o4);
        SET4(cur, r3o, o4);
        /* grand products, W = 20 */
        for (int r = 0; r < 2; r++) {
            const uint64_t *ch = challenges + 21 * r;
            uint64_t lc = orc_gl_add(ch[20], eu[0] % P), rc = orc_gl_add(ch[20], es[0] % P);
            for (int k = 1; k < 20; k++) {
                lc = orc_gl_add(lc, orc_gl_mul(eu[k] % P, ch[k]));
                rc = orc_gl_add(rc, orc_gl_mul(es[k] % P, ch[k]));
            }
            const uint64_t pl = r ? prev.lhs1 : prev.lhs0, pr = r ? prev.rhs1 : prev.rhs0;
            const uint64_t nl = orc_gl_mul(pl, lc), nr = orc_gl_mul(pr, rc);
            if (r == 0) { cur.lc0 = lc; cur.rc0 = rc; cur.nl0 = nl; cur.nr0 = nr; cur.lhs0 = can_pop ? nl : pl; cur.rhs0 = can_pop ? nr : pr; }
            else { cur.lc1 = lc; cur.rc1 = rc; cur.nl1 = nl; cur.nr1 = nr; cur.lhs1 = can_pop ? nl : pl; cur.rhs1 = can_pop ? nr : pr; }
        }
        /* the split of the sorted record's encoding and its normalised form */
        uint64_t cn[20];
        {
            uint32_t rv[8], kb[8];
            for (int k = 0; k < 8; k++) { rv[k] = (uint32_t)es[k]; kb[k] = (uint32_t)(es[k] >> 32); cn[k] = es[k] - rv[k]; }
            cur.rv0 = rv[0]; cur.rv1 = rv[1]; cur.rv2 = rv[2]; cur.rv3 = rv[3]; cur.rv4 = rv[4]; cur.rv5 = rv[5]; cur.rv6 = rv[6]; cur.rv7 = rv[7];
            BYTES4(cur, rv0, rv[0]); BYTES4(cur, rv1, rv[1]); BYTES4(cur, rv2, rv[2]); BYTES4(cur, rv3, rv[3]);
            BYTES4(cur, rv4, rv[4]); BYTES4(cur, rv5, rv[5]); BYTES4(cur, rv6, rv[6]); BYTES4(cur, rv7, rv[7]);
            BYTES3(cur, kb0, kb[0]); BYTES3(cur, kb1, kb[1]); BYTES3(cur, kb2, kb[2]); BYTES3(cur, kb3, kb[3]);
            BYTES3(cur, kb4, kb[4]); BYTES3(cur, kb5, kb[5]); BYTES3(cur, kb6, kb[6]); BYTES3(cur, kb7, kb[7]);
            for (int k = 8; k < 16; k++) cn[k] = es[k];
            cur.ts = (uint32_t)es[16];
            BYTES4(cur, ts, cur.ts); BYTES3(cur, a16, (uint32_t)(es[16] >> 32));
            cn[16] = es[16] - cur.ts;
            cur.tx = (uint32_t)es[17];
            BYTES4(cur, tx, cur.tx);
            cur.a19 = (es[17] >> 32) & 0xFF; cur.aux = (es[17] >> 40) & 0xFF; cur.shard = (es[17] >> 48) & 0xFF;
            cn[17] = es[17] - (cur.aux << 40);
            cur.rw = es[18] & 1; cur.sv = (es[18] >> 1) & 1;
            cn[18] = 2 * cur.sv;
            cur.rb = es[19] & 1;
            cn[19] = 0;
            SET20(cur, cn, cn);
        }
        /* timestamp order and the dedup rule */
        {
            const int64_t t = (int64_t)cur.ts - (int64_t)prev.kts;
            cur.bw = t < 0;
            cur.dts = (uint64_t)(t + (cur.bw ? (1ll << 32) : 0));
            BYTES4(cur, dts, cur.dts);
            const uint64_t d = orc_gl_sub(cur.ts, prev.kts);
            cur.same_ts = d == 0; cur.w_ts = inv_or_zero(d);
            if (can_pop && prev.valid) {
                if (cur.bw) return -3;
                if (cur.same_ts && (!cur.rb || prev.krb)) return -4;
                if (!cur.same_ts && cur.rb) return -5;
            }
            if (can_pop && !prev.valid && cur.rb) return -6;
            cur.push = can_pop && prev.valid && !cur.same_ts && !prev.krb;
            cur.valid = prev.valid | (uint64_t)can_pop;
            cur.kts = can_pop ? cur.ts : prev.kts;
            cur.krb = can_pop ? cur.rb : prev.krb;
        }
        /* queues and registers */
        {
            const uint64_t d = prev.len_u % P;
            cur.z_lu = d == 0; cur.w_lu = inv_or_zero(d); cur.z_ls = cur.z_lu; cur.w_ls = cur.w_lu;
            cur.len_u = prev.len_u - can_pop; cur.len_s = prev.len_s - can_pop; cur.len_r = prev.len_r + cur.push;
            uint64_t a[4], b[4], o[4], pn[20], on[20];
            GET4(a, cur, u3o); GET4(b, prev, uh);
            for (int k = 0; k < 4; k++) o[k] = can_pop ? a[k] : b[k];
            SET4(cur, uh, o);
            GET4(a, cur, s3o); GET4(b, prev, sh);
            for (int k = 0; k < 4; k++) o[k] = can_pop ? a[k] : b[k];
            SET4(cur, sh, o);
            GET4(a, cur, r3o); GET4(b, prev, rh);
            for (int k = 0; k < 4; k++) o[k] = cur.push ? a[k] : b[k];
            SET4(cur, rh, o);
            GET20(pn, prev, ne);
            for (int k = 0; k < 20; k++) on[k] = can_pop ? cn[k] : pn[k];
            SET20(cur, ne, on);
        }
        /* scatter (the Poseidon rows' gate cells are in place; their lookup columns carry the range checks of the record's 70 bytes,
           whose relations are row NTV's: tools/gen_events_sorter_circuit.py) */
#define ROWAT(R) const size_t row = (size_t)(R) * rs + i;
        { ROWAT(ES_ROW_A) ES_FILL_A(XC, XP, XG, XC) }
        { ROWAT(ES_ROW_NTV) ES_FILL_NTV(XC, XP, XG, XC) }
        { ROWAT(ES_ROW_U1) ES_LOOK_U1(XC) } { ROWAT(ES_ROW_U2) ES_LOOK_U2(XC) } { ROWAT(ES_ROW_U3) ES_LOOK_U3(XC) }
        { ROWAT(ES_ROW_S1) ES_LOOK_S1(XC) } { ROWAT(ES_ROW_S2) ES_LOOK_S2(XC) } { ROWAT(ES_ROW_S3) ES_LOOK_S3(XC) }
        { ROWAT(ES_ROW_R1) ES_LOOK_R1(XC) } { ROWAT(ES_ROW_R2) ES_LOOK_R2(XC) } { ROWAT(ES_ROW_R3) ES_LOOK_R3(XC) }
        { ROWAT(ES_ROW_W) ES_FILL_W(XC, XP, XG, XC) }
        { ROWAT(ES_ROW_Q) ES_FILL_Q(XC, XP, XG, XC) }
        prev = cur;
    }

    {
        const size_t row = bnd + ES_ROWOFF_BND_OUT;
        cur = prev;
        SET4(cur, tail_u, inst->initial_log_queue_state.tail);
        SET4(cur, tail_s, inst->intermediate_sorted_queue_state.tail);
        cur.completion = inst->completion_flag ? 1 : 0;
        { const uint64_t d = cur.len_u % P; cur.z_end = d == 0; cur.w_end = inv_or_zero(d); }
        cur.flush = cur.completion & cur.valid & (1 - cur.krb);
        uint64_t pn[20], old[4], o4[4], o[4];
        GET20(pn, cur, ne);
        GET4(old, cur, rh);
        queue_op(trace, n_rows, bnd + ES_ROWOFF_F1, bnd + ES_ROWOFF_F2, bnd + ES_ROWOFF_F3, pn, old, o4);
        SET4(cur, f3o, o4);
        for (int k = 0; k < 4; k++) o[k] = cur.flush ? o4[k] : old[k];
        SET4(cur, final_rh, o);
        cur.final_len_r = cur.len_r + cur.flush;
        ES_FILL_BND_OUT(XC, XP, XG, XC)
        if (cur.completion && !cur.z_end) return -7;
    }

    (void)public_input; /* the PI row is derived by the closed-form section (orc_es_fill_closed_form, closed_form_fill.c), which runs next */
    for (int t = 0; t < 256; t++) CELL(ES_MULT_COL, t) = 0;
    for (int c = ES_G; c < ES_G + ES_L; c++)
        for (size_t r = 0; r < n_rows; r++) {
            uint64_t v = CELL(c, r);
            if (v > 255) return -9;
            CELL(ES_MULT_COL, v) += 1;
        }
    return 0;
}
