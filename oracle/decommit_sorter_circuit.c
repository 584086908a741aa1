/* decommit_sorter_circuit.c — TEST INFRASTRUCTURE: CPU restatement of CodeDecommittmentsSorter synthesis
 * ("zkw trace v2", include/zkw_decommit_sorter_circuit_spec.h) — the counterpart of
 * ZkSyncBaseLayerCircuit::synthesis for that instance type (circuit_definitions/src/circuit_definitions/base_layer/
 * mod.rs:286-323 with the wrapper base_layer/sort_code_decommits.rs:28-39). The fill walks the cycles sequentially,
 * carrying the registers the way a circuit body would; every cell is then scattered through the generated
 * DS_FILL_<row> lists. The satisfiability check (circuit_check.c) shares no code with it. */
#include "oracle.h"
#include "../include/zkw_decommit_sorter_circuit_spec.h"
#include <stdlib.h>
#include <string.h>

#define P ZKW_GOLDILOCKS_P
#define CELL(col, row) trace[(size_t)(col) * n_rows + (row)]

typedef struct {
#define X(n) uint64_t n;
    DS_VARS(X)
#undef X
} ds_vars;

static uint64_t inv_or_zero(uint64_t x) { return x % P ? orc_gl_inv(x) : 0; }

#define SET8(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; \
    dst.pfx##4 = (src)[4]; dst.pfx##5 = (src)[5]; dst.pfx##6 = (src)[6]; dst.pfx##7 = (src)[7]; } while (0)
#define SET12(dst, pfx, src) do { SET8(dst, pfx, src); dst.pfx##8 = (src)[8]; dst.pfx##9 = (src)[9]; dst.pfx##10 = (src)[10]; \
    dst.pfx##11 = (src)[11]; } while (0)
#define GET8(arr, src, pfx) do { (arr)[0] = src.pfx##0; (arr)[1] = src.pfx##1; (arr)[2] = src.pfx##2; (arr)[3] = src.pfx##3; \
    (arr)[4] = src.pfx##4; (arr)[5] = src.pfx##5; (arr)[6] = src.pfx##6; (arr)[7] = src.pfx##7; } while (0)
#define GET12(arr, src, pfx) do { GET8(arr, src, pfx); (arr)[8] = src.pfx##8; (arr)[9] = src.pfx##9; (arr)[10] = src.pfx##10; \
    (arr)[11] = src.pfx##11; } while (0)
#define BYTES(dst, pfx, x) do { uint32_t _x = (uint32_t)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; dst.pfx##_b3 = _x >> 24; } while (0)
/* x = [a == b] with its inverse witness */
#define ISZ(dst, w, z, a, b) do { uint64_t _d = orc_gl_sub((a) % P, (b) % P); dst.z = _d == 0; dst.w = inv_or_zero(_d); } while (0)

static void poseidon_cells(uint64_t *trace, size_t n_rows, size_t row, const uint64_t in[12], uint64_t out[12]) {
    uint64_t slots[130];
    orc_poseidon2_flattened(in, slots);
    for (int k = 0; k < 130; k++) CELL(k, row) = slots[k];
    memcpy(out, slots + 118, 96);
}

/* rq_tail_in / rq_len_in: state of the deduplicated queue before the block (NULL / 0 = empty); only read for the
   first instance. Returns 0 or a negative error. */
int orc_decommit_sorter_synthesize(const zkw_decommit_sorter_instance *inst, const zkw_decommit_query *sorted_q,
                                   const uint64_t *unsorted_enc, const uint64_t *sorted_enc,
                                   const uint64_t *challenges /* [2][9] */, const uint64_t *rq_tail_in, uint32_t rq_len_in,
                                   const uint64_t *public_input /* [4] or NULL */, uint32_t capacity, size_t n_rows,
                                   uint64_t *trace) {
    if (DS_MIN_ROWS(capacity) > n_rows) return -1;
    const size_t first = inst->first_item, m = inst->num_items;
    if (m == 0 || m > capacity) return -2;
    const zkw_decommit_sorter_fsm *fi = &inst->hidden_fsm_input;
    const int start = inst->start_flag != 0;
    const size_t rs = (size_t)DS_REGION_STRIDE(capacity), bnd = (size_t)DS_BOUNDARY_ROW(capacity);
    ds_vars prev, cur, glob;
    memset(&prev, 0, sizeof prev);
    memset(&glob, 0, sizeof glob);

    /* ---- registers at "cycle -1" */
    SET12(prev, uh, start ? inst->initial_queue_state.head : fi->initial_queue_state.head);
    SET12(prev, sh, start ? inst->sorted_queue_initial_state.head : fi->sorted_queue_state.head);
    prev.len_u = start ? inst->initial_queue_state.length : fi->initial_queue_state.length;
    prev.len_s = start ? inst->sorted_queue_initial_state.length : fi->sorted_queue_state.length;
    {
        const uint64_t zero12[12] = {0};
        SET12(prev, rh, start ? (rq_tail_in ? rq_tail_in : zero12) : fi->final_queue_state.tail);
        prev.len_r = start ? rq_len_in : fi->final_queue_state.length;
    }
    prev.lhs0 = start ? 1 : fi->lhs_accumulator[0]; prev.lhs1 = start ? 1 : fi->lhs_accumulator[1];
    prev.rhs0 = start ? 1 : fi->rhs_accumulator[0]; prev.rhs1 = start ? 1 : fi->rhs_accumulator[1];
    prev.ts = fi->previous_packed_key[0];
    prev.page = fi->previous_record.memory_page;
    prev.h0 = fi->previous_packed_key[1]; prev.h1 = fi->previous_packed_key[2]; prev.h2 = fi->previous_packed_key[3];
    prev.es3 = fi->previous_packed_key[4]; prev.es4 = fi->previous_packed_key[5]; prev.es5 = fi->previous_packed_key[6];
    prev.es6 = fi->previous_packed_key[7]; prev.es7 = fi->previous_packed_key[8];
    prev.gvalid = start ? 0 : 1;
    if (!start) { /* the open group's first request: (hash, page, first_encountered_timestamp, fresh) */
        zkw_decommit_query g = fi->previous_record;
        g.timestamp = fi->first_encountered_timestamp;
        g.is_fresh = 1;
        uint64_t e[8];
        orc_encode_decommit_queries(&g, 1, e);
        SET8(prev, ge, e);
    }
    glob.c0_1 = challenges[1]; glob.c0_2 = challenges[2]; glob.c0_3 = challenges[3]; glob.c0_4 = challenges[4];
    glob.c0_5 = challenges[5]; glob.c0_6 = challenges[6]; glob.c0_7 = challenges[7]; glob.c0_8 = challenges[8];
    glob.c1_1 = challenges[10]; glob.c1_2 = challenges[11]; glob.c1_3 = challenges[12]; glob.c1_4 = challenges[13];
    glob.c1_5 = challenges[14]; glob.c1_6 = challenges[15]; glob.c1_7 = challenges[16]; glob.c1_8 = challenges[17];

#define XC(col, v) CELL(col, row) = cur.v;
#define XP(col, v) CELL(col, row) = prev.v;
#define XG(col, v) CELL(col, row) = glob.v;
#define XX(col, v) CELL(col, row) = cur.v;
#define XSKIP(col, v)
    { /* BND_IN holds the cycle -1 registers and the challenges */
        const size_t row = bnd + DS_ROWOFF_BND_IN;
        cur = prev;
        DS_FILL_BND_IN(XC, XP, XG, XX)
    }

    for (size_t i = 0; i < capacity; i++) {
        const size_t idx = first + i;
        const int can_pop = i < m;
        memset(&cur, 0, sizeof cur);
        zkw_decommit_query q;
        memset(&q, 0, sizeof q);
        uint64_t eu[8] = {0}, es[8] = {0}, in[12], out[12];
        if (can_pop) { q = sorted_q[idx]; memcpy(eu, unsorted_enc + 8 * idx, 64); memcpy(es, sorted_enc + 8 * idx, 64); }
        cur.can_pop = can_pop;
        SET8(cur, eu, eu);
        SET8(cur, es, es);
        /* PU / PS / PR: the three permutations */
        memcpy(in, eu, 64); in[8] = prev.uh8; in[9] = prev.uh9; in[10] = prev.uh10; in[11] = prev.uh11;
        poseidon_cells(trace, n_rows, (size_t)DS_ROW_PU * rs + i, in, out);
        SET12(cur, uo, out);
        memcpy(in, es, 64); in[8] = prev.sh8; in[9] = prev.sh9; in[10] = prev.sh10; in[11] = prev.sh11;
        poseidon_cells(trace, n_rows, (size_t)DS_ROW_PS * rs + i, in, out);
        SET12(cur, so, out);
        GET8(in, prev, ge); in[8] = prev.rh8; in[9] = prev.rh9; in[10] = prev.rh10; in[11] = prev.rh11;
        poseidon_cells(trace, n_rows, (size_t)DS_ROW_PR * rs + i, in, out);
        SET12(cur, ro, out);
        /* the sorted request */
        cur.h0 = q.hash[0]; cur.h1 = q.hash[1]; cur.h2 = q.hash[2];
        cur.page = q.memory_page; cur.ts = q.timestamp; cur.fresh = q.is_fresh ? 1 : 0;
        BYTES(cur, h0, q.hash[0]); BYTES(cur, h1, q.hash[1]); BYTES(cur, h2, q.hash[2]);
        BYTES(cur, h3, q.hash[3]); BYTES(cur, h4, q.hash[4]); BYTES(cur, h5, q.hash[5]); BYTES(cur, h6, q.hash[6]);
        BYTES(cur, h7, q.hash[7]); BYTES(cur, page, q.memory_page); BYTES(cur, ts, q.timestamp);
        /* grand products */
        for (int r = 0; r < 2; r++) {
            const uint64_t *ch = challenges + 9 * r;
            uint64_t lc = orc_gl_add(ch[8], eu[0] % P), rc = orc_gl_add(ch[8], es[0] % P);
            for (int k = 1; k < 8; k++) {
                lc = orc_gl_add(lc, orc_gl_mul(eu[k] % P, ch[k]));
                rc = orc_gl_add(rc, orc_gl_mul(es[k] % P, ch[k]));
            }
            const uint64_t pl = r ? prev.lhs1 : prev.lhs0, pr = r ? prev.rhs1 : prev.rhs0;
            const uint64_t nl = orc_gl_mul(pl, lc), nr = orc_gl_mul(pr, rc);
            if (r == 0) { cur.lc0 = lc; cur.rc0 = rc; cur.nl0 = nl; cur.nr0 = nr; cur.lhs0 = can_pop ? nl : pl; cur.rhs0 = can_pop ? nr : pr; }
            else { cur.lc1 = lc; cur.rc1 = rc; cur.nl1 = nl; cur.nr1 = nr; cur.lhs1 = can_pop ? nl : pl; cur.rhs1 = can_pop ? nr : pr; }
        }
        /* key - previous key, nine u32 limbs from the least significant: ts, h0..h7 */
        {
            const uint64_t c9[9] = {q.timestamp, q.hash[0], q.hash[1], q.hash[2], q.hash[3], q.hash[4], q.hash[5], q.hash[6], q.hash[7]};
            const uint64_t p9[9] = {prev.ts, prev.h0, prev.h1, prev.h2, prev.es3, prev.es4, prev.es5, prev.es6, prev.es7};
            uint64_t d[9], bw[9], borrow = 0;
            for (int k = 0; k < 9; k++) {
                const int64_t t = (int64_t)c9[k] - (int64_t)p9[k] - (int64_t)borrow;
                bw[k] = t < 0;
                d[k] = (uint64_t)(t + (bw[k] ? (1ll << 32) : 0));
                borrow = bw[k];
            }
            cur.d0 = d[0]; cur.d1 = d[1]; cur.d2 = d[2]; cur.d3 = d[3]; cur.d4 = d[4]; cur.d5 = d[5]; cur.d6 = d[6]; cur.d7 = d[7]; cur.d8 = d[8];
            cur.bw0 = bw[0]; cur.bw1 = bw[1]; cur.bw2 = bw[2]; cur.bw3 = bw[3]; cur.bw4 = bw[4]; cur.bw5 = bw[5]; cur.bw6 = bw[6];
            cur.bw7 = bw[7]; cur.bw8 = bw[8];
            BYTES(cur, d0, d[0]); BYTES(cur, d1, d[1]); BYTES(cur, d2, d[2]); BYTES(cur, d3, d[3]); BYTES(cur, d4, d[4]);
            BYTES(cur, d5, d[5]); BYTES(cur, d6, d[6]); BYTES(cur, d7, d[7]); BYTES(cur, d8, d[8]);
            if (can_pop && prev.gvalid && bw[8]) return -3; /* the sorted queue is not sorted */
        }
        ISZ(cur, w_e0, z_e0, cur.h0, prev.h0); ISZ(cur, w_e1, z_e1, cur.h1, prev.h1); ISZ(cur, w_e2, z_e2, cur.h2, prev.h2);
        ISZ(cur, w_e3, z_e3, cur.es3, prev.es3); ISZ(cur, w_e4, z_e4, cur.es4, prev.es4); ISZ(cur, w_e5, z_e5, cur.es5, prev.es5);
        ISZ(cur, w_e6, z_e6, cur.es6, prev.es6); ISZ(cur, w_e7, z_e7, cur.es7, prev.es7);
        cur.same_a = cur.z_e0 & cur.z_e1 & cur.z_e2 & cur.z_e3;
        cur.same_hash = cur.same_a & cur.z_e4 & cur.z_e5 & cur.z_e6 & cur.z_e7;
        cur.new_group = can_pop && !(cur.same_hash && prev.gvalid);
        cur.push = cur.new_group && prev.gvalid;
        cur.gvalid = prev.gvalid | (uint64_t)can_pop;
        if (can_pop && cur.fresh != cur.new_group) return -4; /* is_fresh must mark exactly the first request of a hash */
        if (can_pop && cur.same_hash && prev.gvalid && cur.page != prev.page) return -5;
        /* queues and group registers */
        ISZ(cur, w_lu, z_lu, prev.len_u, 0); ISZ(cur, w_ls, z_ls, prev.len_s, 0);
        cur.len_u = prev.len_u - can_pop; cur.len_s = prev.len_s - can_pop; cur.len_r = prev.len_r + cur.push;
        {
            uint64_t a[12], b[12], o[12];
            GET12(a, cur, uo); GET12(b, prev, uh);
            for (int k = 0; k < 12; k++) o[k] = can_pop ? a[k] : b[k];
            SET12(cur, uh, o);
            GET12(a, cur, so); GET12(b, prev, sh);
            for (int k = 0; k < 12; k++) o[k] = can_pop ? a[k] : b[k];
            SET12(cur, sh, o);
            GET12(a, cur, ro); GET12(b, prev, rh);
            for (int k = 0; k < 12; k++) o[k] = cur.push ? a[k] : b[k];
            SET12(cur, rh, o);
            GET8(b, prev, ge);
            for (int k = 0; k < 8; k++) o[k] = cur.new_group ? es[k] : b[k];
            SET8(cur, ge, o);
        }
        /* scatter (the Poseidon rows' 130 gate cells are already in place: only their lookup cells remain) */
#define XC_L(col, v) if ((col) >= DS_G) CELL(col, row) = cur.v;
        { const size_t row = (size_t)DS_ROW_PU * rs + i; DS_FILL_PU(XC_L, XSKIP, XSKIP, XSKIP) }
        { const size_t row = (size_t)DS_ROW_PS * rs + i; DS_FILL_PS(XC_L, XSKIP, XSKIP, XSKIP) }
        { const size_t row = (size_t)DS_ROW_PR * rs + i; DS_FILL_PR(XC_L, XSKIP, XSKIP, XSKIP) }
        { const size_t row = (size_t)DS_ROW_A * rs + i; DS_FILL_A(XC, XP, XG, XX) }
        { const size_t row = (size_t)DS_ROW_B * rs + i; DS_FILL_B(XC, XP, XG, XX) }
        { const size_t row = (size_t)DS_ROW_C * rs + i; DS_FILL_C(XC, XP, XG, XX) }
        { const size_t row = (size_t)DS_ROW_D * rs + i; DS_FILL_D(XC, XP, XG, XX) }
        prev = cur;
    }

    /* ---- BND_OUT: the registers after the last cycle, the queue tails, completion checks and the flush */
    {
        const size_t row = bnd + DS_ROWOFF_BND_OUT;
        cur = prev;
        SET12(cur, tail_u, inst->initial_queue_state.tail);
        SET12(cur, tail_s, inst->sorted_queue_initial_state.tail);
        cur.completion = inst->completion_flag ? 1 : 0;
        ISZ(cur, w_end, z_end, cur.len_u, 0);
        cur.flush = cur.completion & cur.gvalid;
        uint64_t in[12], out[12], rh[12], o[12];
        GET8(in, cur, ge); in[8] = cur.rh8; in[9] = cur.rh9; in[10] = cur.rh10; in[11] = cur.rh11;
        poseidon_cells(trace, n_rows, bnd + DS_ROWOFF_PF, in, out);
        SET12(cur, fo, out);
        GET12(rh, cur, rh);
        for (int k = 0; k < 12; k++) o[k] = cur.flush ? out[k] : rh[k];
        SET12(cur, final_rh, o);
        cur.final_len_r = cur.len_r + cur.flush;
        DS_FILL_BND_OUT(XC, XP, XG, XX)
        if (cur.completion && !cur.z_end) return -6;
    }

    (void)public_input; /* the PI row is derived by the closed-form section (orc_ds_fill_closed_form, closed_form_fill.c), which runs next */

    /* multiplicities of the 8-bit range-check table: every cell of the lookup columns, padding included */
    for (int t = 0; t < 256; t++) CELL(DS_MULT_COL, t) = 0;
    for (int c = DS_G; c < DS_G + DS_L; c++)
        for (size_t r = 0; r < n_rows; r++) {
            uint64_t v = CELL(c, r);
            if (v > 255) return -9;
            CELL(DS_MULT_COL, v) += 1;
        }
    return 0;
}
