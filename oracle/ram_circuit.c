/* ram_circuit.c — TEST INFRASTRUCTURE: CPU restatement of RAMPermutation synthesis ("zkw trace v1",
 * include/zkw_ram_circuit_spec.h) — the counterpart of ZkSyncBaseLayerCircuit::synthesis for the
 * RAMPermutation instance (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:286-323 with the
 * wrapper base_layer/ram_permutation.rs:26-135) — and of the satisfiability check the reference's tests
 * run on every emitted circuit (src/tests/mod.rs:130-259, `check_if_satisfied`).
 *
 * The fill walks the cycles sequentially carrying the registers, the way the circuit body is written;
 * the product's kernels compute every cycle independently from the block-wide arrays. The checker is a
 * generic interpreter of the spec tables and shares no code with the fill.
 */
#include "oracle.h"
#include "../include/zkw_ram_circuit_spec.h"
#include "../include/zkw_poseidon2_params.h"
#include <stdlib.h>
#include <string.h>

#define P ZKW_GOLDILOCKS_P
typedef unsigned __int128 u128;

/* (the satisfiability check lives in circuit_check.c) */

/* ---- Poseidon2 with every flattened-gate variable written out: 12 inputs, the state after each of
   the first 4 full rounds, the S-box output of element 0 in each partial round, the state after each of
   the last 4 full rounds (the last one is the output). 130 values. */
static uint64_t sbox7(uint64_t x) {
    uint64_t x2 = orc_gl_mul(x, x), x3 = orc_gl_mul(x2, x), x4 = orc_gl_mul(x2, x2);
    return orc_gl_mul(x3, x4);
}
static void ext_layer(uint64_t s[12]) {
    static const uint64_t M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
    uint64_t t[12];
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 4; i++) {
            u128 acc = 0;
            for (int j = 0; j < 4; j++) acc += (u128)M4[i][j] * s[4 * c + j];
            t[4 * c + i] = orc_gl_reduce128(acc);
        }
    for (int i = 0; i < 4; i++) {
        uint64_t col = orc_gl_add(orc_gl_add(t[i], t[4 + i]), t[8 + i]);
        for (int c = 0; c < 3; c++) s[4 * c + i] = orc_gl_add(t[4 * c + i], col);
    }
}
void orc_poseidon2_flattened(const uint64_t in[12], uint64_t slots[130]) {
    uint64_t s[12];
    memcpy(s, in, 96);
    memcpy(slots, in, 96);
    int pos = 12, r = 0;
    ext_layer(s);
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(orc_gl_add(s[i], P2_ROUND_CONSTANTS[12 * r + i]));
        ext_layer(s);
        memcpy(slots + pos, s, 96);
        pos += 12;
    }
    for (int k = 0; k < 22; k++, r++) {
        s[0] = sbox7(orc_gl_add(s[0], P2_ROUND_CONSTANTS[12 * r]));
        slots[pos++] = s[0];
        uint64_t sum = 0;
        for (int i = 0; i < 12; i++) sum = orc_gl_add(sum, s[i]);
        for (int i = 0; i < 12; i++) s[i] = orc_gl_add(orc_gl_mul(s[i], 1ULL << P2_INTERNAL_DIAG_SHIFTS[i]), sum);
    }
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(orc_gl_add(s[i], P2_ROUND_CONSTANTS[12 * r + i]));
        ext_layer(s);
        memcpy(slots + pos, s, 96);
        pos += 12;
    }
}

/* column-major trace: cell(col, row) = t[col * n_rows + row] */
#define CELL(col, row) trace[(size_t)(col) * n_rows + (row)]
#define ROWOF(region, cyc) ((size_t)(region) * RC_REGION_STRIDE(capacity) + (cyc))

static uint64_t inv_or_zero(uint64_t x) { return x % P ? orc_gl_inv(x) : 0; }

static void put_bytes(uint64_t *trace, size_t n_rows, size_t row, int col0, uint32_t x) {
    for (int k = 0; k < 4; k++) CELL(col0 + k, row) = (x >> (8 * k)) & 0xFF;
}

/* Fill one instance. Inputs are the block-wide arrays of orc_ram_build_instances and the instance
   record. trace: RC_COLS * n_rows, zero-initialised by the caller. Returns 0 or a negative error. */
int orc_ram_synthesize(const zkw_ram_instance *inst, const zkw_mem_query *sorted_q, const uint64_t *unsorted_enc,
                       const uint64_t *sorted_enc, const uint64_t *unsorted_tails, const uint64_t *sorted_tails,
                       const uint64_t *challenges /* [2][9] */, const uint64_t *lhs_z, const uint64_t *rhs_z,
                       size_t n_total, uint32_t capacity, size_t n_rows, uint64_t *trace) {
    if (RC_MIN_ROWS(capacity) > n_rows) return -1;
    const size_t first = inst->first_item, m = inst->num_items;
    if (m == 0 || m > capacity) return -2;
    const zkw_ram_fsm *fi = &inst->hidden_fsm_input;
    const int start = inst->start_flag != 0;

    /* registers at "cycle -1" (start_flag selects the observable input, ram_permutation.rs:373-384) */
    uint64_t uh[12], sh[12], lhs[2], rhs[2], len_u, len_s, cnt;
    uint32_t p_ts, p_idx, p_page, p_ptr;
    uint64_t p_val[5]; /* es3..es6, v4 of the previous item */
    memcpy(uh, start ? inst->unsorted_queue_initial_state.head : fi->current_unsorted_queue_state.head, 96);
    memcpy(sh, start ? inst->sorted_queue_initial_state.head : fi->current_sorted_queue_state.head, 96);
    len_u = start ? inst->unsorted_queue_initial_state.length : fi->current_unsorted_queue_state.length;
    len_s = start ? inst->sorted_queue_initial_state.length : fi->current_sorted_queue_state.length;
    for (int r = 0; r < 2; r++) { lhs[r] = fi->lhs_accumulator[r]; rhs[r] = fi->rhs_accumulator[r]; }
    p_ts = fi->previous_sorting_key[0]; p_idx = fi->previous_sorting_key[1]; p_page = fi->previous_sorting_key[2];
    p_ptr = fi->previous_is_ptr;
    {
        zkw_mem_query pq;
        memset(&pq, 0, sizeof pq);
        memcpy(pq.value, fi->previous_value, 32);
        uint64_t e[8];
        orc_encode_memory_query(&pq, e);
        for (int k = 0; k < 5; k++) p_val[k] = e[3 + k];
    }
    cnt = fi->num_nondeterministic_writes;

    const size_t bin = ROWOF(RC_ROWS_PER_CYCLE, 0) + RC_ROWOFF_BND_IN, bout = ROWOF(RC_ROWS_PER_CYCLE, 0) + RC_ROWOFF_BND_OUT;
    for (int k = 0; k < 12; k++) { CELL(RC_BND_IN_uh0 + k, bin) = uh[k]; CELL(RC_BND_IN_sh0 + k, bin) = sh[k]; }
    CELL(RC_BND_IN_len_u, bin) = len_u; CELL(RC_BND_IN_len_s, bin) = len_s;
    CELL(RC_BND_IN_lhs0, bin) = lhs[0]; CELL(RC_BND_IN_lhs1, bin) = lhs[1];
    CELL(RC_BND_IN_rhs0, bin) = rhs[0]; CELL(RC_BND_IN_rhs1, bin) = rhs[1];
    CELL(RC_BND_IN_ts, bin) = p_ts; CELL(RC_BND_IN_idx, bin) = p_idx; CELL(RC_BND_IN_page, bin) = p_page;
    CELL(RC_BND_IN_es3, bin) = p_val[0]; CELL(RC_BND_IN_es4, bin) = p_val[1]; CELL(RC_BND_IN_es5, bin) = p_val[2];
    CELL(RC_BND_IN_es6, bin) = p_val[3]; CELL(RC_BND_IN_v4, bin) = p_val[4];
    CELL(RC_BND_IN_ptr, bin) = p_ptr; CELL(RC_BND_IN_cnt, bin) = cnt;
    for (int r = 0; r < 2; r++)
        for (int k = 1; k < 9; k++) CELL(RC_BND_IN_G_c0_1 + 8 * r + (k - 1), bin) = challenges[9 * r + k];

    for (size_t i = 0; i < capacity; i++) {
        const int can_pop = len_u > 0;
        if (can_pop != (i < m)) return -3; /* the instance record and the queue lengths disagree */
        zkw_mem_query q;
        uint64_t eu[8], es[8];
        memset(&q, 0, sizeof q);
        memset(eu, 0, sizeof eu);
        memset(es, 0, sizeof es);
        if (can_pop) {
            q = sorted_q[first + i];
            memcpy(eu, unsorted_enc + 8 * (first + i), 64);
            memcpy(es, sorted_enc + 8 * (first + i), 64);
        }
        const size_t rPU = ROWOF(RC_ROW_PU, i), rPS = ROWOF(RC_ROW_PS, i), rA = ROWOF(RC_ROW_A, i),
                     rB = ROWOF(RC_ROW_B, i), rC = ROWOF(RC_ROW_C, i), rD = ROWOF(RC_ROW_D, i);
        /* Poseidon2 rows */
        uint64_t in[12], slots[130], uo[12], so[12];
        memcpy(in, eu, 64); memcpy(in + 8, uh + 8, 32);
        orc_poseidon2_flattened(in, slots);
        for (int k = 0; k < 130; k++) CELL(k, rPU) = slots[k];
        memcpy(uo, slots + 118, 96);
        memcpy(in, es, 64); memcpy(in + 8, sh + 8, 32);
        orc_poseidon2_flattened(in, slots);
        for (int k = 0; k < 130; k++) CELL(k, rPS) = slots[k];
        memcpy(so, slots + 118, 96);
        if (can_pop && (memcmp(uo, unsorted_tails + 12 * (first + i), 96) || memcmp(so, sorted_tails + 12 * (first + i), 96)))
            return -4; /* queue witness inconsistent with the chain */
        CELL(RC_PU_idx, rPU) = q.index; CELL(RC_PU_v0, rPU) = q.value[0]; CELL(RC_PU_v1, rPU) = q.value[1];
        put_bytes(trace, n_rows, rPU, RC_PU_idx_b0, q.index);
        put_bytes(trace, n_rows, rPU, RC_PU_v0_b0, q.value[0]);
        put_bytes(trace, n_rows, rPU, RC_PU_v1_b0, q.value[1]);
        put_bytes(trace, n_rows, rPS, RC_PS_ts_b0, q.timestamp);
        put_bytes(trace, n_rows, rPS, RC_PS_page_b0, q.page);
        put_bytes(trace, n_rows, rPS, RC_PS_v4_b0, q.value[4]);

        /* row A */
        const uint64_t rw = q.rw_flag ? 1 : 0, ptr = q.value_is_pointer ? 1 : 0;
        CELL(RC_A_rw, rA) = rw; CELL(RC_A_ptr, rA) = ptr; CELL(RC_A_es2, rA) = es[2]; CELL(RC_A_idx, rA) = q.index;
        CELL(RC_A_v2, rA) = q.value[2]; CELL(RC_A_v3, rA) = q.value[3];
        put_bytes(trace, n_rows, rA, RC_A_v2_b0, q.value[2]);
        put_bytes(trace, n_rows, rA, RC_A_v3_b0, q.value[3]);
        put_bytes(trace, n_rows, rA, RC_A_v5_b0, q.value[5]);
        CELL(RC_A_es3, rA) = es[3]; CELL(RC_A_v0, rA) = q.value[0];
        CELL(RC_A_v5_b3c, rA) = (q.value[5] >> 24) & 0xFF;
        CELL(RC_A_can_pop, rA) = (uint64_t)can_pop;
        for (int k = 0; k < 8; k++) CELL(RC_A_eu0 + k, rA) = eu[k];
        CELL(RC_A_ts, rA) = es[0]; CELL(RC_A_page, rA) = es[1]; CELL(RC_A_es4, rA) = es[4];
        CELL(RC_A_es5, rA) = es[5]; CELL(RC_A_es6, rA) = es[6]; CELL(RC_A_v4, rA) = es[7];
        uint64_t new_lhs[2], new_rhs[2];
        for (int r = 0; r < 2; r++) {
            const uint64_t *ch = challenges + 9 * r;
            uint64_t lc = ch[8], rc = ch[8];
            for (int k = 0; k < 8; k++) {
                lc = orc_gl_add(lc, orc_gl_mul(eu[k], ch[k]));
                rc = orc_gl_add(rc, orc_gl_mul(es[k], ch[k]));
            }
            uint64_t nl = orc_gl_mul(lhs[r], lc), nr = orc_gl_mul(rhs[r], rc);
            new_lhs[r] = can_pop ? nl : lhs[r];
            new_rhs[r] = can_pop ? nr : rhs[r];
            const int o = r * (RC_A_lc1 - RC_A_lc0);
            for (int k = 1; k < 9; k++) CELL(RC_A_G_c0_1 + (k - 1) + r * (RC_A_G_c1_1 - RC_A_G_c0_1), rA) = ch[k];
            CELL(RC_A_lc0 + o, rA) = lc; CELL(RC_A_P_lhs0 + o, rA) = lhs[r]; CELL(RC_A_nl0 + o, rA) = nl;
            CELL(RC_A_lhs0 + o, rA) = new_lhs[r];
            CELL(RC_A_rc0 + o, rA) = rc; CELL(RC_A_P_rhs0 + o, rA) = rhs[r]; CELL(RC_A_nr0 + o, rA) = nr;
            CELL(RC_A_rhs0 + o, rA) = new_rhs[r];
            if (can_pop && (new_lhs[r] != lhs_z[(size_t)r * n_total + first + i] || new_rhs[r] != rhs_z[(size_t)r * n_total + first + i]))
                return -5;
        }

        /* row B */
        put_bytes(trace, n_rows, rB, RC_B_v6_b0, q.value[6]);
        put_bytes(trace, n_rows, rB, RC_B_v7_b0, q.value[7]);
        CELL(RC_B_es4, rB) = es[4]; CELL(RC_B_v1, rB) = q.value[1]; CELL(RC_B_v5_b3c, rB) = (q.value[5] >> 24) & 0xFF;
        CELL(RC_B_es5, rB) = es[5]; CELL(RC_B_v2, rB) = q.value[2];
        CELL(RC_B_es6, rB) = es[6]; CELL(RC_B_v3, rB) = q.value[3];
        const uint64_t bw0 = q.timestamp < p_ts;
        const uint32_t d0 = q.timestamp - p_ts; /* wraps by 2^32 exactly when bw0 */
        const uint64_t t1 = (uint64_t)p_idx + bw0, bw1 = (uint64_t)q.index < t1;
        const uint32_t d1 = (uint32_t)((uint64_t)q.index - t1);
        const uint64_t t2 = (uint64_t)p_page + bw1, bw2 = (uint64_t)q.page < t2;
        const uint32_t d2 = (uint32_t)((uint64_t)q.page - t2);
        CELL(RC_B_d0, rB) = d0; put_bytes(trace, n_rows, rB, RC_B_d0_b0, d0);
        CELL(RC_B_bw0, rB) = bw0; CELL(RC_B_ts, rB) = q.timestamp; CELL(RC_B_P_ts, rB) = p_ts;

        /* row C */
        CELL(RC_C_d1, rC) = d1; put_bytes(trace, n_rows, rC, RC_C_d1_b0, d1);
        CELL(RC_C_d2, rC) = d2; put_bytes(trace, n_rows, rC, RC_C_d2_b0, d2);
        CELL(RC_C_bw1, rC) = bw1; CELL(RC_C_bw2, rC) = bw2; CELL(RC_C_bw0, rC) = bw0;
        CELL(RC_C_idx, rC) = q.index; CELL(RC_C_P_idx, rC) = p_idx;
        CELL(RC_C_page, rC) = q.page; CELL(RC_C_P_page, rC) = p_page;
        CELL(RC_C_can_pop, rC) = (uint64_t)can_pop;
        if (can_pop && bw2) return -6; /* not sorted */
        const uint64_t x_idx = orc_gl_sub(q.index, p_idx), x_page = orc_gl_sub(q.page, p_page);
        const uint64_t z_idx = x_idx == 0, z_page = x_page == 0, same = z_idx & z_page;
        CELL(RC_C_w_idx, rC) = inv_or_zero(x_idx); CELL(RC_C_z_idx, rC) = z_idx;
        CELL(RC_C_w_page, rC) = inv_or_zero(x_page); CELL(RC_C_z_page, rC) = z_page; CELL(RC_C_same, rC) = same;
        const uint64_t val[5] = {es[3], es[4], es[5], es[6], es[7]};
        uint64_t zeq[5], zz[5];
        static const int C_VAL[5] = {RC_C_es3, RC_C_es4, RC_C_es5, RC_C_es6, RC_C_v4};
        static const int C_PVAL[5] = {RC_C_P_es3, RC_C_P_es4, RC_C_P_es5, RC_C_P_es6, RC_C_P_v4};
        static const int C_WEQ[5] = {RC_C_w_eq0, RC_C_w_eq1, RC_C_w_eq2, RC_C_w_eq3, RC_C_w_eq4};
        static const int C_ZEQ[5] = {RC_C_z_eq0, RC_C_z_eq1, RC_C_z_eq2, RC_C_z_eq3, RC_C_z_eq4};
        static const int C_WZ[5] = {RC_C_w_z0, RC_C_w_z1, RC_C_w_z2, RC_C_w_z3, RC_C_w_z4};
        static const int C_ZZ[5] = {RC_C_z_z0, RC_C_z_z1, RC_C_z_z2, RC_C_z_z3, RC_C_z_z4};
        for (int k = 0; k < 5; k++) {
            const uint64_t x = orc_gl_sub(val[k], p_val[k]);
            zeq[k] = x == 0; zz[k] = val[k] == 0;
            CELL(C_VAL[k], rC) = val[k]; CELL(C_PVAL[k], rC) = p_val[k];
            CELL(C_WEQ[k], rC) = inv_or_zero(x); CELL(C_ZEQ[k], rC) = zeq[k];
            CELL(C_WZ[k], rC) = inv_or_zero(val[k]); CELL(C_ZZ[k], rC) = zz[k];
        }
        const uint64_t peq = ptr == p_ptr, eq_a = zeq[0] & zeq[1] & zeq[2], value_equal = eq_a & zeq[3] & zeq[4] & peq;
        const uint64_t zz_a = zz[0] & zz[1] & zz[2], all_zero = zz_a & zz[3] & zz[4] & (1 - ptr);
        CELL(RC_C_ptr, rC) = ptr; CELL(RC_C_P_ptr, rC) = p_ptr; CELL(RC_C_peq, rC) = peq;
        CELL(RC_C_eq_a, rC) = eq_a; CELL(RC_C_value_equal, rC) = value_equal;
        CELL(RC_C_zz_a, rC) = zz_a; CELL(RC_C_all_zero, rC) = all_zero; CELL(RC_C_rw, rC) = rw;
        if (can_pop && !rw && same && !value_equal) return -7; /* read does not return the last write */
        if (can_pop && !rw && !same && !all_zero) return -8;   /* first read of a cell must be zero */
        const uint64_t z_ts = q.timestamp == 0, x_heap = orc_gl_sub(q.page, RC_HEAP_PAGE), z_heap = x_heap == 0;
        const uint64_t nd = (uint64_t)can_pop & z_ts & z_heap & rw & (1 - ptr);
        CELL(RC_C_ts, rC) = q.timestamp; CELL(RC_C_w_ts, rC) = inv_or_zero(q.timestamp); CELL(RC_C_z_ts, rC) = z_ts;
        CELL(RC_C_w_heap, rC) = inv_or_zero(x_heap); CELL(RC_C_z_heap, rC) = z_heap; CELL(RC_C_nd, rC) = nd;
        CELL(RC_C_P_cnt, rC) = cnt; CELL(RC_C_cnt, rC) = cnt + nd;

        /* row D */
        CELL(RC_D_P_len_u, rD) = len_u; CELL(RC_D_w_lu, rD) = inv_or_zero(len_u); CELL(RC_D_z_lu, rD) = len_u == 0;
        CELL(RC_D_P_len_s, rD) = len_s; CELL(RC_D_w_ls, rD) = inv_or_zero(len_s); CELL(RC_D_z_ls, rD) = len_s == 0;
        CELL(RC_D_can_pop, rD) = (uint64_t)can_pop;
        CELL(RC_D_len_u, rD) = len_u - can_pop; CELL(RC_D_len_s, rD) = len_s - can_pop;
        for (int k = 0; k < 12; k++) {
            const int o = 3 * k; /* (uo_k, p.uh_k, uh_k) triples in slot order */
            CELL(RC_D_uo0 + o, rD) = uo[k]; CELL(RC_D_P_uh0 + o, rD) = uh[k]; CELL(RC_D_uh0 + o, rD) = can_pop ? uo[k] : uh[k];
            CELL(RC_D_so0 + o, rD) = so[k]; CELL(RC_D_P_sh0 + o, rD) = sh[k]; CELL(RC_D_sh0 + o, rD) = can_pop ? so[k] : sh[k];
        }

        /* registers for the next cycle */
        if (can_pop) { memcpy(uh, uo, 96); memcpy(sh, so, 96); len_u--; len_s--; }
        for (int r = 0; r < 2; r++) { lhs[r] = new_lhs[r]; rhs[r] = new_rhs[r]; }
        p_ts = q.timestamp; p_idx = q.index; p_page = q.page; p_ptr = (uint32_t)ptr;
        memcpy(p_val, val, sizeof p_val);
        cnt += nd;
    }

    /* BND_OUT: the registers after the last cycle + the queue tails + completion checks */
    for (int k = 0; k < 12; k++) { CELL(RC_BND_OUT_uh0 + k, bout) = uh[k]; CELL(RC_BND_OUT_sh0 + k, bout) = sh[k]; }
    CELL(RC_BND_OUT_len_u, bout) = len_u; CELL(RC_BND_OUT_len_s, bout) = len_s;
    CELL(RC_BND_OUT_lhs0, bout) = lhs[0]; CELL(RC_BND_OUT_lhs1, bout) = lhs[1];
    CELL(RC_BND_OUT_rhs0, bout) = rhs[0]; CELL(RC_BND_OUT_rhs1, bout) = rhs[1];
    CELL(RC_BND_OUT_ts, bout) = p_ts; CELL(RC_BND_OUT_idx, bout) = p_idx; CELL(RC_BND_OUT_page, bout) = p_page;
    CELL(RC_BND_OUT_es3, bout) = p_val[0]; CELL(RC_BND_OUT_es4, bout) = p_val[1]; CELL(RC_BND_OUT_es5, bout) = p_val[2];
    CELL(RC_BND_OUT_es6, bout) = p_val[3]; CELL(RC_BND_OUT_v4, bout) = p_val[4];
    CELL(RC_BND_OUT_ptr, bout) = p_ptr; CELL(RC_BND_OUT_cnt, bout) = cnt;
    for (int k = 0; k < 12; k++) {
        CELL(RC_BND_OUT_tail_u0 + k, bout) = inst->unsorted_queue_initial_state.tail[k];
        CELL(RC_BND_OUT_tail_s0 + k, bout) = inst->sorted_queue_initial_state.tail[k];
    }
    CELL(RC_BND_OUT_completion, bout) = inst->completion_flag ? 1 : 0;
    CELL(RC_BND_OUT_w_end, bout) = inv_or_zero(len_u); CELL(RC_BND_OUT_z_end, bout) = len_u == 0;

    /* multiplicities of the 8-bit range-check table: every cell of the lookup columns, padding included */
    for (int t = 0; t < 256; t++) CELL(RC_MULT_COL, t) = 0;
    for (int c = RC_G; c < RC_G + RC_L; c++)
        for (size_t r = 0; r < n_rows; r++) {
            uint64_t v = CELL(c, r);
            if (v > 255) return -9;
            CELL(RC_MULT_COL, v) += 1;
        }
    return 0;
}
