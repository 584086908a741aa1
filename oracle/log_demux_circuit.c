/* log_demux_circuit.c — TEST INFRASTRUCTURE: CPU restatement of LogDemuxer synthesis ("zkw trace v2",
 * include/zkw_log_demux_circuit_spec.h) — the counterpart of ZkSyncBaseLayerCircuit::synthesis for that instance type
 * (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:286-323, wrapper base_layer/log_demux.rs:27-38; witness
 * src/witness/individual_circuits/log_demux.rs:20-388). Sequential: the registers (input queue head, six output queue
 * tails and lengths) are carried cycle by cycle; the route of a record is re-derived from the bytes of its ENCODING
 * (not from the routed queues the builder produced). Cells are scattered through the generated LD_FILL_<row> lists.
 * The satisfiability check (circuit_check.c) shares no code with it. */
#include "oracle.h"
#include "../include/zkw_log_demux_circuit_spec.h"
#include <stdlib.h>
#include <string.h>

#define P ZKW_GOLDILOCKS_P
#define CELL(col, row) trace[(size_t)(col) * n_rows + (row)]

typedef struct {
#define X(n) uint64_t n;
    LD_VARS(X)
#undef X
} ld_vars;

static uint64_t inv_or_zero(uint64_t x) { return x % P ? orc_gl_inv(x) : 0; }

#define SET4(dst, pfx, src) do { dst.pfx##0 = (src)[0]; dst.pfx##1 = (src)[1]; dst.pfx##2 = (src)[2]; dst.pfx##3 = (src)[3]; } while (0)
#define GET4(arr, src, pfx) do { (arr)[0] = src.pfx##0; (arr)[1] = src.pfx##1; (arr)[2] = src.pfx##2; (arr)[3] = src.pfx##3; } while (0)
#define BYTES4(dst, pfx, x) do { uint32_t _x = (uint32_t)(x); dst.pfx##_b0 = _x & 0xFF; dst.pfx##_b1 = (_x >> 8) & 0xFF; \
    dst.pfx##_b2 = (_x >> 16) & 0xFF; dst.pfx##_b3 = _x >> 24; } while (0)
#define IS_ZERO(x, w, z) do { const uint64_t _d = (x) % P; cur.z = _d == 0; cur.w = inv_or_zero(_d); } while (0)

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

int orc_log_demux_synthesize(const zkw_log_demux_instance *inst, const uint64_t *in_enc, const uint64_t *public_input /* [4] or NULL */, uint32_t capacity,
        size_t n_rows, uint64_t *trace) {
    if (LD_MIN_ROWS(capacity) > n_rows) return -1;
    const size_t first = inst->first_item, m = inst->num_items;
    if (m > capacity) return -2;
    const zkw_log_demux_fsm *fi = &inst->hidden_fsm_input;
    const int start = inst->start_flag != 0;
    const size_t rs = (size_t)LD_REGION_STRIDE(capacity), bnd = (size_t)LD_BOUNDARY_ROW(capacity);
    ld_vars prev, cur, glob;
    memset(&prev, 0, sizeof prev);
    memset(&glob, 0, sizeof glob);

    SET4(prev, ih, start ? inst->initial_log_queue_state.head : fi->initial_log_queue_state.head);
    prev.len_i = start ? inst->initial_log_queue_state.length : fi->initial_log_queue_state.length;
    if (!start) { /* the output queues start empty (log_demux.rs:110-168) */
        SET4(prev, qt_st, fi->queue_state[ZKW_DEMUX_STORAGE].tail);      prev.ql_st = fi->queue_state[ZKW_DEMUX_STORAGE].length;
        SET4(prev, qt_ev, fi->queue_state[ZKW_DEMUX_EVENTS].tail);       prev.ql_ev = fi->queue_state[ZKW_DEMUX_EVENTS].length;
        SET4(prev, qt_l1, fi->queue_state[ZKW_DEMUX_L1_MESSAGES].tail);  prev.ql_l1 = fi->queue_state[ZKW_DEMUX_L1_MESSAGES].length;
        SET4(prev, qt_kc, fi->queue_state[ZKW_DEMUX_KECCAK256].tail);    prev.ql_kc = fi->queue_state[ZKW_DEMUX_KECCAK256].length;
        SET4(prev, qt_sh, fi->queue_state[ZKW_DEMUX_SHA256].tail);       prev.ql_sh = fi->queue_state[ZKW_DEMUX_SHA256].length;
        SET4(prev, qt_ec, fi->queue_state[ZKW_DEMUX_ECRECOVER].tail);    prev.ql_ec = fi->queue_state[ZKW_DEMUX_ECRECOVER].length;
    }

#define XC(col, v) CELL(col, row) = cur.v;
#define XP(col, v) CELL(col, row) = prev.v;
#define XG(col, v) CELL(col, row) = glob.v;
    {
        const size_t row = bnd + LD_ROWOFF_BND_IN;
        cur = prev;
        LD_FILL_BND_IN(XC, XP, XG, XC)
    }

    for (size_t i = 0; i < capacity; i++) {
        const size_t idx = first + i;
        const int can_pop = i < m;
        if (can_pop != (prev.len_i != 0)) return -3;
        memset(&cur, 0, sizeof cur);
        uint64_t es[20] = {0}, old[4], o4[4];
        if (can_pop) memcpy(es, in_enc + 20 * idx, 160);
        cur.can_pop = can_pop;
        cur.one = 1;
        /* the byte split of words 10..17 (log_query.rs:118-196) */
        uint8_t ab[20];
        BYTES4(cur, w10, es[10]); BYTES4(cur, w11, es[11]); BYTES4(cur, w12, es[12]); BYTES4(cur, w13, es[13]);
        BYTES4(cur, w14, es[14]); BYTES4(cur, w15, es[15]); BYTES4(cur, w16, es[16]); BYTES4(cur, w17, es[17]);
        cur.kb30 = (es[10] >> 32) & 0xFF; cur.kb31 = (es[10] >> 40) & 0xFF;
        ab[0] = (uint8_t)(es[10] >> 48);
        for (int k = 11; k <= 16; k++)
            for (int j = 0; j < 3; j++) ab[1 + 3 * (k - 11) + j] = (uint8_t)(es[k] >> (32 + 8 * j));
        ab[19] = (uint8_t)(es[17] >> 32);
        cur.aux = (es[17] >> 40) & 0xFF; cur.shard = (es[17] >> 48) & 0xFF;
        for (int k = 10; k < 18; k++)
            if (es[k] >> 56) return -4; /* not an encoding */
        cur.a0 = ab[0]; cur.a1 = ab[1]; cur.a2 = ab[2]; cur.a3 = ab[3]; cur.a4 = ab[4]; cur.a5 = ab[5]; cur.a6 = ab[6]; cur.a7 = ab[7];
        cur.a8 = ab[8]; cur.a9 = ab[9]; cur.a10 = ab[10]; cur.a11 = ab[11]; cur.a12 = ab[12]; cur.a13 = ab[13]; cur.a14 = ab[14];
        cur.a15 = ab[15]; cur.a16 = ab[16]; cur.a17 = ab[17]; cur.a18 = ab[18]; cur.a19 = ab[19];
        /* the route (log_demux.rs:171-251) */
        IS_ZERO(cur.aux, w_st, is_st);
        IS_ZERO(orc_gl_sub(cur.aux, 1), w_ev, is_ev);
        IS_ZERO(orc_gl_sub(cur.aux, 2), w_l1, is_l1);
        IS_ZERO(orc_gl_sub(cur.aux, 3), w_pre, is_pre);
        uint64_t hsum = 0;
        for (int k = 4; k < 20; k++) hsum += ab[k];
        IS_ZERO(hsum, w_hz, hz);
        const uint64_t limb0 = (uint64_t)ab[0] | (uint64_t)ab[1] << 8 | (uint64_t)ab[2] << 16 | (uint64_t)ab[3] << 24;
        IS_ZERO(orc_gl_sub(limb0, 0x8010), w_akc, eq_kc);
        IS_ZERO(orc_gl_sub(limb0, 0x02), w_ash, eq_sh);
        IS_ZERO(orc_gl_sub(limb0, 0x01), w_aec, eq_ec);
        if (can_pop) {
            if (!(cur.is_st | cur.is_ev | cur.is_l1 | cur.is_pre)) return -5;
            if (cur.is_st && cur.shard) return -5;
            if (cur.is_pre && (es[19] % P)) return -5;
        }
        cur.r_st = can_pop & cur.is_st; cur.r_ev = can_pop & cur.is_ev; cur.r_l1 = can_pop & cur.is_l1;
        cur.pre_hz = can_pop & cur.is_pre & cur.hz;
        cur.r_kc = cur.pre_hz & cur.eq_kc; cur.r_sh = cur.pre_hz & cur.eq_sh; cur.r_ec = cur.pre_hz & cur.eq_ec;
        /* pop */
        GET4(old, prev, ih);
        queue_op(trace, n_rows, (size_t)LD_ROW_I1 * rs + i, (size_t)LD_ROW_I2 * rs + i, (size_t)LD_ROW_I3 * rs + i, es, old, o4);
        SET4(cur, i3o, o4);
        for (int k = 0; k < 4; k++) o4[k] = can_pop ? o4[k] : old[k];
        SET4(cur, ih, o4);
        IS_ZERO(prev.len_i, w_li, z_li);
        cur.len_i = prev.len_i - can_pop;
        /* the one conditional push */
        uint64_t sel[4] = {0}, t[4];
#define PICK(q) if (cur.r_##q) GET4(sel, prev, qt_##q);
        PICK(st) PICK(ev) PICK(l1) PICK(kc) PICK(sh) PICK(ec)
#undef PICK
        SET4(cur, sel, sel);
        queue_op(trace, n_rows, (size_t)LD_ROW_P1 * rs + i, (size_t)LD_ROW_P2 * rs + i, (size_t)LD_ROW_P3 * rs + i, es, sel, o4);
        SET4(cur, p3o, o4);
#define KEEP(q) GET4(t, prev, qt_##q); for (int k = 0; k < 4; k++) t[k] = cur.r_##q ? o4[k] : t[k]; SET4(cur, qt_##q, t); \
        cur.ql_##q = prev.ql_##q + cur.r_##q;
        KEEP(st) KEEP(ev) KEEP(l1) KEEP(kc) KEEP(sh) KEEP(ec)
#undef KEEP
        /* es0..19 are named variables of the general rows too */
        cur.es10 = es[10]; cur.es11 = es[11]; cur.es12 = es[12]; cur.es13 = es[13]; cur.es14 = es[14]; cur.es15 = es[15];
        cur.es16 = es[16]; cur.es17 = es[17]; cur.es19 = es[19];
#define ROWAT(R) const size_t row = (size_t)(R) * rs + i;
        { ROWAT(LD_ROW_X0) LD_FILL_X0(XC, XP, XG, XC) } { ROWAT(LD_ROW_X1) LD_FILL_X1(XC, XP, XG, XC) }
        { ROWAT(LD_ROW_X2) LD_FILL_X2(XC, XP, XG, XC) } { ROWAT(LD_ROW_X3) LD_FILL_X3(XC, XP, XG, XC) }
        { ROWAT(LD_ROW_R) LD_FILL_R(XC, XP, XG, XC) }
        { ROWAT(LD_ROW_Q) LD_FILL_Q(XC, XP, XG, XC) }
        prev = cur;
    }

    {
        const size_t row = bnd + LD_ROWOFF_BND_OUT;
        cur = prev;
        SET4(cur, tail_i, inst->initial_log_queue_state.tail);
        cur.completion = inst->completion_flag ? 1 : 0;
        IS_ZERO(cur.len_i, w_end, z_end);
        LD_FILL_BND_OUT(XC, XP, XG, XC)
        if (cur.completion && !cur.z_end) return -7;
    }

    (void)public_input; /* the PI row is derived by the closed-form section (orc_ld_fill_closed_form, closed_form_fill.c), which runs next */
    for (int t = 0; t < 256; t++) CELL(LD_MULT_COL, t) = 0;
    for (int c = LD_G; c < LD_G + LD_L; c++)
        for (size_t r = 0; r < n_rows; r++) {
            uint64_t v = CELL(c, r);
            if (v > 255) return -9;
            CELL(LD_MULT_COL, v) += 1;
        }
    return 0;
}
