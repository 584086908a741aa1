/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * The CLOSED-FORM SECTION of the netlist circuits (format and reference citations: include/zkw_netlist_closed_form.h): the flat
 * encodings of observable input / output and hidden FSM input / output as cells, their commitments, the compact form and the public
 * input as flattened Poseidon2 rows (ClosedFormInputCompactForm::from_full_form, src/witness/utils.rs:269-306; the sponge restated in
 * public_input.c), and the ties between those words and the registers of the trace (queue states of the queue section, the hash state
 * of the netlist's boundary rows) under the start / completion flags. Sequential and obvious: one cell after the other.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "../include/zkw_netlist_closed_form.h"

#define P ZKW_GOLDILOCKS_P
#define TR(c, r) trace[(size_t)(c) * n_rows + (size_t)(r)]

const nl_spec *orc_nl_spec(int circuit_type);
int orc_cf_encode(int circuit_type, const void *instances, size_t i, uint64_t *w, size_t n[4], int flags[2]);

typedef struct { const nl_spec *sp; const nlcf_desc *d; const nlq_desc *qd; uint32_t cycles, G; size_t n_rows, c0; } geom;

static int geom_of(int circuit_type, uint32_t cycles, size_t n_rows, geom *g) {
    g->sp = orc_nl_spec(circuit_type);
    g->d = nlcf_desc_of(circuit_type);
    g->qd = nlq_desc_of(circuit_type);
    if (!g->sp || !g->d || cycles == 0) return -1;
    g->cycles = cycles; g->G = g->sp->g; g->n_rows = n_rows;
    g->c0 = nlcf_first_row(circuit_type, g->sp, cycles);
    return nlcf_used_rows(circuit_type, g->sp, cycles) <= n_rows ? 0 : -1;
}
/* cell k of the header block / cell v of P2 block `perm` */
static uint64_t *hcell(const geom *g, uint64_t *trace, uint32_t k) { return &trace[(size_t)(k % g->G) * g->n_rows + g->c0 + k / g->G]; }
static size_t hrow(const geom *g, uint32_t k) { return g->c0 + k / g->G; }
static uint64_t *pcell(const geom *g, uint64_t *trace, uint32_t perm, uint32_t v) {
    return &trace[(size_t)(v % g->G) * g->n_rows + g->c0 + nlcf_perm_row0(g->d, g->G, perm) + v / g->G];
}
static size_t prow(const geom *g, uint32_t perm, uint32_t v) { return g->c0 + nlcf_perm_row0(g->d, g->G, perm) + v / g->G; }
/* digit t of the register tie j of a group binds (the GATED kinds: digit 0 the operation's cell, 1 / 2 the enables of the gate operations) */
static uint64_t *reg_cell(const geom *g, uint64_t *trace, const nlcf_group *gr, uint32_t j, uint32_t t) {
    const nl_spec *sp = g->sp;
    if (gr->reg_kind == NLCF_REG_QUEUE_BEFORE || gr->reg_kind == NLCF_REG_QUEUE_AFTER)
        return &trace[(size_t)nlq_bnd_col(g->qd, gr->queue, gr->reg_kind == NLCF_REG_QUEUE_AFTER, gr->reg0 + j) * g->n_rows + NLQ_BASE(sp, g->cycles)];
    if (gr->reg_kind == NLCF_REG_OP_FIRST || gr->reg_kind == NLCF_REG_OP_LAST) {
        const uint32_t op = t == 0 ? gr->queue : t == 1 ? gr->gate : gr->gate2, cell = t == 0 ? gr->reg0 : 0;
        const uint32_t c = gr->reg_kind == NLCF_REG_OP_LAST ? g->cycles - 1 : 0;
        if (op == NLCF_GATE_ACTIVE) return &trace[(size_t)NL_HDR_IDLE * g->n_rows + (size_t)c * sp->rows_per_cycle]; /* the cycle's idle bit */
        return &trace[(size_t)(cell % g->G) * g->n_rows + NLQ_ROW(sp, g->cycles, nlq_op_row0(g->qd, g->G, op) + cell / g->G, c)];
    }
    if (gr->reg_kind == NLCF_REG_FO_WORD) return hcell(g, trace, nlcf_word_cell(g->d, NLCF_FO, gr->reg0 + j)); /* the header block's own cell */
    const uint32_t e = (gr->reg0 + j) * gr->n_cells + t;
    const size_t row = NL_BOUNDARY_ROW(sp, g->cycles) + (gr->reg_kind == NLCF_REG_STATE_OUT ? NL_BND_ROWS(sp) : 0) + e / g->G;
    return &trace[(size_t)(e % g->G) * g->n_rows + row];
}
static uint64_t recompose(const geom *g, const uint64_t *trace, const nlcf_group *gr, uint32_t j, const uint64_t *digits /* or NULL: the registers */) {
    uint64_t acc = 0;
    for (uint32_t t = gr->n_cells; t-- > 0;) {
        const uint64_t x = digits ? digits[t] : *reg_cell(g, (uint64_t *)trace, gr, j, t);
        acc = orc_gl_add(gr->n_cells > 1 ? orc_gl_mul(acc, 1ull << gr->bits) : 0, x % P);
    }
    return acc;
}

/* the source of input v of P2 block `perm`: a cell of the section (returns its address) or a constant (returns NULL, *konst) */
static const uint64_t *perm_source(const geom *g, const uint64_t *trace, uint32_t perm, uint32_t v, uint64_t *konst) {
    const nlcf_desc *d = g->d;
    uint64_t *tr = (uint64_t *)trace;
    *konst = 0;
    const uint32_t cp0 = nlcf_perm0(d, 4);
    if (perm >= cp0) { /* the compact form: [start, completion, c(OI), c(OO), c(FI), c(FO)] */
        const uint32_t q = perm - cp0;
        if (v >= 8) { if (q) return pcell(g, tr, perm - 1, 118 + v); *konst = v == 11 ? NLCF_CP_WORDS : 0; return NULL; }
        const uint32_t w = 8 * q + v;
        if (w >= NLCF_CP_WORDS) return NULL;
        if (w < 2) return hcell(g, tr, w);
        const uint32_t part = (w - 2) / 4, k = (w - 2) % 4;
        if (d->n[part] == 0) return NULL; /* an empty encoding commits to zero */
        return pcell(g, tr, nlcf_perm0(d, part + 1) - 1, 118 + k);
    }
    uint32_t part = 0;
    while (perm >= nlcf_perm0(d, part + 1)) part++;
    const uint32_t q = perm - nlcf_perm0(d, part);
    if (v >= 8) { if (q) return pcell(g, tr, perm - 1, 118 + v); *konst = v == 11 ? d->n[part] : 0; return NULL; }
    const uint32_t w = 8 * q + v;
    return w < d->n[part] ? hcell(g, tr, nlcf_word_cell(d, part, w)) : NULL;
}

/* ---- fill from the words */
static int fill_words(int circuit_type, uint32_t cycles, size_t n_rows, uint64_t *trace, const int flags[2], uint64_t *const w[4], const size_t n[4]) {
    geom g;
    if (geom_of(circuit_type, cycles, n_rows, &g)) return -1;
    const nlcf_desc *d = g.d;
    for (int p = 0; p < 4; p++) if (n[p] != d->n[p]) return -2;
    /* the section's rows start from zero */
    for (size_t row = g.c0; row < g.c0 + nlcf_rows(d, g.G); row++)
        for (uint32_t col = 0; col < g.G; col++) TR(col, row) = 0;
    *hcell(&g, trace, NLCF_CELL_START) = (uint64_t)flags[0];
    *hcell(&g, trace, NLCF_CELL_COMPLETION) = (uint64_t)flags[1];
    for (uint32_t p = 0; p < 4; p++)
        for (uint32_t k = 0; k < d->n[p]; k++) *hcell(&g, trace, nlcf_word_cell(d, p, k)) = w[p][k];
    for (uint32_t gi = 0; gi < d->n_groups; gi++) {
        const nlcf_group *gr = &d->g[gi];
        for (uint32_t j = 0; j < gr->count; j++) {
            const uint32_t c = nlcf_tie_cell0(d, gi, j);
            const int32_t wa = nlcf_tie_word(gr, gr->a_word0, j), wb = nlcf_tie_word(gr, gr->b_word0, j);
            const uint64_t a = wa >= 0 ? w[nlcf_a_part(gr)][wa] : gr->a_const, b = wb >= 0 ? w[nlcf_b_part(gr)][wb] : 0;
            *hcell(&g, trace, c) = a;
            *hcell(&g, trace, c + 1) = b;
            for (uint32_t t = 0; t < gr->n_cells; t++) *hcell(&g, trace, c + 2 + t) = *reg_cell(&g, trace, gr, j, t); /* copies of what the other sections hold */
        }
    }
    for (uint32_t perm = 0; perm < nlcf_n_perms(d); perm++) {
        uint64_t in[12], slots[130];
        for (uint32_t v = 0; v < 12; v++) {
            uint64_t konst;
            const uint64_t *src = perm_source(&g, trace, perm, v, &konst);
            in[v] = src ? *src : konst;
        }
        orc_poseidon2_flattened(in, slots);
        for (uint32_t v = 0; v < NLQ_P2_CELLS; v++) *pcell(&g, trace, perm, v) = slots[v];
    }
    const size_t pi_row = NL_PI_ROW(g.sp, cycles);
    for (uint32_t k = 0; k < 4; k++) TR(k, pi_row) = *pcell(&g, trace, nlcf_n_perms(d) - 1, 118 + k);
    return 0;
}

/* The section of instance i of a block's instance records (zkw_precompile_instance / zkw_decommitter_instance / zkw_linear_hasher_instance /
   zkw_storage_application_instance), written below a trace whose other sections are filled; also (re)writes the PI row. */
int orc_nlcf_fill(int circuit_type, const void *instances, size_t i, uint32_t cycles, size_t n_rows, uint64_t *trace) {
    uint64_t *buf = malloc(4 * ORC_CF_MAX_FSM_LEN * sizeof(uint64_t));
    size_t n[4];
    int flags[2];
    int rc = orc_cf_encode(circuit_type, instances, i, buf, n, flags);
    uint64_t *const w[4] = {buf, buf + ORC_CF_MAX_FSM_LEN, buf + 2 * ORC_CF_MAX_FSM_LEN, buf + 3 * ORC_CF_MAX_FSM_LEN};
    if (rc == 0) rc = fill_words(circuit_type, cycles, n_rows, trace, flags, w, n);
    free(buf);
    return rc;
}

/* The section a trace without instance records implies (the bare-record entry points of netlist_circuit.c): not the first and not the
   last instance of its block, the hidden FSM words = what the registers hold, every other word zero; circuits without a hidden FSM:
   start = completion = 1 and the observable words = the registers. The PI row becomes the commitment of THAT closed form. */
int orc_nlcf_standalone(int circuit_type, uint32_t cycles, size_t n_rows, uint64_t *trace) {
    geom g;
    if (geom_of(circuit_type, cycles, n_rows, &g)) return -1;
    const nlcf_desc *d = g.d;
    uint64_t *buf = calloc(4 * ORC_CF_MAX_FSM_LEN, sizeof(uint64_t));
    uint64_t *const w[4] = {buf, buf + ORC_CF_MAX_FSM_LEN, buf + 2 * ORC_CF_MAX_FSM_LEN, buf + 3 * ORC_CF_MAX_FSM_LEN};
    const int no_fsm = d->n[NLCF_FI] == 0;
    const int flags[2] = {no_fsm, no_fsm};
    for (uint32_t gi = 0; gi < d->n_groups; gi++) {
        const nlcf_group *gr = &d->g[gi];
        for (uint32_t j = 0; j < gr->count; j++) {
            const uint64_t R = recompose(&g, trace, gr, j, NULL);
            const int32_t wa = nlcf_tie_word(gr, gr->a_word0, j), wb = nlcf_tie_word(gr, gr->b_word0, j);
            if (gr->kind == NLCF_IN_GATED) { /* start = 0: the FSM word the relation wants (register - add) */
                if (wb >= 0) w[NLCF_FI][wb] = orc_gl_sub(*reg_cell(&g, trace, gr, j, 0) % P, gr->add >= 0 ? (uint64_t)gr->add : P - (uint64_t)(-gr->add));
                continue;
            }
            if (gr->kind == NLCF_OUT_GATED) { /* completion = 0: the FSM word the relation wants */
                const uint64_t r = *reg_cell(&g, trace, gr, j, 0) % P, addf = gr->add >= 0 ? (uint64_t)gr->add : P - (uint64_t)(-gr->add);
                w[NLCF_FO][wa] = gr->negate ? orc_gl_sub(addf, r) : orc_gl_add(r, addf);
                continue;
            }
            if (gr->kind == NLCF_DONE) continue; /* (completion = 0 in the standalone form) */
            if (gr->kind == NLCF_IN && gr->reg_kind == NLCF_REG_FO_WORD) continue; /* second pass below: the FO words are being made here */
            if (gr->kind == NLCF_IN && wb >= 0) w[NLCF_FI][wb] = R;
            else if (gr->kind == NLCF_IN_ALWAYS || gr->kind == NLCF_OUT || gr->kind == NLCF_OUT_LIVE) w[nlcf_a_part(gr)][wa] = R;
            else if (gr->kind == NLCF_OUT_OO && wb < 0) w[NLCF_OO][wa] = R; /* (with an FSM: completion = 0, the word stays zero) */
        }
    }
    for (uint32_t gi = 0; gi < d->n_groups; gi++) { /* words an instance hands on unchanged: FI := what the FO pass above made */
        const nlcf_group *gr = &d->g[gi];
        if (gr->kind != NLCF_IN || gr->reg_kind != NLCF_REG_FO_WORD) continue;
        for (uint32_t j = 0; j < gr->count; j++) w[NLCF_FI][gr->b_word0 + j] = w[NLCF_FO][gr->reg0 + j];
    }
    const size_t n[4] = {d->n[0], d->n[1], d->n[2], d->n[3]};
    const int rc = fill_words(circuit_type, cycles, n_rows, trace, flags, w, n);
    free(buf);
    return rc;
}

/* ---- checker. Violation kinds as orc_nl_check: 2 copy constraint (a tie cell / a permutation input is not its source), 3 flag not
   boolean, 4 the PI row is not the compact form's commitment, 6 non-zero unused cell, 7 tie relation, 8 flattened Poseidon2 relation.
   code = (kind << 56) | (index << 32) | row with index = the header cell, or 2^20 + 130 perm + variable; the smallest code is reported. */
typedef struct { uint64_t n, first; } result;
static void flag(result *r, uint64_t kind, uint64_t idx, uint64_t row) {
    const uint64_t code = (kind << 56) | (idx << 32) | row;
    r->n++;
    if (code < r->first) r->first = code;
}
uint64_t orc_nlcf_check(int circuit_type, const uint64_t *trace, uint32_t cycles, size_t n_rows, uint64_t *first_bad) {
    geom g;
    result res = {0, ~0ull};
    if (geom_of(circuit_type, cycles, n_rows, &g)) { *first_bad = 0; return ~0ull; }
    const nlcf_desc *d = g.d;
    uint64_t *tr = (uint64_t *)trace;
    const uint64_t start = *hcell(&g, tr, NLCF_CELL_START), completion = *hcell(&g, tr, NLCF_CELL_COMPLETION);
    if (start > 1) flag(&res, 3, NLCF_CELL_START, hrow(&g, NLCF_CELL_START));
    if (completion > 1) flag(&res, 3, NLCF_CELL_COMPLETION, hrow(&g, NLCF_CELL_COMPLETION));
    for (uint32_t gi = 0; gi < d->n_groups; gi++) {
        const nlcf_group *gr = &d->g[gi];
        for (uint32_t j = 0; j < gr->count; j++) {
            const uint32_t c = nlcf_tie_cell0(d, gi, j);
            const int32_t wa = nlcf_tie_word(gr, gr->a_word0, j), wb = nlcf_tie_word(gr, gr->b_word0, j);
            const uint64_t a = *hcell(&g, tr, c), b = *hcell(&g, tr, c + 1);
            if (a != (wa >= 0 ? *hcell(&g, tr, nlcf_word_cell(d, nlcf_a_part(gr), (uint32_t)wa)) : gr->a_const)) flag(&res, 2, c, hrow(&g, c));
            if (b != (wb >= 0 ? *hcell(&g, tr, nlcf_word_cell(d, nlcf_b_part(gr), (uint32_t)wb)) : 0)) flag(&res, 2, c + 1, hrow(&g, c + 1));
            uint64_t digits[16];
            for (uint32_t t = 0; t < gr->n_cells; t++) {
                digits[t] = *hcell(&g, tr, c + 2 + t);
                if (digits[t] != *reg_cell(&g, tr, gr, j, t)) flag(&res, 2, c + 2 + t, hrow(&g, c + 2 + t));
            }
            const uint64_t R = recompose(&g, trace, gr, j, digits), am = a % P, bm = b % P;
            const uint64_t addf = gr->add >= 0 ? (uint64_t)gr->add : P - (uint64_t)(-gr->add);
            int ok;
            switch (gr->kind) {
                case NLCF_IN_GATED: { /* (g1 - g2) (r - (b + start (a - b)) - add) = 0 */
                    const uint64_t g1 = gr->gate == NLCF_GATE_ACTIVE ? orc_gl_sub(1, digits[1] % P) : digits[1] % P;
                    const uint64_t gate = orc_gl_sub(g1, gr->n_cells > 2 ? digits[2] % P : 0);
                    const uint64_t sel = orc_gl_add(bm, orc_gl_mul(start % P, orc_gl_sub(am, bm)));
                    ok = orc_gl_mul(gate, orc_gl_sub(orc_gl_sub(digits[0] % P, sel), addf)) == 0;
                    break;
                }
                case NLCF_OUT_GATED: { /* (1 - completion) (g1 - g2) (a -/+ r - add) = 0 */
                    const uint64_t r = digits[0] % P, lhs = gr->negate ? orc_gl_add(am, r) : orc_gl_sub(am, r);
                    const uint64_t gate = gr->n_cells < 2 ? 1 : orc_gl_sub(digits[1] % P, gr->n_cells > 2 ? digits[2] % P : 0);
                    ok = orc_gl_mul(orc_gl_mul(orc_gl_sub(1, completion % P), gate), orc_gl_sub(lhs, addf)) == 0;
                    break;
                }
                case NLCF_IN: ok = R == orc_gl_add(bm, orc_gl_mul(start % P, orc_gl_sub(am, bm))); break;
                case NLCF_OUT_LIVE: ok = completion == 1 || R == am; break;
                case NLCF_DONE: ok = orc_gl_mul(completion % P, orc_gl_sub(R, am)) == 0; break;
                case NLCF_OUT_OO: ok = wb >= 0 ? (R == bm && am == orc_gl_mul(completion % P, bm)) : (R == am && am == orc_gl_mul(completion % P, R)); break;
                default: ok = R == am; break;
            }
            if (!ok) flag(&res, 7, c, hrow(&g, c));
        }
    }
    const uint32_t hc = nlcf_header_cells(d), hr = nlcf_header_rows(d, g.G);
    for (uint32_t k = hc; k < hr * g.G; k++)
        if (*hcell(&g, tr, k)) { flag(&res, 6, k % g.G, hrow(&g, k)); break; }
    const uint32_t prows = nlq_rows_for(NLQ_P2_CELLS, g.G);
    for (uint32_t perm = 0; perm < nlcf_n_perms(d); perm++) {
        uint64_t in[12], slots[130];
        for (uint32_t v = 0; v < 12; v++) {
            uint64_t konst;
            const uint64_t *src = perm_source(&g, trace, perm, v, &konst);
            in[v] = *pcell(&g, tr, perm, v);
            if (in[v] != (src ? *src : konst)) flag(&res, 2, (1u << 20) + 130 * perm + v, prow(&g, perm, v));
        }
        orc_poseidon2_flattened(in, slots);
        for (uint32_t v = 12; v < NLQ_P2_CELLS; v++)
            if (*pcell(&g, tr, perm, v) != slots[v]) { flag(&res, 8, (1u << 20) + 130 * perm + v, prow(&g, perm, v)); break; }
        for (uint32_t v = NLQ_P2_CELLS; v < prows * g.G; v++)
            if (*pcell(&g, tr, perm, v)) { flag(&res, 6, v % g.G, prow(&g, perm, v)); break; }
    }
    const size_t pi_row = NL_PI_ROW(g.sp, cycles);
    for (uint32_t k = 0; k < 4; k++)
        if (TR(k, pi_row) != *pcell(&g, tr, nlcf_n_perms(d) - 1, 118 + k)) flag(&res, 4, k, pi_row);
    *first_bad = res.n ? res.first : 0;
    return res.n;
}

/* for the tests: {first row, rows, rows used by the whole trace, header cells, permutations, words of OI / OO / FI / FO} */
void orc_nlcf_geometry(int circuit_type, uint32_t cycles, uint64_t out[9]) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    const nlcf_desc *d = nlcf_desc_of(circuit_type);
    memset(out, 0, 9 * sizeof(uint64_t));
    if (!sp || !d) return;
    out[0] = nlcf_first_row(circuit_type, sp, cycles);
    out[1] = nlcf_rows(d, sp->g);
    out[2] = nlcf_used_rows(circuit_type, sp, cycles);
    out[3] = nlcf_header_cells(d);
    out[4] = nlcf_n_perms(d);
    for (int p = 0; p < 4; p++) out[5 + p] = d->n[p];
}
/* (column, row) of a cell: what = 0 header cell k (0 start, 1 completion); 1..4 word k of OI / OO / FI / FO; 5 variable k % 130 of P2
   block k / 130; 6 cell `k & 0x3FFF` of tie `k >> 14 & 0xFFF` of group `k >> 26`; 7 PI cell k */
int orc_nlcf_cell(int circuit_type, uint32_t cycles, int what, uint32_t k, uint64_t out[2]) {
    geom g;
    if (geom_of(circuit_type, cycles, (size_t)1 << 40, &g)) return -1;
    uint32_t cell;
    if (what == 0) cell = k;
    else if (what >= 1 && what <= 4) { if (k >= g.d->n[what - 1]) return -2; cell = nlcf_word_cell(g.d, (uint32_t)what - 1, k); }
    else if (what == 5) {
        if (k / 130 >= nlcf_n_perms(g.d)) return -2;
        out[0] = (k % 130) % g.G; out[1] = prow(&g, k / 130, k % 130);
        return 0;
    } else if (what == 6) {
        const uint32_t gi = k >> 26, j = (k >> 14) & 0xFFF, c = k & 0x3FFF;
        if (gi >= g.d->n_groups || j >= g.d->g[gi].count || c >= 2u + g.d->g[gi].n_cells) return -2;
        cell = nlcf_tie_cell0(g.d, gi, j) + c;
    } else if (what == 7) { out[0] = k; out[1] = NL_PI_ROW(g.sp, cycles); return 0; }
    else return -2;
    out[0] = cell % g.G; out[1] = hrow(&g, cell);
    return 0;
}
