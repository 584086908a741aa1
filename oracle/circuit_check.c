/* circuit_check.c — TEST INFRASTRUCTURE: satisfiability check of a filled trace against a spec table
 * (include/zkw_*_circuit_spec.h), the counterpart of `check_if_satisfied` in the reference's tests
 * (src/tests/mod.rs:130-259). Generic interpreter of the tables; shares no code with any fill. */
#include "oracle.h"
#include "../include/zkw_ram_circuit_spec.h"
#include "../include/zkw_decommit_sorter_circuit_spec.h"
#include "../include/zkw_events_sorter_circuit_spec.h"
#include "../include/zkw_log_demux_circuit_spec.h"
#include "../include/zkw_storage_sorter_circuit_spec.h"
#include <stdlib.h>
#include <string.h>

#define P ZKW_GOLDILOCKS_P
#define CELL(col, row) trace[(size_t)(col) * n_rows + (row)]

typedef struct {
    int G, L, rows_per_cycle, n_row_types, n_links, off_bin, off_bout;
    const rc_term *terms;
    const rc_constraint *cons;
    const uint16_t *row_first;
    const uint8_t *is_poseidon;
    const rc_link *links;
} orc_spec;

static const rc_term DS_TERMS[] = DS_TERMS_INIT;
static const rc_constraint DS_CONS[] = DS_CONSTRAINTS_INIT;
static const uint16_t DS_ROW_FIRST[] = DS_ROW_FIRST_CONSTRAINT_INIT;
static const uint8_t DS_IS_POSEIDON[] = DS_ROW_IS_POSEIDON_INIT;
static const rc_link DS_LINKS[] = DS_LINKS_INIT;
static const orc_spec SPEC_DS = {DS_G, DS_L, DS_ROWS_PER_CYCLE, DS_NUM_ROW_TYPES, DS_NUM_LINKS, DS_ROWOFF_BND_IN, DS_ROWOFF_BND_OUT,
                                 DS_TERMS, DS_CONS, DS_ROW_FIRST, DS_IS_POSEIDON, DS_LINKS};

static const rc_term ES_TERMS[] = ES_TERMS_INIT;
static const rc_constraint ES_CONS[] = ES_CONSTRAINTS_INIT;
static const uint16_t ES_ROW_FIRST[] = ES_ROW_FIRST_CONSTRAINT_INIT;
static const uint8_t ES_IS_POSEIDON[] = ES_ROW_IS_POSEIDON_INIT;
static const rc_link ES_LINKS[] = ES_LINKS_INIT;
static const orc_spec SPEC_ES = {ES_G, ES_L, ES_ROWS_PER_CYCLE, ES_NUM_ROW_TYPES, ES_NUM_LINKS, ES_ROWOFF_BND_IN, ES_ROWOFF_BND_OUT,
                                 ES_TERMS, ES_CONS, ES_ROW_FIRST, ES_IS_POSEIDON, ES_LINKS};

static const rc_term LD_TERMS[] = LD_TERMS_INIT;
static const rc_constraint LD_CONS[] = LD_CONSTRAINTS_INIT;
static const uint16_t LD_ROW_FIRST[] = LD_ROW_FIRST_CONSTRAINT_INIT;
static const uint8_t LD_IS_POSEIDON[] = LD_ROW_IS_POSEIDON_INIT;
static const rc_link LD_LINKS[] = LD_LINKS_INIT;
static const orc_spec SPEC_LD = {LD_G, LD_L, LD_ROWS_PER_CYCLE, LD_NUM_ROW_TYPES, LD_NUM_LINKS, LD_ROWOFF_BND_IN, LD_ROWOFF_BND_OUT,
                                 LD_TERMS, LD_CONS, LD_ROW_FIRST, LD_IS_POSEIDON, LD_LINKS};

static const rc_term SS_TERMS[] = SS_TERMS_INIT;
static const rc_constraint SS_CONS[] = SS_CONSTRAINTS_INIT;
static const uint16_t SS_ROW_FIRST[] = SS_ROW_FIRST_CONSTRAINT_INIT;
static const uint8_t SS_IS_POSEIDON[] = SS_ROW_IS_POSEIDON_INIT;
static const rc_link SS_LINKS[] = SS_LINKS_INIT;
static const orc_spec SPEC_SS = {SS_G, SS_L, SS_ROWS_PER_CYCLE, SS_NUM_ROW_TYPES, SS_NUM_LINKS, SS_ROWOFF_BND_IN, SS_ROWOFF_BND_OUT,
                                 SS_TERMS, SS_CONS, SS_ROW_FIRST, SS_IS_POSEIDON, SS_LINKS};

static const rc_term RC_TERMS[] = RC_TERMS_INIT;
static const rc_constraint RC_CONS[] = RC_CONSTRAINTS_INIT;
static const uint16_t RC_ROW_FIRST[] = RC_ROW_FIRST_CONSTRAINT_INIT;
static const uint8_t RC_IS_POSEIDON[] = RC_ROW_IS_POSEIDON_INIT;
static const rc_link RC_LINKS[] = RC_LINKS_INIT;
static const orc_spec SPEC_RC = {RC_G, RC_L, RC_ROWS_PER_CYCLE, RC_NUM_ROW_TYPES, RC_NUM_LINKS, RC_ROWOFF_BND_IN, RC_ROWOFF_BND_OUT,
                                 RC_TERMS, RC_CONS, RC_ROW_FIRST, RC_IS_POSEIDON, RC_LINKS};

static uint64_t eval_constraint(const orc_spec *sp, const rc_constraint *c, const uint64_t *trace, size_t n_rows, size_t row) {
    uint64_t acc = 0;
    for (int t = 0; t < c->n_terms; t++) {
        const rc_term *tm = &sp->terms[c->first_term + t];
        uint64_t v = tm->coef;
        for (int f = 0; f < tm->nf; f++) v = orc_gl_mul(v, CELL(tm->f[f], row));
        acc = orc_gl_add(acc, v);
    }
    return acc % P;
}

/* Returns the number of violated relations (0 = satisfied); `first_bad` receives a description code:
   (kind << 56) | (index << 32) | row  with kind 1 constraint, 2 poseidon, 3 lookup range, 4 copy link,
   5 multiplicity, 6 non-zero padding / non-canonical value. */
static uint64_t check(const orc_spec *sp, const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    uint64_t bad = 0;
#define FLAG(kind, idx, row)                                                                        \
    do {                                                                                            \
        if (!bad && first_bad) *first_bad = ((uint64_t)(kind) << 56) | ((uint64_t)(idx) << 32) | (uint64_t)(row); \
        bad++;                                                                                      \
    } while (0)
    const size_t rs = (size_t)RC_REGION_STRIDE(capacity), bnd = rs * sp->rows_per_cycle;
    const int n_bnd = sp->n_row_types - sp->rows_per_cycle, cols = sp->G + sp->L, mult = sp->G + sp->L;
#define ROWOF(rt, i) ((size_t)(rt) < (size_t)sp->rows_per_cycle ? (size_t)(rt) * rs + (i) : bnd + (size_t)((rt) - sp->rows_per_cycle))
    for (int rt = 0; rt < sp->n_row_types; rt++) {
        const size_t n_in = rt < sp->rows_per_cycle ? capacity : 1;
        for (size_t i = 0; i < n_in; i++) {
            const size_t row = ROWOF(rt, i);
            for (int c = sp->row_first[rt]; c < sp->row_first[rt + 1]; c++)
                if (eval_constraint(sp, &sp->cons[c], trace, n_rows, row)) FLAG(1, c, row);
            if (sp->is_poseidon[rt]) {
                uint64_t in[12], slots[130];
                for (int k = 0; k < 12; k++) in[k] = CELL(k, row);
                orc_poseidon2_flattened(in, slots);
                for (int k = 0; k < 130; k++)
                    if (slots[k] != CELL(k, row)) { FLAG(2, k, row); break; }
            }
        }
    }
    for (int l = 0; l < sp->n_links; l++) {
        const rc_link *k = &sp->links[l];
        if (k->kind == 3) {
            if (CELL(k->col_a, bnd + sp->off_bout) != CELL(k->col_b, ROWOF(k->row_b, capacity - 1))) FLAG(4, l, bnd + sp->off_bout);
            continue;
        }
        if (k->kind == 4) { /* a boundary row's cell equals a BND_OUT cell */
            if (CELL(k->col_a, ROWOF(k->row_a, 0)) != CELL(k->col_b, bnd + sp->off_bout)) FLAG(4, l, ROWOF(k->row_a, 0));
            continue;
        }
        if (k->kind == 5) { /* a boundary row's cell equals a cell of another boundary row */
            if (CELL(k->col_a, ROWOF(k->row_a, 0)) != CELL(k->col_b, ROWOF(k->row_b, 0))) FLAG(4, l, ROWOF(k->row_a, 0));
            continue;
        }
        for (size_t i = 0; i < capacity; i++) {
            const uint64_t a = CELL(k->col_a, ROWOF(k->row_a, i));
            uint64_t b;
            if (k->kind == 0) b = CELL(k->col_b, ROWOF(k->row_b, i));
            else if (k->kind == 1) b = i ? CELL(k->col_b, ROWOF(k->row_b, i - 1)) : CELL(k->bin_col, bnd + sp->off_bin);
            else b = CELL(k->col_b, bnd + sp->off_bin);
            if (a != b) FLAG(4, l, ROWOF(k->row_a, i));
        }
    }
    uint64_t *hist = (uint64_t *)calloc(256, 8);
    for (int c = sp->G; c < cols; c++)
        for (size_t r = 0; r < n_rows; r++) {
            uint64_t v = CELL(c, r);
            if (v > 255) FLAG(3, c, r); else hist[v]++;
        }
    for (size_t r = 0; r < n_rows; r++) {
        uint64_t want = r < 256 ? hist[r] : 0;
        if (CELL(mult, r) != want) FLAG(5, 0, r);
    }
    free(hist);
    for (size_t r = bnd + n_bnd; r < n_rows; r++)
        for (int c = 0; c < cols; c++)
            if (CELL(c, r)) { FLAG(6, c, r); break; }
    for (int rt = 0; rt < sp->rows_per_cycle; rt++) /* the alignment gap at the end of every region */
        for (size_t r = (size_t)rt * rs + capacity; r < (size_t)(rt + 1) * rs; r++)
            for (int c = 0; c < cols; c++)
                if (CELL(c, r)) { FLAG(6, c, r); break; }
    for (int c = 0; c <= mult; c++)
        for (size_t r = 0; r < n_rows; r++)
            if (CELL(c, r) >= P) FLAG(6, c, r);
    return bad;
#undef FLAG
#undef ROWOF
}

uint64_t orc_ram_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    return check(&SPEC_RC, trace, capacity, n_rows, first_bad);
}
uint64_t orc_decommit_sorter_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    return check(&SPEC_DS, trace, capacity, n_rows, first_bad);
}
uint64_t orc_events_sorter_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    return check(&SPEC_ES, trace, capacity, n_rows, first_bad);
}
uint64_t orc_log_demux_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    return check(&SPEC_LD, trace, capacity, n_rows, first_bad);
}
uint64_t orc_storage_sorter_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    return check(&SPEC_SS, trace, capacity, n_rows, first_bad);
}
