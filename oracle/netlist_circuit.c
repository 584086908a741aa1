/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * "zkw trace v4": the generic netlist circuits (format: include/zkw_netlist.h, tools/netlist.py) — Sha256RoundFunction (6),
 * CodeDecommitter (3), Keccak256RoundFunction (5), L1MessagesHasher (13) on the reference's geometry and lookup-table sets
 * (wrappers: circuit_definitions/src/circuit_definitions/base_layer/{sha256_round_function,code_decommitter,
 * keccak256_round_function,linear_hasher}.rs:28-39 and their add_tables). The circuit bodies are in the absent crate
 * era-zkevm_circuits: PARITY UNPINNED at the placement level; geometry, table sets (= `total_tables_len` of vk_{6,3,5,13}.json) and
 * capacities are the reference's. ONE fill and ONE checker, driven by the per-circuit spec; sequential and obvious: the fill
 * evaluates a step's items in dependency order, the checker re-derives every relation from the cells alone.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "../include/zkw_sha256_circuit_spec.h"
#include "../include/zkw_code_decommitter_circuit_spec.h"
#include "../include/zkw_keccak_circuit_spec.h"
#include "../include/zkw_linear_hasher_circuit_spec.h"
#include "../include/zkw_storage_application_circuit_spec.h"
#include "../include/zkw_ecrecover_circuit_spec.h"
#include "../include/zkw_netlist_closed_form.h"

NL_DEFINE_SPEC(sc, SC);
NL_DEFINE_SPEC(dc, DC);
NL_DEFINE_SPEC(kc, KC);
NL_DEFINE_SPEC(lh, LH);
NL_DEFINE_SPEC(sa, SA);
NL_DEFINE_SPEC(ek, EK); /* ECRecover (7): the Keccak-f netlist over the public key; its EC section: ecrecover_circuit.c */

const nl_spec *orc_nl_spec(int circuit_type) {
    switch (circuit_type) {
        case 6: return &sc_spec;
        case 3: return &dc_spec;
        case 5: return &kc_spec;
        case 13: return &lh_spec;
        case 10: return &sa_spec;
        case 7: return &ek_spec;
        default: return NULL;
    }
}

#define P ZKW_GOLDILOCKS_P
#define TR(c, r) trace[(size_t)(c) * n_rows + (size_t)(r)]

typedef struct {
    const nl_spec *sp;
    const uint64_t *trace;
    size_t n_rows;
    uint32_t capacity;
} view;

/* the cell (or constant) a reference names, seen from step `s` of cycle `c` */
static uint64_t home(const view *v, uint32_t c, uint32_t s, uint32_t ref) {
    const nl_spec *sp = v->sp;
    const uint64_t *trace = v->trace;
    const size_t n_rows = v->n_rows;
    for (;;) {
        const nl_cycle_step *cs = &sp->cycle[s];
        const nl_step_type *T = &sp->step_types[cs->type];
        const size_t base = (size_t)c * sp->rows_per_cycle + cs->row0;
        if (ref < NL_REF_HDR) {
            const nl_home h = sp->homes[T->home0 + ref];
            if (h.kind == 1) {
                const nl_gate *g = &sp->gates[T->gate0 + h.item];
                return TR(g->col + h.cell, base + g->row);
            }
            return TR(sp->g + sp->w * (h.item % sp->r) + h.cell, base + 1 + h.item / sp->r);
        }
        if (ref < NL_REF_PREV) return TR(ref - NL_REF_HDR, base);
        if (ref >= NL_REF_CONST) return ref - NL_REF_CONST;
        if (ref >= NL_REF_RC) return cs->rc[ref - NL_REF_RC];
        if (ref >= NL_REF_FREE) return 0; /* a free witness has no home: callers skip it */
        uint32_t k;
        if (ref >= NL_REF_CYC || s == 0) { /* the state before this cycle */
            k = ref >= NL_REF_CYC ? ref - NL_REF_CYC : ref - NL_REF_PREV;
            if (c == 0) {
                const size_t bnd = NL_BOUNDARY_ROW(sp, v->capacity);
                return TR(k % sp->g, bnd + k / sp->g);
            }
            c--;
            s = sp->steps_per_cycle - 1;
        } else {
            k = ref - NL_REF_PREV;
            s--;
        }
        ref = sp->out[(size_t)sp->cycle[s].type * sp->state + k];
    }
}

uint64_t orc_nl_home(const nl_spec *sp, const uint64_t *trace, size_t n_rows, uint32_t capacity, uint32_t c, uint32_t s, uint32_t ref) {
    const view v = {sp, trace, n_rows, capacity};
    return home(&v, c, s, ref);
}

static int type_of_spec(const nl_spec *sp) {
    static const int types[] = {6, 3, 5, 13, 10, 7};
    for (int i = 0; i < 6; i++)
        if (orc_nl_spec(types[i]) == sp) return types[i];
    return 0;
}

/* free elements of a cycle are laid out step after step */
static uint32_t free_offset(const nl_spec *sp, uint32_t s) {
    uint32_t off = 0;
    for (uint32_t i = 0; i < s; i++) off += sp->step_types[sp->cycle[i].type].n_free;
    return off;
}

/* The one cell that holds FREE element `free_index` of a cycle (an element is used once: tools/netlist.py asserts it): its row within
   the cycle and its column. Returns 0, or -1 when no item uses the element. */
int orc_nl_free_home(const nl_spec *sp, uint32_t free_index, uint32_t *row_in_cycle, uint32_t *col) {
    for (uint32_t s = 0; s < sp->steps_per_cycle; s++) {
        const nl_cycle_step *cs = &sp->cycle[s];
        const nl_step_type *T = &sp->step_types[cs->type];
        const uint32_t off = free_offset(sp, s);
        if (free_index < off || free_index >= off + T->n_free) continue;
        const uint32_t ref = NL_REF_FREE + (free_index - off);
        for (uint32_t j = 0; j < T->n_ops; j++) {
            const nl_op *op = &sp->ops[T->op0 + j];
            if (op->out == 0xFFFF) continue;
            for (uint32_t i = 0; i < sp->tables[op->table - 1].n_in; i++)
                if (op->in[i] == ref) { *row_in_cycle = cs->row0 + 1 + j / sp->r; *col = sp->g + sp->w * (j % sp->r) + i; return 0; }
        }
        for (uint32_t gi = 0; gi < T->n_gates; gi++) {
            const nl_gate *g = &sp->gates[T->gate0 + gi];
            const nl_term *tm = &sp->terms[T->term0 + g->first_term];
            for (uint32_t i = 0; i < g->n_known; i++)
                if (tm[i].ref == ref) { *row_in_cycle = cs->row0 + g->row; *col = g->col + i; return 0; }
        }
    }
    return -1;
}

int orc_nl_synthesize(const nl_spec *sp, uint32_t capacity, const uint8_t *hdr_bits, const uint8_t *free_elems, const uint8_t *state_before,
                      const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    if (!sp || n_rows < NL_USED_ROWS(sp, capacity) || n_rows < sp->total_table_rows) return -1;
    memset(trace, 0, (size_t)sp->cols * n_rows * sizeof(uint64_t));
    uint32_t *hist = calloc(sp->total_table_rows, sizeof(uint32_t));
    uint8_t *val = malloc(sp->max_values);
    uint8_t *prev = malloc(sp->state), *cyc = malloc(sp->state), *next = malloc(sp->state);
    int rc = 0;
    for (uint32_t c = 0; c < capacity && rc == 0; c++) {
        const uint32_t reset = hdr_bits[c] & 1, idle = (hdr_bits[c] >> 1) & 1;
        const uint8_t hdr[4] = {(uint8_t)reset, (uint8_t)idle, (uint8_t)(sp->masks[0] + sp->masks[1] * (int)reset), (uint8_t)(sp->masks[2] + sp->masks[3] * (int)idle)};
        memcpy(cyc, state_before + (size_t)c * sp->state, sp->state);
        memcpy(prev, cyc, sp->state);
        for (uint32_t s = 0; s < sp->steps_per_cycle; s++) {
            const nl_cycle_step *cs = &sp->cycle[s];
            const nl_step_type *T = &sp->step_types[cs->type];
            const size_t base = (size_t)c * sp->rows_per_cycle + cs->row0;
            const uint8_t *fr = free_elems + (size_t)c * sp->free_per_cycle + free_offset(sp, s);
#define REF(x) ((x) < NL_REF_HDR ? val[x] : (x) < NL_REF_PREV ? hdr[(x) - NL_REF_HDR] : (x) < NL_REF_CYC ? prev[(x) - NL_REF_PREV] \
                : (x) < NL_REF_FREE ? cyc[(x) - NL_REF_CYC] : (x) < NL_REF_RC ? fr[(x) - NL_REF_FREE] : (x) < NL_REF_CONST ? cs->rc[(x) - NL_REF_RC] : (uint8_t)((x) - NL_REF_CONST))
            for (int f = 0; f < NL_HDR_FIELDS; f++) TR(f, base) = hdr[f];
            const uint32_t n_items = sp->level_start[T->level0 + T->n_levels]; /* (a fused hint + lookup is one item) */
            for (uint32_t e = 0; e < n_items; e++) {
                uint32_t it = sp->order[T->order0 + e];
                if (it >= NL_ORDER_FUSED && it < NL_ORDER_GATE) { /* the hint first, then the lookup it keys */
                    const nl_hint *h = &sp->hints[T->hint0 + (it - NL_ORDER_FUSED)];
                    const uint32_t a = REF(h->ref_a), b = REF(h->ref_b);
                    val[h->value] = (uint8_t)(((a >> h->lo_a) & ((1u << h->n_a) - 1)) | (((b >> h->lo_b) & ((1u << h->n_b) - 1)) << h->n_a));
                    it = h->fused_slot;
                }
                if (it < NL_ORDER_GATE) {
                    const nl_op *op = &sp->ops[T->op0 + it];
                    const nl_table *t = &sp->tables[op->table - 1];
                    const size_t row = base + 1 + it / sp->r;
                    const uint32_t col = sp->g + sp->w * (it % sp->r);
                    uint32_t a[3] = {0, 0, 0}, o[3];
                    for (uint32_t i = 0; i < t->n_in; i++) a[i] = REF(op->in[i]);
                    orc_nl_lookup(t, a, o);
                    for (uint32_t i = 0; i < t->n_in; i++) TR(col + i, row) = a[i];
                    for (uint32_t i = 0; i < t->n_out; i++) TR(col + t->n_in + i, row) = o[i];
                    if (op->out != 0xFFFF) {
                        for (uint32_t i = 0; i < t->n_out; i++) val[op->out + i] = (uint8_t)o[i];
                        hist[orc_nl_multiplicity_row(t, a)]++;
                    }
                } else if (it < NL_ORDER_HINT) {
                    const nl_gate *g = &sp->gates[T->gate0 + (it - NL_ORDER_GATE)];
                    const nl_term *tm = &sp->terms[T->term0 + g->first_term];
                    int64_t S = g->constant;
                    for (uint32_t i = 0; i < g->n_known; i++) {
                        if (tm[i].code & NL_TERM_LATE) continue; /* in the constraint, not in the evaluation: written below */
                        const int64_t x = REF(tm[i].ref);
                        TR(g->col + i, base + g->row) = (uint64_t)x;
                        S += (tm[i].code & 0x80) ? -(x << (tm[i].code & 0x7F)) : (x << (tm[i].code & 0x7F));
                    }
                    if (S < 0) rc = -2;
                    int has_late = 0;
                    for (uint32_t i = 0; i < g->n_known; i++) has_late |= (tm[i].code & NL_TERM_LATE) != 0;
                    for (uint32_t i = 0; i < g->n_new && rc == 0; i++) {
                        const uint32_t sh = tm[g->n_known + i].code & 0x7F;
                        uint64_t x = (uint64_t)S >> sh;
                        if (i + 1 < g->n_new) x &= (1ull << ((tm[g->n_known + i + 1].code & 0x7F) - sh)) - 1;
                        else if (has_late && i > 0) x &= (1ull << (sh - (tm[g->n_known + i - 1].code & 0x7F))) - 1; /* a digit like the others */
                        if (x > 255) rc = -3;
                        val[tm[g->n_known + i].ref] = (uint8_t)x;
                        TR(g->col + g->n_known + i, base + g->row) = x;
                    }
                    if (g->n_new == 0 && S != 0) rc = -4;
                } else {
                    const nl_hint *h = &sp->hints[T->hint0 + (it - NL_ORDER_HINT)];
                    const uint32_t a = REF(h->ref_a), b = REF(h->ref_b);
                    val[h->value] = (uint8_t)(((a >> h->lo_a) & ((1u << h->n_a) - 1)) | (((b >> h->lo_b) & ((1u << h->n_b) - 1)) << h->n_a));
                }
            }
            for (uint32_t gi = 0; gi < T->n_gates; gi++) { /* the late cells of the gates: their producers have run by now */
                const nl_gate *g = &sp->gates[T->gate0 + gi];
                const nl_term *tm = &sp->terms[T->term0 + g->first_term];
                for (uint32_t i = 0; i < g->n_known; i++)
                    if (tm[i].code & NL_TERM_LATE) TR(g->col + i, base + g->row) = (uint64_t)REF(tm[i].ref);
            }
            const uint16_t *out = sp->out + (size_t)cs->type * sp->state;
            for (uint32_t k = 0; k < sp->state; k++) next[k] = REF(out[k]);
            memcpy(prev, next, sp->state);
#undef REF
        }
        if (rc == 0 && memcmp(prev, state_before + (size_t)(c + 1) * sp->state, sp->state) != 0) rc = -5; /* the netlist disagrees with the builder */
    }
    /* padding lookups (all inputs 0) hit entry 0 of their table */
    for (uint32_t s = 0; s < sp->steps_per_cycle; s++) {
        const nl_step_type *T = &sp->step_types[sp->cycle[s].type];
        for (uint32_t j = 0; j < T->n_ops; j++)
            if (sp->ops[T->op0 + j].out == 0xFFFF) hist[sp->tables[sp->ops[T->op0 + j].table - 1].offset] += capacity;
    }
    for (uint32_t e = 0; e < sp->total_table_rows; e++) TR(sp->mult_col, e) = hist[e];
    const size_t bnd = NL_BOUNDARY_ROW(sp, capacity), brows = NL_BND_ROWS(sp);
    for (uint32_t k = 0; k < sp->state; k++) {
        TR(k % sp->g, bnd + k / sp->g) = state_before[k];
        TR(k % sp->g, bnd + brows + k / sp->g) = state_before[(size_t)capacity * sp->state + k];
    }
    for (int k = 0; k < 4; k++) TR(k, bnd + 2 * brows) = pi ? pi[k] : 0;
    free(hist); free(val); free(prev); free(cyc); free(next);
    return rc;
}

/* ---- checker: violation kinds 1 lookup relation / range, 2 copy constraint, 3 header, 4 boundary, 5 multiplicity, 6 non-zero
   unused cell, 7 gate arithmetic; code = (kind << 56) | (index << 32) | row, the smallest code is reported */
typedef struct { uint64_t n, first; } result;
static void flag(result *r, uint64_t kind, uint64_t idx, uint64_t row) {
    const uint64_t code = (kind << 56) | (idx << 32) | row;
    r->n++;
    if (code < r->first) r->first = code;
}

static uint64_t fadd(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }
static uint64_t fmul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)(a % P) * (b % P)) % P); }

uint64_t orc_nl_check(const nl_spec *sp, const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    result res = {0, ~0ull};
    const int ctype = type_of_spec(sp);
    const nlq_desc *qd = nlq_desc_of(ctype);
    if (!sp || n_rows < nlcf_used_rows(ctype, sp, capacity)) { *first_bad = 0; return ~0ull; }
    const size_t q_begin = NL_USED_ROWS(sp, capacity), q_end = nlq_used_rows(sp, qd, capacity); /* the queue section's rows (zkw_netlist_queue.h) */
    const size_t c_begin = nlcf_first_row(ctype, sp, capacity), c_end = nlcf_used_rows(ctype, sp, capacity); /* the closed-form section's (zkw_netlist_closed_form.h) */
    const view v = {sp, trace, n_rows, capacity};
    uint32_t *hist = calloc(sp->total_table_rows, sizeof(uint32_t));
    for (uint32_t c = 0; c < capacity; c++)
        for (uint32_t s = 0; s < sp->steps_per_cycle; s++) {
            const nl_cycle_step *cs = &sp->cycle[s];
            const nl_step_type *T = &sp->step_types[cs->type];
            const size_t base = (size_t)c * sp->rows_per_cycle + cs->row0;
            /* header row */
            if (s == 0) {
                const uint64_t reset = TR(NL_HDR_RESET, base), idle = TR(NL_HDR_IDLE, base);
                if (reset > 1 || idle > 1) flag(&res, 3, 0, base);
                else if (TR(NL_HDR_M0, base) != (uint64_t)(sp->masks[0] + sp->masks[1] * (int64_t)reset) ||
                         TR(NL_HDR_M1, base) != (uint64_t)(sp->masks[2] + sp->masks[3] * (int64_t)idle)) flag(&res, 3, 1, base);
            } else {
                const size_t b0 = (size_t)c * sp->rows_per_cycle;
                for (int f = 0; f < NL_HDR_FIELDS; f++)
                    if (TR(f, base) != TR(f, b0)) { flag(&res, 3, 2, base); break; }
            }
            /* lookups */
            for (uint32_t j = 0; j < T->n_ops; j++) {
                const nl_op *op = &sp->ops[T->op0 + j];
                const nl_table *t = &sp->tables[op->table - 1];
                const size_t row = base + 1 + j / sp->r;
                const uint32_t slot = j % sp->r, col = sp->g + sp->w * slot;
                uint32_t a[3] = {0, 0, 0}, o[3];
                int ok = 1;
                for (uint32_t i = 0; i < t->n_in; i++) {
                    const uint64_t x = TR(col + i, row);
                    if (x >> t->in_bits) ok = 0;
                    a[i] = (uint32_t)x;
                }
                if (ok) {
                    orc_nl_lookup(t, a, o);
                    for (uint32_t i = 0; i < t->n_out; i++) ok &= TR(col + t->n_in + i, row) == o[i];
                    for (uint32_t i = t->n_in + t->n_out; i < sp->w; i++) ok &= TR(col + i, row) == 0;
                }
                if (!ok) { flag(&res, 1, slot, row); continue; }
                int copies = 1;
                for (uint32_t i = 0; i < t->n_in; i++) {
                    const uint32_t ref = op->in[i];
                    if (ref >= NL_REF_FREE && ref < NL_REF_RC) continue;
                    if (ref < NL_REF_HDR) {
                        const nl_home h = sp->homes[T->home0 + ref];
                        if (h.kind == 2 && h.item == j && h.cell == i) continue; /* the hint's own cell */
                    }
                    if (a[i] != home(&v, c, s, ref)) copies = 0;
                }
                if (!copies) flag(&res, 2, slot, row);
                hist[orc_nl_multiplicity_row(t, a)]++; /* padding lookups hit entry 0 of their table like any other */
            }
            /* gates */
            for (uint32_t gi = 0; gi < T->n_gates; gi++) {
                const nl_gate *g = &sp->gates[T->gate0 + gi];
                const nl_term *tm = &sp->terms[T->term0 + g->first_term];
                const size_t row = base + g->row;
                uint64_t acc = g->constant % P;
                int copies = 1;
                for (uint32_t i = 0; i < (uint32_t)g->n_known + g->n_new; i++) {
                    const uint64_t x = TR(g->col + i, row);
                    if (i < g->n_known && !(tm[i].ref >= NL_REF_FREE && tm[i].ref < NL_REF_RC) && x != home(&v, c, s, tm[i].ref)) copies = 0;
                    const uint64_t term = fmul(x, 1ull << (tm[i].code & 0x7F));
                    acc = (tm[i].code & 0x80) ? fadd(acc, P - term) : fadd(acc, term);
                }
                if (!copies) flag(&res, 2, 0x10000 + gi, row);
                if (acc != 0) flag(&res, 7, gi, row);
            }
            /* cells that hold nothing */
            for (uint32_t r = 0; r < T->rows; r++) {
                for (uint32_t col = sp->gate_row_end[T->rowend0 + r]; col < sp->g; col++)
                    if (TR(col, base + r)) { flag(&res, 6, col, base + r); break; }
                if (r == 0 || r > T->lookup_rows)
                    for (uint32_t col = sp->g; col < sp->mult_col; col++)
                        if (TR(col, base + r)) { flag(&res, 6, col, base + r); break; }
            }
        }
    size_t e_begin = 0, e_end = 0; /* ECRecover: the EC section below the queue section has its own checker (it adds its lookups to hist) */
    if (ctype == 7) {
        uint64_t efirst = ~0ull;
        e_begin = orc_ec_first_row(capacity); e_end = orc_ec_used_rows(capacity);
        if (e_end > n_rows) { free(hist); *first_bad = 0; return ~0ull; }
        res.n += orc_ec_check(trace, capacity, n_rows, hist, &efirst);
        if (efirst && efirst < res.first) res.first = efirst;
    }
    const size_t bnd = NL_BOUNDARY_ROW(sp, capacity), brows = NL_BND_ROWS(sp);
    for (size_t row = 0; row < n_rows; row++) {
        if (TR(sp->mult_col, row) != (row < sp->total_table_rows ? hist[row] : 0)) flag(&res, 5, 0, row);
        if (row < bnd || (row >= e_begin && row < e_end)) continue;
        const size_t off = row - bnd;
        for (uint32_t col = ((row >= q_begin && row < q_end) || (row >= c_begin && row < c_end)) ? sp->g : 0; col < sp->mult_col; col++) { /* (the sections' general-purpose cells: orc_nlq_check, orc_nlcf_check) */
            int allowed = 0;
            if (off < 2 * brows) allowed = col < sp->g && (off % brows) * sp->g + col < sp->state;
            else if (off == 2 * brows) allowed = col < 4;
            const uint64_t x = TR(col, row);
            if (!allowed && x) { flag(&res, 6, col, row); break; }
            if (allowed && off < brows && x > 255) { flag(&res, 4, col, row); break; }
        }
        if (off >= brows && off < 2 * brows && capacity)
            for (uint32_t col = 0; col < sp->g; col++) {
                const uint32_t k = (uint32_t)(off - brows) * sp->g + col;
                if (k < sp->state && TR(col, row) != home(&v, capacity, 0, NL_REF_CYC + k)) flag(&res, 4, k, row);
            }
    }
    free(hist);
    if (qd) {
        uint64_t qfirst = ~0ull;
        res.n += orc_nlq_check(ctype, trace, capacity, n_rows, &qfirst);
        if (qfirst < res.first) res.first = qfirst;
    }
    if (nlcf_desc_of(ctype)) {
        uint64_t cfirst = ~0ull;
        const uint64_t nc = orc_nlcf_check(ctype, trace, capacity, n_rows, &cfirst);
        res.n += nc;
        if (nc && cfirst < res.first) res.first = cfirst;
    }
    *first_bad = res.n ? res.first : 0;
    return res.n;
}

/* ---- the four circuits: round records -> the engine's inputs (header bits, free elements, the state before every cycle) ---- */
static int sha_like(const nl_spec *sp, const uint8_t state_in[32], const zkw_sha256_round_record *rounds, uint32_t n_active, uint32_t capacity,
                    const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    if (n_active > capacity) return -1;
    const size_t ST = sp->state; /* the chaining value's 64 nibbles, then zeros (the steps of a compression pass more between them) */
    uint8_t *hdr = calloc(capacity ? capacity : 1, 1), *fr = calloc((size_t)capacity * 128 + 1, 1), *st = calloc((size_t)(capacity + 1) * ST, 1);
    for (int k = 0; k < 64; k++) st[k] = (state_in[k / 2] >> (4 * (k & 1))) & 15; /* nibble i of word j = byte 4j + i/2 of the LE bytes */
    for (uint32_t c = 0; c < capacity; c++) {
        uint8_t *nx = st + (size_t)(c + 1) * ST;
        if (c < n_active) {
            hdr[c] = rounds[c].reset ? 1 : 0;
            for (int b = 0; b < 64; b++) {
                fr[(size_t)c * 128 + 2 * b] = rounds[c].block[b] & 15;
                fr[(size_t)c * 128 + 2 * b + 1] = rounds[c].block[b] >> 4;
            }
            for (int k = 0; k < 64; k++) nx[k] = (rounds[c].state_after[k / 8] >> (4 * (k % 8))) & 15;
        } else {
            hdr[c] = 2;
            memcpy(nx, nx - ST, ST);
        }
    }
    int rc = orc_nl_synthesize(sp, capacity, hdr, fr, st, pi, n_rows, trace);
    free(hdr); free(fr); free(st);
    /* the queue section the bare records imply; a caller that holds the block's queues writes the real one over it (orc_nlq_synthesize) */
    if (rc == 0 && nlq_used_rows(sp, nlq_desc_of(type_of_spec(sp)), capacity) <= n_rows) rc = orc_nlq_standalone(type_of_spec(sp), rounds, n_active, capacity, n_rows, trace);
    if (rc == 0 && nlcf_used_rows(type_of_spec(sp), sp, capacity) <= n_rows) rc = orc_nlcf_standalone(type_of_spec(sp), capacity, n_rows, trace); /* (replaces `pi`) */
    return rc;
}
int orc_sha256_round_synthesize(const uint8_t state_in[32], const zkw_sha256_round_record *rounds, uint32_t n_active, uint32_t capacity,
                                const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    return sha_like(&sc_spec, state_in, rounds, n_active, capacity, pi, n_rows, trace);
}
uint64_t orc_sha256_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) { return orc_nl_check(&sc_spec, trace, capacity, n_rows, first_bad); }
int orc_code_decommitter_round_synthesize(const uint8_t state_in[32], const zkw_sha256_round_record *rounds, uint32_t n_active, uint32_t capacity,
                                          const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    return sha_like(&dc_spec, state_in, rounds, n_active, capacity, pi, n_rows, trace);
}
uint64_t orc_code_decommitter_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) { return orc_nl_check(&dc_spec, trace, capacity, n_rows, first_bad); }

static int keccak_like(const nl_spec *sp, const uint8_t state_in[200], const zkw_keccak_round_record *rounds, uint32_t n_active, uint32_t capacity,
                       const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    if (n_active > capacity) return -1;
    uint8_t *hdr = calloc(capacity ? capacity : 1, 1), *fr = calloc((size_t)capacity * 136 + 1, 1), *st = calloc((size_t)(capacity + 1) * 200, 1);
    memcpy(st, state_in, 200);
    for (uint32_t c = 0; c < capacity; c++) {
        uint8_t *nx = st + (size_t)(c + 1) * 200;
        if (c < n_active) {
            hdr[c] = rounds[c].reset ? 1 : 0;
            memcpy(fr + (size_t)c * 136, rounds[c].block, 136);
            memcpy(nx, rounds[c].state_after, 200);
        } else {
            hdr[c] = 2;
            memcpy(nx, nx - 200, 200);
        }
    }
    int rc = orc_nl_synthesize(sp, capacity, hdr, fr, st, pi, n_rows, trace);
    free(hdr); free(fr); free(st);
    if (rc == 0 && nlq_used_rows(sp, nlq_desc_of(type_of_spec(sp)), capacity) <= n_rows) rc = orc_nlq_standalone_keccak(type_of_spec(sp), rounds, n_active, capacity, n_rows, trace);
    if (rc == 0 && nlcf_used_rows(type_of_spec(sp), sp, capacity) <= n_rows) rc = orc_nlcf_standalone(type_of_spec(sp), capacity, n_rows, trace); /* (replaces `pi`) */
    return rc;
}
int orc_keccak_round_synthesize(const uint8_t state_in[200], const zkw_keccak_round_record *rounds, uint32_t n_active, uint32_t capacity,
                                const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    return keccak_like(&kc_spec, state_in, rounds, n_active, capacity, pi, n_rows, trace);
}
uint64_t orc_keccak_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) { return orc_nl_check(&kc_spec, trace, capacity, n_rows, first_bad); }
int orc_linear_hasher_round_synthesize(const uint8_t state_in[200], const zkw_keccak_round_record *rounds, uint32_t n_active, uint32_t capacity,
                                       const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    return keccak_like(&lh_spec, state_in, rounds, n_active, capacity, pi, n_rows, trace);
}
uint64_t orc_linear_hasher_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) { return orc_nl_check(&lh_spec, trace, capacity, n_rows, first_bad); }

/* geometry of the four layouts, for the tests: {cols, general, lookup width, lookups per row, total table rows, rows per cycle} */
uint32_t orc_nl_state(int circuit_type) { const nl_spec *sp = orc_nl_spec(circuit_type); return sp ? sp->state : 0; }
void orc_nl_geometry(int circuit_type, uint32_t out[6]) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    memset(out, 0, 6 * sizeof(uint32_t));
    if (!sp) return;
    out[0] = sp->cols; out[1] = sp->g; out[2] = sp->w; out[3] = sp->r; out[4] = sp->total_table_rows; out[5] = sp->rows_per_cycle;
}

/* The cycles of the L1MessagesHasher circuit (type 13): Keccak-256 over the concatenated 88-byte messages, pad10*1, one record per
   136-byte block (compute_linear_keccak256, data_hasher_and_merklizer.rs:8-67; wrapper geometry base_layer/linear_hasher.rs:28-138).
   records: at most n * 88 / 136 + 1 of them; returns their number. */
void orc_keccak_f1600(uint64_t a[25]);
size_t orc_linear_hasher_rounds(const zkw_log_query *q, size_t n, zkw_keccak_round_record *records) {
    const size_t len = n * 88, n_rounds = len / 136 + 1;
    uint8_t *buf = calloc(n_rounds * 136, 1);
    for (size_t i = 0; i < n; i++) orc_serialize_l1_message(q + i, buf + 88 * i);
    buf[len] ^= 0x01; /* pad10*1 */
    buf[n_rounds * 136 - 1] ^= 0x80;
    uint64_t st[25] = {0};
    for (size_t r = 0; r < n_rounds; r++) {
        zkw_keccak_round_record *rec = records + r;
        memset(rec, 0, sizeof *rec);
        memcpy(rec->block, buf + 136 * r, 136);
        rec->reset = r == 0;
        for (int k = 0; k < 17; k++) {
            uint64_t lane = 0;
            for (int b = 0; b < 8; b++) lane |= (uint64_t)rec->block[8 * k + b] << (8 * b);
            st[k] ^= lane;
        }
        orc_keccak_f1600(st);
        for (int l = 0; l < 25; l++)
            for (int b = 0; b < 8; b++) rec->state_after[8 * l + b] = (uint8_t)(st[l] >> (8 * b));
    }
    free(buf);
    return n_rounds;
}

void orc_nl_slots_per_cycle(int circuit_type, uint32_t *out) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    *out = 0;
    if (!sp) return;
    for (uint32_t s = 0; s < sp->steps_per_cycle; s++) *out += sp->step_types[sp->cycle[s].type].n_ops;
}

/* ---- StorageApplication (type 10): the Merkle walks of an instance's tree queries as cycles of the Blake2s netlist
   (tools/gen_storage_application_circuit.py). A read is one walk, a write two — the old leaf's path, then the new leaf's
   (src/witness/individual_circuits/storage_application.rs:141-153, 216-247); a walk = leaf hash (tree/mod.rs:322-329) + 256
   node hashes (:394-402) with the sibling on the side the key's bit names (:187-217). items: the instance's queries; keys /
   paths / read_indexes: their rows of orc_storage_application_build's outputs; next_enumeration_index: before the instance
   (a first write takes the next one, tree/mod.rs:305-313). capacity in walks. */
void orc_blake2s256(const uint8_t *msg, size_t len, uint8_t out[32]);
int orc_storage_application_synthesize(const zkw_log_query *items, size_t n_items, const uint8_t *keys, const uint8_t *paths,
                                       const uint64_t *read_indexes, uint64_t next_enumeration_index, uint32_t capacity,
                                       const uint64_t pi[4], size_t n_rows, uint64_t *trace, const uint8_t *idle_root /* [32] or NULL: what an instance
                                       without walks carries (its current root: the closed-form section ties the root handed on to the last state) */) {
    const uint32_t cycles = capacity * SA_CYCLES_PER_WALK;
    uint8_t *hdr = calloc(cycles, 1), *fr = calloc((size_t)cycles * SA_FREE_PER_CYCLE + 1, 1), *st = calloc((size_t)(cycles + 1) * SA_STATE, 1);
    uint32_t c = 0;
    int rc = 0;
    if (n_items == 0 && idle_root) memcpy(st, idle_root, 32);
    for (size_t i = 0; i < n_items && rc == 0; i++) {
        const zkw_log_query *q = &items[i];
        const uint8_t *key = keys + 32 * i;
        uint64_t write_index = read_indexes[i];
        if (q->rw_flag && write_index == 0) write_index = next_enumeration_index++;
        for (int phase = 0; phase < (q->rw_flag ? 2 : 1); phase++) {
            if (c + SA_CYCLES_PER_WALK > cycles) { rc = -10; break; } /* more walks than the capacity */
            const uint64_t index = phase ? write_index : read_indexes[i];
            const uint32_t *val = phase ? q->written_value : q->read_value;
            uint8_t msg[64] = {0}, cur[32], key33[33] = {0};
            for (int b = 0; b < 8; b++) msg[b] = (uint8_t)(index >> (8 * (7 - b)));
            for (int b = 0; b < 32; b++) msg[8 + b] = (uint8_t)(val[(31 - b) / 4] >> (8 * ((31 - b) % 4))); /* big-endian U256 */
            for (int b = 0; b < 257; b++) /* key << 1 */
                if (b && ((key[(b - 1) / 8] >> ((b - 1) % 8)) & 1)) key33[b / 8] |= (uint8_t)(1u << (b % 8));
            /* the leaf cycle */
            hdr[c] = 1;
            memcpy(fr + (size_t)c * SA_FREE_PER_CYCLE + SA_FREE_X, msg, 32);
            memcpy(fr + (size_t)c * SA_FREE_PER_CYCLE + SA_FREE_Y, msg + 32, 32);
            memcpy(fr + (size_t)c * SA_FREE_PER_CYCLE + SA_FREE_KEY, key33, 33);
            orc_blake2s256(msg, 40, cur);
            for (int level = 0; level <= 256; level++) {
                /* state after cycle c: the running hash, the key shifted once more */
                uint8_t *nx = st + (size_t)(c + 1) * SA_STATE;
                memcpy(nx, cur, 32);
                for (int b = 0; b < 32; b++) key33[b] = (uint8_t)((key33[b] >> 1) | (key33[b + 1] << 7));
                key33[32] >>= 1;
                memcpy(nx + 32, key33, 33);
                c++;
                if (level == 256) break;
                const uint8_t *sib = paths + ((size_t)i * 256 + level) * 32;
                const int right = (key[level / 8] >> (level % 8)) & 1;
                uint8_t buf[64];
                memcpy(buf, right ? sib : cur, 32);
                memcpy(buf + 32, right ? cur : sib, 32);
                orc_blake2s256(buf, 64, cur);
                hdr[c] = 0;
                memcpy(fr + (size_t)c * SA_FREE_PER_CYCLE + SA_FREE_Y, sib, 32);
            }
        }
    }
    for (; c < cycles; c++) { /* padding cycles carry the hash; the key is zero by now */
        hdr[c] = 2;
        memcpy(st + (size_t)(c + 1) * SA_STATE, st + (size_t)c * SA_STATE, SA_STATE);
    }
    if (rc == 0) rc = orc_nl_synthesize(&sa_spec, cycles, hdr, fr, st, pi, n_rows, trace);
    free(hdr); free(fr); free(st);
    return rc;
}
uint64_t orc_storage_application_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    return orc_nl_check(&sa_spec, trace, capacity * SA_CYCLES_PER_WALK, n_rows, first_bad);
}

/* for the tests: {row within the cycle, column} of the cell that holds FREE element `free_index` */
int orc_nl_free_home_of(int circuit_type, uint32_t free_index, uint32_t out[2]) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    return sp ? orc_nl_free_home(sp, free_index, &out[0], &out[1]) : -1;
}
