/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * The ECRecover base-layer circuit (type 7): fill and satisfiability check of a whole trace on the CPU, sequential and obvious.
 * Geometry and tables: circuit_definitions/src/circuit_definitions/base_layer/ecrecover.rs:30-41 (80 + 3 x 16 columns), :138-176
 * (Xor8, And8, 8 x 32 FixedBaseMul, ByteSplit<1..4> = 197 632 rows = vk_7.json's total_tables_len); requests and their 4 reads +
 * 2 writes: src/witness/individual_circuits/ecrecover.rs:143-178. The circuit body is in the absent era-zkevm_circuits crate:
 * PARITY UNPINNED at the placement level (tools/gen_ecrecover_circuit.py states the layout and the statement); what IS pinned are the
 * results — recovered addresses against public secp256k1 / Ethereum vectors (tests/test_oracle_ecrecover_circuit.py).
 *
 * A trace = the EK byte netlist (one Keccak-f per cycle over the public key; oracle/netlist_circuit.c), the queue section (pop, 4
 * reads, 2 writes; oracle/netlist_queue.c) and the EC section, evaluated and checked by oracle/ecrecover_eval.c — the oracle's own
 * code over the spec's FORMAT (include/zkw_ecrecover_layout.h); the library's evaluator (include/zkw_ecrecover.h) is not compiled here.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "../include/zkw_ecrecover_circuit_spec.h"
#include "../include/zkw_netlist_queue.h"
#include "../include/zkw_ecrecover_layout.h"

/* oracle/ecrecover_eval.c */
void orc_ec_build_fixed_own(uint32_t *out);
uint32_t orc_ec_eval_cycle_own(const ec_spec *S, const uint8_t *in, uint64_t *tape);
int orc_ec_check_item_own(const ec_spec *S, const uint32_t *w, const uint64_t *trace, size_t n_rows, size_t row, uint32_t inst);

EC_DEFINE_SPEC(ecs);
static uint32_t *g_fixed = NULL;
static ec_spec g_spec;

static pthread_once_t g_spec_once = PTHREAD_ONCE_INIT;
static void spec_init(void) { /* once per process, whichever thread comes first (the FixedBaseMul tables take a moment) */
    g_fixed = malloc(sizeof(uint32_t) * EC_FIXED_WORDS);
    orc_ec_build_fixed_own(g_fixed);
    const ec_spec s = {ecs_types, ecs_runs, ecs_items, ecs_item_index, ecs_cells, ecs_homes, ecs_outs, ecs_rowtab, ecs_globs, ecs_bigs, ecs_in_home, ecs_key_byte, g_fixed};
    g_spec = s;
}
const ec_spec *orc_ec_spec(void) {
    pthread_once(&g_spec_once, spec_init);
    return &g_spec;
}

#define TR(c, r) trace[(size_t)(c) * n_rows + (size_t)(r)]

static const nl_spec *ek(void) { return orc_nl_spec(7); }
/* first row of the EC section of a trace of `capacity` cycles */
uint64_t orc_ec_first_row(uint32_t capacity) { return nlq_used_rows(ek(), nlq_desc_of(7), capacity); }
uint64_t orc_ec_used_rows(uint32_t capacity) { return orc_ec_first_row(capacity) + (uint64_t)capacity * EC_ROWS_PER_CYCLE; }
/* {first row, rows per cycle, rows used, tape values per cycle, segment types, runs} */
void orc_ec_geometry(uint32_t capacity, uint64_t out[8]) {
    memset(out, 0, 8 * sizeof(uint64_t));
    out[0] = orc_ec_first_row(capacity); out[1] = EC_ROWS_PER_CYCLE; out[2] = orc_ec_used_rows(capacity); out[3] = EC_TAPE_PER_CYCLE;
    out[4] = EC_NUM_TYPES; out[5] = EC_NUM_RUNS;
}
uint32_t orc_ec_eval_cycle(const uint8_t in[128], uint64_t *tape) { return orc_ec_eval_cycle_own(orc_ec_spec(), in, tape); }
/* (ok, mask, the 64 key bytes as the netlist hashes them) of an evaluated tape */
void orc_ec_outputs(const uint64_t *tape, uint8_t out[66]) {
    const ec_spec *S = orc_ec_spec();
    out[0] = (uint8_t)tape[S->globs[EC_GL_OK]];
    out[1] = (uint8_t)tape[S->globs[EC_GL_MASK]];
    for (int k = 0; k < 64; k++) out[2 + k] = (uint8_t)tape[S->runs[EC_NUM_RUNS - 1].tape0 + S->key_byte[k]];
}
/* (row within the cycle's EC rows, column) of a cell, for the tests: what = 0 home of global k, 1 home of input byte k, 2 home of key byte k,
   3 the MUL row of item `k` of DAA instance 0 (column 0) */
int orc_ec_cell(int what, uint32_t k, uint32_t out[2]) {
    const ec_spec *S = orc_ec_spec();
    if (what == 0) { ec_home_of_tape(S, S->globs[k], &out[0], &out[1]); return 0; }
    if (what == 1) { out[0] = S->in_home[k] >> 8; out[1] = S->in_home[k] & 0xFF; return 0; }
    if (what == 2) { ec_home_of_tape(S, S->runs[EC_NUM_RUNS - 1].tape0 + S->key_byte[k], &out[0], &out[1]); return 0; }
    const ec_seg_type *T = &S->types[1];
    const uint32_t *w = S->items + T->item0;
    uint32_t seen = 0;
    for (uint32_t n = 0; n < T->n_items; n++, w += ec_item_words(w))
        if ((w[0] & 15) == EC_I_MUL && seen++ == k) { out[0] = S->runs[1].row0 + ((w[0] >> 4) & 0xFFF); out[1] = 0; return 0; }
    return -1;
}

void orc_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]);

/* The whole trace of one instance but for its queue section (orc_nlq_synthesize writes that over it): inputs[capacity][128] = the value
   bytes (little end first) of the four reads of every cycle, zeros for the idle ones. Returns 0; -1 bad arguments / rows; -2 - cycle
   when a cycle's inputs have no witness (the incomplete addition met x1 == x2). */
int orc_ecrecover_synthesize(const uint8_t *inputs, uint32_t n_active, uint32_t capacity, const uint64_t pi[4], size_t n_rows, uint64_t *trace) {
    const nl_spec *sp = ek();
    const ec_spec *S = orc_ec_spec();
    if (!sp || n_active > capacity || orc_ec_used_rows(capacity) > n_rows) return -1;
    uint64_t *tapes = malloc((size_t)capacity * EC_TAPE_PER_CYCLE * sizeof(uint64_t));
    uint8_t *hdr = calloc(capacity ? capacity : 1, 1), *fr = calloc((size_t)capacity * EK_FREE_PER_CYCLE + 1, 1), *st = calloc((size_t)(capacity + 1) * 200, 1);
    int rc = 0;
    for (uint32_t c = 0; c < capacity && rc == 0; c++) {
        uint64_t *tape = tapes + (size_t)c * EC_TAPE_PER_CYCLE;
        if (orc_ec_eval_cycle_own(S, inputs + (size_t)c * 128, tape)) { rc = -2 - (int)c; break; }
        uint8_t o[66], dig[32];
        orc_ec_outputs(tape, o);
        uint8_t *f = fr + (size_t)c * EK_FREE_PER_CYCLE;
        memcpy(f, o + 2, 64);
        f[EK_FREE_MASK] = o[1];
        f[EK_FREE_OK] = o[0];
        hdr[c] = c < n_active ? 0 : 2; /* idle: the queue operations are disabled (the EC section runs on its zero inputs all the same) */
        orc_keccak256(o + 2, 64, dig);
        uint8_t *nx = st + (size_t)(c + 1) * 200; /* the state after the cycle: the masked address, `ok` */
        for (int k = 12; k < 32; k++) nx[k] = dig[k] & o[1];
        nx[EK_STATE_OK] = o[0];
    }
    if (rc == 0) rc = orc_nl_synthesize(sp, capacity, hdr, fr, st, pi, n_rows, trace);
    if (rc == 0) {
        const size_t e0 = orc_ec_first_row(capacity);
        for (uint32_t c = 0; c < capacity; c++) {
            const uint64_t *tape = tapes + (size_t)c * EC_TAPE_PER_CYCLE;
            const uint8_t *in = inputs + (size_t)c * 128;
            for (uint32_t r = 0; r < EC_NUM_RUNS; r++) {
                const ec_seg_type *T = &S->types[S->runs[r].type];
                for (uint32_t j = 0; j < S->runs[r].count; j++)
                    for (uint32_t row = 0; row < T->n_rows; row++) {
                        const size_t tr = e0 + (size_t)c * EC_ROWS_PER_CYCLE + S->runs[r].row0 + (size_t)j * T->n_rows + row;
                        const uint32_t *cells = S->cells + T->cell0 + (size_t)row * EC_ROW_CELLS;
                        for (uint32_t col = 0; col < EC_ROW_CELLS; col++) TR(col, tr) = ec_cell_value(S, tape, in, r, j, cells[col]);
                        const uint32_t tb = ec_row_table(S, r, j, row);
                        if (tb)
                            for (uint32_t slot = 0; slot < EC_R; slot++)
                                TR(sp->mult_col, ec_table_key(tb, TR(EC_G + EC_W * slot, tr), TR(EC_G + EC_W * slot + 1, tr)))++;
                    }
            }
        }
    }
    free(tapes); free(hdr); free(fr); free(st);
    return rc;
}

static void flag(uint64_t *n, uint64_t *first, uint64_t kind, uint64_t idx, uint64_t row) {
    const uint64_t code = (kind << 56) | (idx << 32) | row;
    (*n)++;
    if (code < *first) *first = code;
}

/* The EC section from its cells alone: every item's relation (kind 7; lookups kind 1), every copy (kind 2: a cell against the home cell
   of the value it names, constants, the input bytes against the read queries' value bytes in the queue section, the netlist's FREE
   elements against the key bytes / mask / ok), empty cells (kind 6); hist[] += the section's lookups (the caller compares the
   multiplicity column). Violation codes like orc_nl_check's. */
uint64_t orc_ec_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint32_t *hist, uint64_t *first_bad) {
    const nl_spec *sp = ek();
    const ec_spec *S = orc_ec_spec();
    const nlq_desc *qd = nlq_desc_of(7);
    uint64_t n = 0, first = ~0ull;
    const size_t e0 = orc_ec_first_row(capacity);
    for (uint32_t c = 0; c < capacity; c++) {
        const size_t cyc0 = e0 + (size_t)c * EC_ROWS_PER_CYCLE;
        for (uint32_t r = 0; r < EC_NUM_RUNS; r++) {
            const ec_seg_type *T = &S->types[S->runs[r].type];
            for (uint32_t j = 0; j < S->runs[r].count; j++) {
                const size_t seg0 = cyc0 + S->runs[r].row0 + (size_t)j * T->n_rows;
                const uint32_t *w = S->items + T->item0;
                for (uint32_t i = 0; i < T->n_items; i++, w += ec_item_words(w)) {
                    const size_t irow = seg0 + ((w[0] >> 4) & 0xFFF);
                    if (orc_ec_check_item_own(S, w, trace, n_rows, irow, j)) flag(&n, &first, (w[0] & 15) == EC_I_LOOKUP ? 1 : 7, i, irow);
                }
                uint32_t prun, pinst;
                ec_prev_segment(S, r, j, &prun, &pinst);
                const uint32_t base = S->runs[r].tape0 + j * T->n_tape, pbase = S->runs[prun].tape0 + pinst * S->types[S->runs[prun].type].n_tape;
                for (uint32_t row = 0; row < T->n_rows; row++) {
                    const uint32_t *cells = S->cells + T->cell0 + (size_t)row * EC_ROW_CELLS;
                    const size_t tr = seg0 + row;
                    for (uint32_t col = 0; col < EC_ROW_CELLS; col++) {
                        const uint32_t ref = cells[col];
                        const uint64_t x = TR(col, tr);
                        if (ref == EC_NONE) { if (x) flag(&n, &first, 6, col, tr); continue; }
                        const uint32_t t = ec_ref_tape(S, ref, base, pbase, S->runs[prun].type, j);
                        if (t != EC_NONE) {
                            uint32_t hr, hc;
                            ec_home_of_tape(S, t, &hr, &hc);
                            if (x != TR(hc, cyc0 + hr)) flag(&n, &first, 2, col, tr);
                        } else if ((ref >> 28) == EC_K_IN) {
                            const uint32_t k = ref & 0xFFFF, h = S->in_home[k];
                            if (r == 0 && row == (h >> 8) && col == (h & 0xFF)) { /* the home: a copy of value byte k % 32 of read k / 32 */
                                const uint32_t op = 1 + k / 32, cell = NLQ_MEM_NIBBLE0 + k % 32;
                                const size_t qrow = NLQ_ROW(sp, capacity, nlq_op_row0(qd, sp->g, op) + cell / sp->g, c);
                                if (x != TR(cell % sp->g, qrow)) flag(&n, &first, 2, 0x1000 + k, tr);
                            } else if (x != TR(h & 0xFF, cyc0 + (h >> 8))) flag(&n, &first, 2, col, tr);
                        } else if (x != ec_ref_const(S, ref, NULL)) flag(&n, &first, 2, col, tr);
                    }
                    const uint32_t tb = ec_row_table(S, r, j, row);
                    if (tb)
                        for (uint32_t slot = 0; slot < EC_R; slot++) {
                            const uint64_t a = TR(EC_G + EC_W * slot, tr), b = TR(EC_G + EC_W * slot + 1, tr);
                            if (a < 256 && (tb != EC_T_XOR8 || b < 256)) hist[ec_table_key(tb, a, b)]++; /* (a bad key: flagged by its item) */
                        }
                }
            }
        }
        /* the netlist's FREE elements of the cycle are copies of EC values: the key bytes, the mask, ok */
        for (uint32_t k = 0; k < EK_FREE_PER_CYCLE; k++) {
            uint32_t frow, fcol, hr, hc;
            if (orc_nl_free_home(sp, k, &frow, &fcol) != 0) continue;
            const uint32_t t = k < 64 ? S->runs[EC_NUM_RUNS - 1].tape0 + S->key_byte[k] : S->globs[k == EK_FREE_MASK ? EC_GL_MASK : EC_GL_OK];
            ec_home_of_tape(S, t, &hr, &hc);
            const size_t nrow = (size_t)c * sp->rows_per_cycle + frow;
            if (TR(fcol, nrow) != TR(hc, cyc0 + hr)) flag(&n, &first, 2, 0x2000 + k, nrow);
        }
    }
    *first_bad = n ? first : 0;
    return n;
}

uint64_t orc_ecrecover_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) { return orc_nl_check(ek(), trace, capacity, n_rows, first_bad); }

/* the queue feed of an instance: cycle c < n_active is request first_round + c with its six memory queries, all operations enabled;
   an idle cycle names the items after the last one (null records). feed: [capacity][7] */
void orc_ecrecover_queue_feed(size_t first_round, uint32_t n_active, uint32_t capacity, nlq_feed *feed) {
    for (uint32_t c = 0; c < capacity; c++) {
        const int on = c < n_active;
        const size_t r = first_round + (on ? c : n_active);
        nlq_feed *f = feed + (size_t)c * 7;
        f[0] = (nlq_feed){(uint32_t)on, (uint32_t)r, 0};
        for (uint32_t k = 0; k < 6; k++) f[1 + k] = (nlq_feed){(uint32_t)on, (uint32_t)(6 * r + (on ? k : 0)), 0};
    }
}
