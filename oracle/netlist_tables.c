/* TEST INFRASTRUCTURE — netlist_tables.c: the CONTENTS of the lookup tables of the netlist circuits, enumerated row by row by the
 * oracle itself. The library evaluates a table by a word formula (era_zkevm_test_harness_amd/csrc/netlist_eval.cuh); this file shares
 * no code with it ("give the oracle its own evaluator"): every table is materialised the way boojum's create_*_table
 * functions build theirs — a loop over all input tuples in row order, the outputs from the DEFINITION of the function bit by bit
 * (exclusive or = parity of the bits, choose = "x ? y : z" per bit, majority = at least two of three per bit, a split = quotient and
 * remainder by 2^k) — and a lookup is a row access. The tables' shapes (inputs, their widths, outputs, rows) come from the spec.
 * boojum (era-boojum, branch main) is absent from /root/reference: the table NAMES and row counts are the reference's
 * (circuit_definitions/src/circuit_definitions/base_layer/sha256_round_function.rs:120-135, keccak256_round_function.rs:118-131:
 * 12 320 and 132 096 = total_tables_len of vk_6 / vk_5), their contents are the standard functions of those names.
 *
 * Row order: input 0 is the fastest-running coordinate (row = in_0 + in_1 * 2^bits + in_2 * 2^(2 bits)). */
#include "oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_table { uint8_t fn, param, n_in, in_bits, n_out; uint32_t rows; uint8_t *out; /* [rows][3] */ } orc_table;
static orc_table g_tabs[64];
static int g_n_tabs;

static unsigned bit_of(unsigned v, int i) { return (v >> i) & 1u; }

static void fill(orc_table *t) {
    t->out = (uint8_t *)calloc((size_t)t->rows * 3, 1);
    const unsigned mask = (1u << t->in_bits) - 1;
    for (uint32_t row = 0; row < t->rows; row++) {
        unsigned in[3] = {0, 0, 0};
        for (int i = 0; i < t->n_in; i++) in[i] = (row >> (t->in_bits * i)) & mask;
        unsigned o[3] = {0, 0, 0};
        switch (t->fn) {
            case NL_FN_XOR8: case NL_FN_TRIXOR4: /* parity of the inputs' bits */
                for (int b = 0; b < t->in_bits; b++) o[0] |= ((bit_of(in[0], b) + bit_of(in[1], b) + bit_of(in[2], b)) & 1u) << b;
                break;
            case NL_FN_AND8: /* both bits set */
                for (int b = 0; b < t->in_bits; b++) o[0] |= (bit_of(in[0], b) + bit_of(in[1], b) == 2 ? 1u : 0u) << b;
                break;
            case NL_FN_CH4: /* choose: bit of in_1 where in_0 has a one, bit of in_2 where it has a zero */
                for (int b = 0; b < t->in_bits; b++) o[0] |= (bit_of(in[0], b) ? bit_of(in[1], b) : bit_of(in[2], b)) << b;
                break;
            case NL_FN_MAJ4: /* majority of three */
                for (int b = 0; b < t->in_bits; b++) o[0] |= (bit_of(in[0], b) + bit_of(in[1], b) + bit_of(in[2], b) >= 2 ? 1u : 0u) << b;
                break;
            case NL_FN_BYTESPLIT: /* x = low + 2^k high */
                o[0] = in[0] % (1u << t->param); o[1] = in[0] / (1u << t->param);
                break;
            case NL_FN_SPLIT4: /* a 4-bit chunk: low k bits, high 4 - k bits, and the chunk rotated right by k (= low * 2^(4-k) + high) */
                o[0] = in[0] % (1u << t->param); o[1] = in[0] / (1u << t->param); o[2] = o[0] * (1u << (4 - t->param)) + o[1];
                break;
            default: break;
        }
        for (int i = 0; i < 3; i++) t->out[(size_t)row * 3 + i] = (uint8_t)o[i];
    }
}

static pthread_mutex_t g_tabs_lock = PTHREAD_MUTEX_INITIALIZER;
static const orc_table *table_for_locked(const nl_table *d);
/* the cache is filled on first use; instances are synthesized from several threads (bench.py's CPU leg), hence the lock */
static const orc_table *table_for(const nl_table *d) {
    pthread_mutex_lock(&g_tabs_lock);
    const orc_table *t = table_for_locked(d);
    pthread_mutex_unlock(&g_tabs_lock);
    return t;
}
static const orc_table *table_for_locked(const nl_table *d) {
    for (int i = 0; i < g_n_tabs; i++)
        if (g_tabs[i].fn == d->fn && g_tabs[i].param == d->param && g_tabs[i].n_in == d->n_in && g_tabs[i].in_bits == d->in_bits && g_tabs[i].rows == d->rows) return &g_tabs[i];
    if (g_n_tabs == 64) abort();
    orc_table *t = &g_tabs[g_n_tabs];
    t->fn = d->fn; t->param = d->param; t->n_in = d->n_in; t->in_bits = d->in_bits; t->n_out = d->n_out; t->rows = d->rows;
    fill(t);
    g_n_tabs++;
    return t;
}

/* row of `d` (inside the table, without the stack offset) that inputs a[] address; rows of a table with fewer than 2^(bits n_in) rows do not exist */
static uint32_t row_of(const nl_table *d, const uint32_t a[3]) {
    uint32_t row = 0, w = 1;
    for (int i = 0; i < d->n_in; i++) { row += a[i] * w; w <<= d->in_bits; }
    return row;
}

void orc_nl_lookup(const nl_table *d, const uint32_t a[3], uint32_t out[3]) {
    const orc_table *t = table_for(d);
    const uint32_t row = row_of(d, a);
    out[0] = out[1] = out[2] = 0;
    if (row >= t->rows) return;
    for (int i = 0; i < 3; i++) out[i] = t->out[(size_t)row * 3 + i];
}

uint32_t orc_nl_multiplicity_row(const nl_table *d, const uint32_t a[3]) { return d->offset + row_of(d, a); }
