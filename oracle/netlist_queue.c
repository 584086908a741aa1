/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * The QUEUE SECTION of the netlist circuits (format and reference citations: include/zkw_netlist_queue.h): request-queue pops and
 * memory-queue pushes of Sha256RoundFunction (6) and CodeDecommitter (3) as Poseidon2 rows below the hash netlist, tied to it by
 * copy constraints. Queue arithmetic = circuit_encodings/src/lib.rs:180-203 (QueueSimulator), :391-429 (FullWidthQueueSimulator),
 * encodings = memory_query.rs:24-118, log_query.rs:102-396, decommittment_request.rs:9-74; the gadget placement is in the absent
 * era-zkevm_circuits crate: PARITY UNPINNED at the placement level. Sequential and obvious: one operation after the other.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "../include/zkw_netlist_queue.h"
#pragma GCC diagnostic ignored "-Wmissing-field-initializers" /* (nlq_feed){en, idx}: aux = 0 */

#define P ZKW_GOLDILOCKS_P
#define TR(c, r) trace[(size_t)(c) * n_rows + (size_t)(r)]

const nl_spec *orc_nl_spec(int circuit_type);
uint64_t orc_nl_home(const nl_spec *sp, const uint64_t *trace, size_t n_rows, uint32_t capacity, uint32_t c, uint32_t s, uint32_t ref);
int orc_nl_free_home(const nl_spec *sp, uint32_t free_index, uint32_t *row_in_cycle, uint32_t *col);

static uint64_t fmul_pow2(uint64_t x, uint32_t shift) { return orc_gl_mul(x % P, 1ull << shift); }

/* cell k of a block whose first row (within the cycle's operations) is r0 */
#define QCELL(r0, k) TR((k) % G, NLQ_ROW(sp, capacity, (r0) + (k) / G, c))

/* the netlist cell a linked component copies; *has = 0 when the cell has no link in this cycle */
static uint64_t linked_value(const nl_spec *sp, const uint64_t *trace, size_t n_rows, uint32_t capacity, uint32_t c, const nlq_op *op, uint32_t cell, int *has) {
    uint32_t cyc = 0, ref = 0;
    *has = nlq_link_target(op, c, capacity, cell, &cyc, &ref);
    if (!*has) return 0;
    if (ref >= NL_REF_CYC && ref < NL_REF_FREE) return orc_nl_home(sp, trace, n_rows, capacity, cyc, 0, ref);
    uint32_t row = 0, col = 0;
    if (orc_nl_free_home(sp, ref - NL_REF_FREE, &row, &col)) { *has = 0; return 0; }
    return TR(col, (size_t)cyc * sp->rows_per_cycle + row);
}

static void state_before(const orc_nlq_queue *q, uint32_t width, uint64_t idx, uint64_t out[12]) {
    memset(out, 0, 12 * sizeof(uint64_t));
    if (idx == 0) { if (q->init) memcpy(out, q->init, width * sizeof(uint64_t)); }
    else memcpy(out, q->states + (idx - 1) * width, width * sizeof(uint64_t));
}

/* The section of one instance, written below a netlist trace that orc_nl_synthesize has already filled (the linked cells are read
   from it). feed: [capacity][n_ops]. Returns 0, or < 0 when the section does not fit. */
int orc_nlq_synthesize(int circuit_type, uint32_t capacity, const nlq_feed *feed, const orc_nlq_queue *queues, size_t n_rows, uint64_t *trace) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    const nlq_desc *d = nlq_desc_of(circuit_type);
    if (!sp || !d || capacity == 0 || nlq_used_rows(sp, d, capacity) > n_rows) return -1;
    const uint32_t G = sp->g;
    const size_t q0 = NLQ_BASE(sp, capacity);
    for (uint32_t c = 0; c < capacity; c++)
        for (uint32_t j = 0; j < d->n_ops; j++) {
            const nlq_op *op = &d->ops[j];
            const nlq_feed f = feed[(size_t)c * d->n_ops + j];
            const orc_nlq_queue *Q = &queues[op->queue];
            const uint32_t w = nlq_kind_width(op->kind), ncomp = nlq_item_comps(op->item), nenc = nlq_item_enc(op->item), r0 = nlq_op_row0(d, G, j);
            if (f.en && f.idx >= Q->n_items) return -2;
            const void *rec = f.en ? (const char *)Q->items + (size_t)f.idx * nlq_item_bytes(op->item) : NULL;
            uint64_t cells[160], old[12], out[12];
            cells[0] = f.en ? 1 : 0;
            for (uint32_t k = 1; k < ncomp; k++) {
                int has = 0;
                const uint64_t v = linked_value(sp, trace, n_rows, capacity, c, op, k, &has);
                cells[k] = has ? v : nlq_item_component(op->item, rec, k);
            }
            for (uint32_t r = 0; r < nlq_aux_n(op->item); r++) { /* a limb is the recomposition of its byte cells */
                uint64_t acc = 0;
                for (uint32_t i = 0; i < nlq_aux_n_terms(op->item, r); i++) {
                    const nlq_term tm = nlq_aux_term(op->item, r, i);
                    acc = orc_gl_add(acc, fmul_pow2(cells[tm.cell], tm.shift));
                }
                cells[nlq_aux_result(op->item, r)] = acc;
            }
            for (uint32_t e = 0; e < nenc; e++) {
                uint64_t acc = 0;
                for (uint32_t i = 0; i < nlq_enc_n_terms(op->item, e); i++) {
                    const nlq_term tm = nlq_enc_term(op->item, e, i);
                    acc = orc_gl_add(acc, fmul_pow2(cells[tm.cell], tm.shift));
                }
                cells[ncomp + e] = acc;
            }
            state_before(Q, w, f.idx, old);
            const uint64_t *enc = cells + ncomp;
            for (uint32_t p = 0; p < nlq_kind_perms(op->kind); p++) {
                uint64_t in[12], slots[NLQ_P2_CELLS];
                if (op->kind != NLQ_POP4) { memcpy(in, enc, 64); memcpy(in + 8, old + 8, 32); }
                else if (p == 0) { memcpy(in, enc, 64); memset(in + 8, 0, 32); }
                else if (p == 1) { memcpy(in, enc + 8, 64); memcpy(in + 8, out + 8, 32); }
                else { memcpy(in, enc + 16, 32); memcpy(in + 4, old, 32); memcpy(in + 8, out + 8, 32); }
                orc_poseidon2_flattened(in, slots);
                memcpy(out, slots + NLQ_P2_CELLS - 12, 96);
                const uint32_t pr0 = nlq_p2_row0(d, G, j, p);
                for (uint32_t k = 0; k < NLQ_P2_CELLS; k++) QCELL(pr0, k) = slots[k];
            }
            for (uint32_t k = 0; k < w; k++) {
                cells[ncomp + nenc + k] = old[k];
                cells[ncomp + nenc + w + k] = f.en ? out[k] : old[k];
            }
            for (uint32_t r = 0; r < op->extra; r++) { /* registers: a limb of the request the cycle is working on (the last one operation 0 popped) */
                const nlq_feed f0 = feed[(size_t)c * d->n_ops];
                const orc_nlq_queue *Q0 = &queues[d->ops[0].queue];
                const int64_t cur = (int64_t)f0.idx - (f0.en ? 0 : 1);
                const void *rec0 = cur >= 0 && (uint64_t)cur < Q0->n_items ? (const char *)Q0->items + (size_t)cur * nlq_item_bytes(d->ops[0].item) : NULL;
                uint64_t v = 0;
                if (r >= 2) v = f.aux; /* a counter the feed supplies */
                else for (uint32_t i = 0; i < 4; i++) v |= nlq_item_component(d->ops[0].item, rec0, op->reg_cell[r] + i) << (8 * i);
                cells[ncomp + nenc + 2 * w + r] = v;
            }
            for (uint32_t k = 0; k < nlq_enc_cells(op); k++) QCELL(r0, k) = cells[k];
        }
    /* QBND: the queue states before cycle 0 and after the last cycle */
    for (uint32_t q = 0; q < d->n_queues; q++) {
        uint32_t first = d->n_ops, last = 0;
        for (uint32_t j = 0; j < d->n_ops; j++)
            if (d->ops[j].queue == q) { if (first == d->n_ops) first = j; last = j; }
        uint64_t st[12];
        const nlq_feed f0 = feed[first], f1 = feed[(size_t)(capacity - 1) * d->n_ops + last];
        state_before(&queues[q], d->width[q], f0.idx, st);
        for (uint32_t k = 0; k < d->width[q]; k++) TR(nlq_bnd_col(d, q, 0, k), q0) = st[k];
        state_before(&queues[q], d->width[q], (uint64_t)f1.idx + (f1.en ? 1 : 0), st);
        for (uint32_t k = 0; k < d->width[q]; k++) TR(nlq_bnd_col(d, q, 1, k), q0) = st[k];
    }
    return 0;
}

/* ---- checker: kinds 2 copy, 3 flag, 4 boundary, 6 non-zero unused cell, 7 arithmetic (encoding / select), 8 Poseidon2 relation */
typedef struct { uint64_t n, first; } result;
static void flag(result *r, uint64_t kind, uint64_t idx, uint64_t row) {
    const uint64_t code = (kind << 56) | (idx << 32) | row;
    r->n++;
    if (code < r->first) r->first = code;
}

/* `new` of the last operation on `queue` strictly before operation j of cycle c, or the QBND cell */
static uint64_t prev_new(const nl_spec *sp, const nlq_desc *d, const uint64_t *trace, size_t n_rows, uint32_t capacity, uint32_t c, uint32_t j, uint32_t queue, uint32_t k) {
    const uint32_t G = sp->g;
    for (;;) {
        while (j > 0) {
            j--;
            if (d->ops[j].queue == queue) return QCELL(nlq_op_row0(d, G, j), nlq_new0(&d->ops[j]) + k);
        }
        if (c == 0) return TR(nlq_bnd_col(d, queue, 0, k), NLQ_BASE(sp, capacity));
        c--;
        j = d->n_ops;
    }
}

uint64_t orc_nlq_check(int circuit_type, const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    const nlq_desc *d = nlq_desc_of(circuit_type);
    result res = {0, ~0ull};
    *first_bad = ~0ull;
    if (!sp || !d) return 0;
    if (nlq_used_rows(sp, d, capacity) > n_rows) return ~0ull;
    const uint32_t G = sp->g, p2rows = nlq_rows_for(NLQ_P2_CELLS, G);
    const size_t q0 = NLQ_BASE(sp, capacity);
    for (uint32_t c = 0; c < capacity; c++)
        for (uint32_t j = 0; j < d->n_ops; j++) {
            const nlq_op *op = &d->ops[j];
            const uint32_t w = nlq_kind_width(op->kind), ncomp = nlq_item_comps(op->item), nenc = nlq_item_enc(op->item), r0 = nlq_op_row0(d, G, j);
            const uint32_t ncells = nlq_enc_cells(op), erows = nlq_rows_for(ncells, G);
            const uint64_t row_e = NLQ_ROW(sp, capacity, r0, c);
            uint64_t cells[160];
            for (uint32_t k = 0; k < ncells; k++) cells[k] = QCELL(r0, k);
            const uint64_t en = cells[0];
            /* flag */
            if (en > 1) flag(&res, 3, j, row_e);
            const uint64_t hdr_row = (uint64_t)c * sp->rows_per_cycle;
            if (op->en_rule == NLQ_EN_RESET && en != TR(NL_HDR_RESET, hdr_row)) flag(&res, 3, 0x100 + j, row_e);
            if (op->en_rule == NLQ_EN_ACTIVE && en % P != orc_gl_sub(1, TR(NL_HDR_IDLE, hdr_row) % P)) flag(&res, 3, 0x100 + j, row_e);
            /* linked components */
            int ok = 1;
            for (uint32_t k = 1; k < ncomp; k++)
            {
                int has = 0;
                const uint64_t v = linked_value(sp, trace, n_rows, capacity, c, op, k, &has);
                if (has && cells[k] != v) ok = 0;
            }
            if (!ok) flag(&res, 2, j, row_e);
            /* encodings */
            for (uint32_t e = 0; e < nenc; e++) {
                uint64_t acc = 0;
                for (uint32_t i = 0; i < nlq_enc_n_terms(op->item, e); i++) {
                    const nlq_term tm = nlq_enc_term(op->item, e, i);
                    acc = orc_gl_add(acc, fmul_pow2(cells[tm.cell], tm.shift));
                }
                if (acc != cells[ncomp + e] % P) flag(&res, 7, 32 * j + e, row_e);
            }
            /* recomposition gates (a limb = its bytes) */
            for (uint32_t r = 0; r < nlq_aux_n(op->item); r++) {
                uint64_t acc = 0;
                for (uint32_t i = 0; i < nlq_aux_n_terms(op->item, r); i++) {
                    const nlq_term tm = nlq_aux_term(op->item, r, i);
                    acc = orc_gl_add(acc, fmul_pow2(cells[tm.cell], tm.shift));
                }
                if (acc != cells[nlq_aux_result(op->item, r)] % P) flag(&res, 7, 0x400 + 16 * j + r, row_e);
            }
            /* permutations */
            const uint64_t *enc = cells + ncomp, *old = cells + ncomp + nenc, *nw = old + w;
            uint64_t out[12] = {0};
            for (uint32_t p = 0; p < nlq_kind_perms(op->kind); p++) {
                const uint32_t pr0 = nlq_p2_row0(d, G, j, p);
                const uint64_t row_p = NLQ_ROW(sp, capacity, pr0, c);
                uint64_t got[NLQ_P2_CELLS], want_in[12], slots[NLQ_P2_CELLS];
                for (uint32_t k = 0; k < NLQ_P2_CELLS; k++) got[k] = QCELL(pr0, k);
                if (op->kind != NLQ_POP4) { memcpy(want_in, enc, 64); memcpy(want_in + 8, old + 8, 32); }
                else if (p == 0) { memcpy(want_in, enc, 64); memset(want_in + 8, 0, 32); }
                else if (p == 1) { memcpy(want_in, enc + 8, 64); memcpy(want_in + 8, out + 8, 32); }
                else { memcpy(want_in, enc + 16, 32); memcpy(want_in + 4, old, 32); memcpy(want_in + 8, out + 8, 32); }
                if (memcmp(got, want_in, 96) != 0) flag(&res, 2, 0x1000 + 4 * j + p, row_p);
                uint64_t in[12];
                for (int k = 0; k < 12; k++) in[k] = got[k] % P;
                orc_poseidon2_flattened(in, slots);
                if (memcmp(got + 12, slots + 12, (NLQ_P2_CELLS - 12) * 8) != 0) flag(&res, 8, 4 * j + p, row_p);
                memcpy(out, got + NLQ_P2_CELLS - 12, 96);
                for (uint32_t r = 0; r < p2rows; r++) /* unused cells of the block's last row */
                    for (uint32_t col = (r + 1 == p2rows ? NLQ_P2_CELLS - r * G : G); col < G; col++)
                        if (TR(col, NLQ_ROW(sp, capacity, pr0 + r, c))) { flag(&res, 6, col, NLQ_ROW(sp, capacity, pr0 + r, c)); break; }
            }
            /* select: new = old + en * (out - old) */
            ok = 1;
            for (uint32_t k = 0; k < w; k++) {
                const uint64_t o = old[k] % P, want = orc_gl_add(o, orc_gl_mul(en % P, orc_gl_sub(out[k] % P, o)));
                if (want != nw[k] % P) ok = 0;
            }
            if (!ok) flag(&res, 7, 0x800 + j, row_e);
            /* the queue's chain */
            ok = 1;
            for (uint32_t k = 0; k < w; k++)
                if (old[k] != prev_new(sp, d, trace, n_rows, capacity, c, j, op->queue, k)) ok = 0;
            if (!ok) flag(&res, 2, 0x2000 + j, row_e);
            for (uint32_t r = 0; r < erows; r++)
                for (uint32_t col = (r + 1 == erows ? ncells - r * G : G); col < G; col++)
                    if (TR(col, NLQ_ROW(sp, capacity, r0 + r, c))) { flag(&res, 6, col, NLQ_ROW(sp, capacity, r0 + r, c)); break; }
        }
    /* relations between the operations of a cycle (kind 7, index 0x1000 + relation, at the gate operation's ENC row) */
    const nlq_rels *rels = nlq_rels_of(circuit_type);
    for (uint32_t c = 0; c < capacity; c++)
        for (uint32_t i = 0; i < rels->n; i++) {
            const nlq_rel *r = &rels->r[i];
            if (r->prev && c == 0) continue;
            uint64_t en = r->gate == NLQ_REL_ACTIVE ? orc_gl_sub(1, TR(NL_HDR_IDLE, (size_t)c * sp->rows_per_cycle) % P) : QCELL(nlq_op_row0(d, G, r->gate), 0) % P;
            if (r->gate2 != NLQ_REL_CONST) en = orc_gl_sub(en, QCELL(nlq_op_row0(d, G, r->gate2), 0) % P);
            const uint64_t b = QCELL(nlq_op_row0(d, G, r->op_b), r->cell_b) % P;
            const uint32_t ca = r->prev ? c - 1 : c;
            uint64_t a = 0;
            for (uint32_t k = 0; r->op_a != NLQ_REL_CONST && k < (r->span ? r->span : 1u); k++) { /* (span: little-endian recomposition of byte cells) */
                const uint32_t cell = r->cell_a + k;
                a = orc_gl_add(a, fmul_pow2(TR(cell % G, NLQ_ROW(sp, capacity, nlq_op_row0(d, G, r->op_a) + cell / G, ca)), 8 * k));
            }
            const uint64_t diff = orc_gl_sub(r->prev == 3 ? orc_gl_add(b, a) : orc_gl_sub(b, a), r->add >= 0 ? (uint64_t)r->add : P - (uint64_t)(-r->add));
            if (orc_gl_mul(en, diff) != 0) flag(&res, 7, 0x1000 + i, NLQ_ROW(sp, capacity, nlq_op_row0(d, G, r->gate == NLQ_REL_ACTIVE ? r->op_b : r->gate), c));
        }
    /* QBND */
    for (uint32_t q = 0; q < d->n_queues; q++) {
        int ok = 1;
        for (uint32_t k = 0; k < d->width[q]; k++)
            if (TR(nlq_bnd_col(d, q, 1, k), q0) != prev_new(sp, d, trace, n_rows, capacity, capacity, 0, q, k)) ok = 0;
        if (!ok) flag(&res, 4, 0x100 + q, q0);
    }
    for (uint32_t col = nlq_bnd_cells(d); col < G; col++)
        if (TR(col, q0)) { flag(&res, 6, col, q0); break; }
    *first_bad = res.first;
    return res.n;
}

/* ---- feeds. Sha256RoundFunction: rounds in the global order; a request = `reset` round .. the round before the next reset; per
   round two reads, the last round of a request also writes (sha256_round_function.rs:204-246). The instance covers the rounds
   [first_round, first_round + n_active), idle cycles after them. feed: [capacity][4]. */
void orc_sha256_queue_feed(const zkw_sha256_round_record *rounds, size_t total_rounds, size_t first_round, uint32_t n_active, uint32_t capacity, nlq_feed *feed) {
    size_t req = 0, q = 0; /* requests popped / queries pushed before the current round */
    for (size_t r = 0; r < first_round; r++) {
        if (rounds[r].reset) req++;
        q += 2;
        if (r + 1 == total_rounds || rounds[r + 1].reset) q++;
    }
    for (uint32_t c = 0; c < capacity; c++) {
        nlq_feed *f = feed + (size_t)c * 4;
        if (c >= n_active) {
            f[0] = (nlq_feed){0, (uint32_t)req}; f[1] = f[2] = f[3] = (nlq_feed){0, (uint32_t)q};
            continue;
        }
        const size_t r = first_round + c;
        const int pop = rounds[r].reset != 0, last = r + 1 == total_rounds || rounds[r + 1].reset;
        f[0] = (nlq_feed){(uint32_t)pop, (uint32_t)req};
        if (pop) req++;
        f[1] = (nlq_feed){1, (uint32_t)q};
        f[2] = (nlq_feed){1, (uint32_t)q + 1};
        q += 2;
        size_t end = r + 1; /* the request's rounds end before the next reset */
        while (end < total_rounds && !rounds[end].reset) end++;
        f[3] = (nlq_feed){(uint32_t)last, (uint32_t)q, (uint32_t)(end - r - 1)};
        if (last) q++;
    }
}

/* CodeDecommitter: a bytecode = `reset` round .. the round before the next reset; a round writes two code words, the last round of a
   bytecode with an odd word count one (decommit_code.rs:246-330). words_of_request: code words per request. feed: [capacity][3]. */
void orc_code_decommitter_queue_feed(const zkw_sha256_round_record *rounds, size_t total_rounds, const uint64_t *word_offsets, size_t first_round,
                                     uint32_t n_active, uint32_t capacity, nlq_feed *feed) {
    size_t req = 0, q = 0;
    (void)total_rounds;
    /* walk the rounds from 0: the words a round writes follow from its request's word count */
    size_t words_left = 0;
    for (size_t r = 0; r < first_round + capacity; r++) {
        const int in_range = r >= first_round;
        nlq_feed *f = in_range ? feed + (r - first_round) * 3 : NULL;
        if (r >= first_round + n_active) {
            f[0] = (nlq_feed){0, (uint32_t)req}; f[1] = f[2] = (nlq_feed){0, (uint32_t)q};
            continue;
        }
        const int pop = rounds[r].reset != 0;
        if (pop) words_left = (size_t)(word_offsets[req + 1] - word_offsets[req]);
        if (f) f[0] = (nlq_feed){(uint32_t)pop, (uint32_t)req};
        if (pop) req++;
        const int two = words_left >= 2;
        if (f) { f[1] = (nlq_feed){1, (uint32_t)q}; f[2] = (nlq_feed){(uint32_t)two, (uint32_t)q + 1}; }
        q += two ? 2 : 1;
        words_left -= two ? 2 : 1;
    }
}

/* A section for round records that come without their queues (the netlist tests drive the SHA-256 circuits from bare records): the
   queues such records imply — an all-zero request per `reset` round, a memory query per hashed word (value = the block's word, all
   other fields zero) and, for Sha256RoundFunction, a write of the digest after a request's last round — pushed from empty queues. */
int orc_nlq_standalone(int circuit_type, const zkw_sha256_round_record *rounds, uint32_t n_active, uint32_t capacity, size_t n_rows, uint64_t *trace) {
    const nlq_desc *d = nlq_desc_of(circuit_type);
    if (!d) return 0;
    const int sha = circuit_type == 6;
    size_t n_req = 0, n_q = 0;
    zkw_mem_query *mq = calloc((size_t)n_active * 3 + 1, sizeof *mq);
    uint64_t *woff = calloc((size_t)n_active + 2, sizeof *woff);
    for (uint32_t r = 0; r < n_active; r++) {
        if (rounds[r].reset) { n_req++; woff[n_req] = woff[n_req - 1]; }
        const int last = r + 1 == n_active || rounds[r + 1].reset;
        for (int k = 0; k < 2; k++) {
            if (!sha && k == 1 && last) break; /* the padding half-block of a bytecode is not a code word */
            zkw_mem_query *q = &mq[n_q++];
            q->rw_flag = sha ? 0 : 1;
            q->index = (uint32_t)(n_req ? woff[n_req] - woff[n_req - 1] : 0); /* the word's number within its request: consecutive words */
            for (int b = 0; b < 32; b++) q->value[(31 - b) / 4] |= (uint32_t)rounds[r].block[32 * k + b] << (8 * ((31 - b) % 4)); /* U256::from_big_endian */
            if (n_req) woff[n_req]++;
        }
        if (sha && last) {
            zkw_mem_query *q = &mq[n_q++];
            q->rw_flag = 1;
            q->timestamp = 1; /* one tick after the reads */
            for (int j = 0; j < 8; j++) q->value[j] = rounds[r].state_after[7 - j];
        }
    }
    uint64_t *menc = calloc(n_q * 8 + 1, 8), *mtails = calloc(n_q * 12 + 1, 8), zero12[12] = {0};
    orc_encode_memory_queries(mq, n_q, menc);
    orc_queue_push_chain_full(menc, n_q, zero12, mtails);
    void *req = NULL;
    uint64_t *rstates = NULL;
    if (sha) {
        req = calloc(n_req + 1, sizeof(zkw_log_query));
        for (size_t k = 0, r = 0; r < n_active; r++) { /* a call's ABI names its number of rounds (key limb 6) */
            if (rounds[r].reset) k++;
            if (k) ((zkw_log_query *)req)[k - 1].key[6]++;
        }
        uint64_t *renc = calloc(n_req * 20 + 1, 8);
        rstates = calloc(n_req * 4 + 1, 8);
        orc_encode_log_queries(req, n_req, NULL, renc);
        orc_queue_push_chain_log(renc, n_req, zero12, NULL, rstates);
        free(renc);
    } else {
        req = calloc(n_req + 1, sizeof(zkw_decommit_query));
        uint64_t *renc = calloc(n_req * 8 + 1, 8);
        rstates = calloc(n_req * 12 + 1, 8);
        orc_encode_decommit_queries(req, n_req, renc);
        orc_queue_push_chain_full(renc, n_req, zero12, rstates);
        free(renc);
    }
    nlq_feed *feed = calloc((size_t)capacity * d->n_ops + 1, sizeof *feed);
    if (sha) orc_sha256_queue_feed(rounds, n_active, 0, n_active, capacity, feed);
    else orc_code_decommitter_queue_feed(rounds, n_active, woff, 0, n_active, capacity, feed);
    const orc_nlq_queue queues[2] = {{req, rstates, NULL, n_req}, {mq, mtails, NULL, n_q}};
    const int rc = orc_nlq_synthesize(circuit_type, capacity, feed, queues, n_rows, trace);
    free(mq); free(woff); free(menc); free(mtails); free(req); free(rstates); free(feed);
    return rc;
}

/* for the tests: {has a section, first row (QBND), rows per cycle, rows used by netlist + section, operations per cycle, the largest
   capacity in 2^20 rows, queues, reserved} */
void orc_nlq_geometry(int circuit_type, uint32_t capacity, uint64_t out[8]) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    const nlq_desc *d = nlq_desc_of(circuit_type);
    memset(out, 0, 8 * sizeof(uint64_t));
    if (!sp) return;
    out[3] = nlq_used_rows(sp, d, capacity);
    out[5] = nlq_max_capacity(sp, d, 1u << 20);
    if (!d) return;
    out[0] = 1; out[1] = NLQ_BASE(sp, capacity); out[2] = nlq_rows_per_cycle(d, sp->g); out[4] = d->n_ops; out[6] = d->n_queues;
}
/* (column, row) of cell k of operation j of cycle c: block -1 = the ENC block, p >= 0 = P2 block p; j == n_ops: the QBND row (k = column).
   region: 0 = the cell itself, 1 = first enc cell + k, 2 = old + k, 3 = new + k (ENC block only) */
int orc_nlq_cell(int circuit_type, uint32_t capacity, uint32_t c, uint32_t j, int block, int region, uint32_t k, uint64_t out[2]) {
    const nl_spec *sp = orc_nl_spec(circuit_type);
    const nlq_desc *d = nlq_desc_of(circuit_type);
    if (!sp || !d || j > d->n_ops) return -1;
    if (j == d->n_ops) { out[0] = k; out[1] = NLQ_BASE(sp, capacity); return 0; }
    const nlq_op *op = &d->ops[j];
    if (block < 0) k += region == 1 ? nlq_enc0(op) : region == 2 ? nlq_old0(op) : region == 3 ? nlq_new0(op) : 0;
    const uint32_t r0 = block < 0 ? nlq_op_row0(d, sp->g, j) : nlq_p2_row0(d, sp->g, j, (uint32_t)block);
    out[0] = k % sp->g;
    out[1] = NLQ_ROW(sp, capacity, r0 + k / sp->g, c);
    return 0;
}

/* Keccak256RoundFunction: per round up to ZKW_KECCAK_MEMORY_READS_PER_CYCLE unaligned reads into the 192-byte buffer — a word is read
   when it has meaningful bytes left and the buffer has room for them (keccak256_round_function.rs:232-290) — and the digest write
   after a request's last round (:366-380). The rounds of all requests in their global order; feed: [capacity][8]. */
void orc_keccak_queue_feed(const zkw_log_query *requests, size_t n_req, size_t first_round, uint32_t n_active, uint32_t capacity, nlq_feed *feed) {
    size_t g = 0, q = 0, popped = 0;
    for (size_t r = 0; r < n_req && g < first_round + n_active; r++) {
        uint32_t off = requests[r].key[0], len = requests[r].key[1], filled = 0;
        const size_t rounds = (len + 135) / 136 + (len % 136 == 0 ? 1 : 0);
        for (size_t round = 0; round < rounds && g < first_round + n_active; round++, g++) {
            nlq_feed *f = g >= first_round ? feed + (g - first_round) * 8 : NULL;
            if (f) f[0] = (nlq_feed){round == 0, (uint32_t)popped}; /* (a later round: the call is popped already, `popped` is the next one) */
            if (round == 0) popped++;
            uint32_t n_reads = 0;
            for (int slot = 0; slot < ZKW_KECCAK_MEMORY_READS_PER_CYCLE; slot++) {
                const uint32_t at_most = 32 - off % 32, meaningful = len >= at_most ? at_most : len;
                if (meaningful == 0 || filled + meaningful > ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE) continue;
                off += meaningful; len -= meaningful; filled += meaningful;
                n_reads++;
            }
            filled = filled < 136 ? 0 : filled - 136;
            const int last = round + 1 == rounds;
            for (uint32_t k = 0; k < 6 && f; k++) f[1 + k] = (nlq_feed){k < n_reads, (uint32_t)(q + (k < n_reads ? k : n_reads))};
            q += n_reads;
            if (f) f[7] = (nlq_feed){(uint32_t)last, (uint32_t)q};
            if (last) q++;
        }
    }
    for (uint32_t c = n_active; c < capacity; c++) {
        nlq_feed *f = feed + (size_t)c * 8;
        f[0] = (nlq_feed){0, (uint32_t)popped};
        for (int k = 1; k < 8; k++) f[k] = (nlq_feed){0, (uint32_t)q};
    }
}

/* the section that bare Keccak round records imply: an all-zero request per `reset` round, no reads (their alignment is not in the
   records), a write of the digest after a request's last round */
int orc_nlq_standalone_keccak(int circuit_type, const zkw_keccak_round_record *rounds, uint32_t n_active, uint32_t capacity, size_t n_rows, uint64_t *trace) {
    const nlq_desc *d = nlq_desc_of(circuit_type);
    if (!d) return 0;
    if (circuit_type == 13) return orc_linear_hasher_queue_section(NULL, 0, NULL, capacity, n_rows, trace); /* bare records: no message is known, nothing is popped */
    size_t n_req = 0, n_q = 0;
    zkw_mem_query *mq = calloc((size_t)n_active + 1, sizeof *mq);
    nlq_feed *feed = calloc((size_t)capacity * d->n_ops + 1, sizeof *feed);
    for (uint32_t c = 0; c < capacity; c++) {
        nlq_feed *f = feed + (size_t)c * d->n_ops;
        const int active = c < n_active, pop = active && rounds[c].reset, last = active && (c + 1 == n_active || rounds[c + 1].reset);
        f[0] = (nlq_feed){(uint32_t)pop, (uint32_t)n_req};
        if (pop) n_req++;
        if (!pop) f[0].idx = (uint32_t)n_req;
        for (uint32_t k = 1; k + 1 < d->n_ops; k++) f[k] = (nlq_feed){0, (uint32_t)n_q};
        f[d->n_ops - 1] = (nlq_feed){(uint32_t)last, (uint32_t)n_q};
        if (last) {
            zkw_mem_query *w = &mq[n_q++];
            w->rw_flag = 1;
            for (int b = 0; b < 32; b++) w->value[(31 - b) / 4] |= (uint32_t)rounds[c].state_after[b] << (8 * ((31 - b) % 4));
        }
    }
    uint64_t *menc = calloc(n_q * 8 + 1, 8), *mtails = calloc(n_q * 12 + 1, 8), zero12[12] = {0};
    orc_encode_memory_queries(mq, n_q, menc);
    orc_queue_push_chain_full(menc, n_q, zero12, mtails);
    zkw_log_query *req = calloc(n_req + 1, sizeof *req);
    uint64_t *renc = calloc(n_req * 20 + 1, 8), *rstates = calloc(n_req * 4 + 1, 8);
    orc_encode_log_queries(req, n_req, NULL, renc);
    orc_queue_push_chain_log(renc, n_req, zero12, NULL, rstates);
    const orc_nlq_queue queues[2] = {{req, rstates, NULL, n_req}, {mq, mtails, NULL, n_q}};
    const int rc = orc_nlq_synthesize(circuit_type, capacity, feed, queues, n_rows, trace);
    free(mq); free(feed); free(menc); free(mtails); free(req); free(renc); free(rstates);
    return rc;
}

/* L1MessagesHasher: message m of the n is popped in the cycle that absorbs its first byte; feed: [cycles][2] */
void orc_linear_hasher_queue_feed(size_t n_messages, uint32_t cycles, nlq_feed *feed) {
    for (uint32_t c = 0; c < cycles; c++) {
        uint64_t m0 = NLQ_LH_FIRST(c), m1 = NLQ_LH_FIRST(c + 1);
        if (m0 > n_messages) m0 = n_messages;
        if (m1 > n_messages) m1 = n_messages;
        for (uint32_t k = 0; k < 2; k++) {
            const int en = m0 + k < m1;
            feed[(size_t)c * 2 + k] = (nlq_feed){(uint32_t)en, (uint32_t)(en ? m0 + k : m1)};
        }
    }
}
/* the section of a LinearHasher trace: `messages` popped from a queue whose head is `head` (NULL: zeros, the empty queue's state) */
int orc_linear_hasher_queue_section(const zkw_log_query *messages, size_t n, const uint64_t *head, uint32_t cycles, size_t n_rows, uint64_t *trace) {
    uint64_t *enc = calloc(n * 20 + 1, 8), *states = calloc(n * 4 + 1, 8), zero[4] = {0};
    if (!head) head = zero;
    nlq_feed *feed = calloc((size_t)cycles * 2 + 1, sizeof *feed);
    orc_encode_log_queries(messages, n, NULL, enc);
    orc_queue_push_chain_log(enc, n, head, NULL, states);
    orc_linear_hasher_queue_feed(n, cycles, feed);
    const orc_nlq_queue queues[2] = {{messages, states, head, n}, {NULL, NULL, NULL, 0}};
    const int rc = orc_nlq_synthesize(13, cycles, feed, queues, n_rows, trace);
    free(enc); free(states); free(feed);
    return rc;
}
