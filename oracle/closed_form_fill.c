/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * The CLOSED-FORM SECTION of a queue circuit's trace (tools/gen_ram_circuit.py, class ClosedForm): what the reference's circuits derive
 * inside the trace — the Fiat-Shamir challenges (produce_fs_challenges, src/witness/utils.rs:498-550), the commitments of the closed-form
 * input and the public input (commit_variable_length_encodable_item / ClosedFormInputCompactForm::from_full_form, utils.rs:269-306), the
 * start-flag selection of the initial state (W/ram_permutation.rs:355-384) — as boundary rows: flattened Poseidon2 rows chained into
 * overwrite-mode sponges, selection rows, copies. One generic walk over the generated tables (links of kind 5, constant cells, FREE
 * cells), sequential and obvious; shares no code with the checker (circuit_check.c). */
#include <string.h>
#include "oracle.h"
#include "../include/zkw_ram_circuit_spec.h"
#include "../include/zkw_decommit_sorter_circuit_spec.h"
#include "../include/zkw_events_sorter_circuit_spec.h"
#include "../include/zkw_log_demux_circuit_spec.h"
#include "../include/zkw_storage_sorter_circuit_spec.h"

#define CELL(col, row) trace[(size_t)(col) * n_rows + (row)]

typedef struct {
    int first, n, n_links, n_consts, n_free, n_products, rows_per_cycle, pi_row_type, n_bytes, n_linears, mult_col;
    const rc_link *links;
    const uint8_t *is_poseidon;
    const rc_cf_const *consts;
    const rc_cf_free *frees;
    const rc_cf_product *products;
    const rc_cf_bytes *bytes;       /* lookup cells = the bytes of a limb cell of the row (the multiplicity column follows: +1 byte, -1 zero) */
    const rc_cf_linear *linears;    /* cell = constant + sum coef * cell of the row */
    const rc_cf_lin_term *lin_terms;
} orc_cf_spec;
/* the tables of the spec header with prefix PFX as an orc_cf_spec named CF_<PFX> */
#define CF_SPEC(PFX)                                                                                                                  \
    static const rc_link PFX##_links_[] = PFX##_LINKS_INIT;                                                                           \
    static const uint8_t PFX##_is_poseidon_[] = PFX##_ROW_IS_POSEIDON_INIT;                                                           \
    static const rc_cf_const PFX##_consts_[] = PFX##_CF_CONSTS_INIT;                                                                  \
    static const rc_cf_free PFX##_frees_[] = PFX##_CF_FREE_INIT;                                                                      \
    static const rc_cf_product PFX##_products_[] = PFX##_CF_PRODUCTS_INIT;                                                            \
    static const rc_cf_bytes PFX##_bytes_[] = PFX##_CF_BYTES_INIT;                                                                    \
    static const rc_cf_linear PFX##_linears_[] = PFX##_CF_LINEARS_INIT;                                                               \
    static const rc_cf_lin_term PFX##_lin_terms_[] = PFX##_CF_LIN_TERMS_INIT;                                                         \
    static const orc_cf_spec CF_##PFX = {PFX##_CF_FIRST_ROW_TYPE, PFX##_CF_NUM_ROWS, PFX##_NUM_LINKS, PFX##_CF_NUM_CONSTS, PFX##_CF_NUM_FREE, \
                                         PFX##_CF_NUM_PRODUCTS, PFX##_ROWS_PER_CYCLE, PFX##_ROW_PI, PFX##_CF_NUM_BYTES, PFX##_CF_NUM_LINEARS,  \
                                         PFX##_MULT_COL, PFX##_links_, PFX##_is_poseidon_, PFX##_consts_, PFX##_frees_, PFX##_products_, \
                                         PFX##_bytes_, PFX##_linears_, PFX##_lin_terms_}
typedef void (*orc_cf_hook)(void *user, int row_type, uint64_t *trace, size_t n_rows, size_t row);

/* src[k]: the encoding FREE cells of source k are taken from (0 observable input, 1 hidden FSM input, 2 hidden FSM output, 3 flags,
   4 observable output); bnd = first boundary row of the trace */
static void cf_fill(const orc_cf_spec *S, uint64_t *trace, size_t n_rows, size_t bnd, const uint64_t *const src[5], orc_cf_hook hook, void *user) {
#define ROW_OF(rt) (bnd + (size_t)((rt) - S->rows_per_cycle))
    for (int r = S->first; r <= S->first + S->n; r++) {
        const int rt = r < S->first + S->n ? r : S->pi_row_type; /* last: the public-input row takes its copies */
        const size_t row = ROW_OF(rt);
        for (int l = 0; l < S->n_links; l++)
            if (S->links[l].kind == 5 && S->links[l].row_a == rt) CELL(S->links[l].col_a, row) = CELL(S->links[l].col_b, ROW_OF(S->links[l].row_b));
        if (rt == S->pi_row_type) break;
        for (int k = 0; k < S->n_consts; k++)
            if (S->consts[k].row == rt) CELL(S->consts[k].col, row) = S->consts[k].value;
        for (int k = 0; k < S->n_free; k++)
            if (S->frees[k].row == rt) CELL(S->frees[k].col, row) = src[S->frees[k].src][S->frees[k].idx];
        for (int k = 0; k < S->n_bytes; k++)
            if (S->bytes[k].row == rt) {
                const uint32_t limb = (uint32_t)CELL(S->bytes[k].col_limb, row);
                for (int b = 0; b < 4; b++) {
                    const uint64_t v = (limb >> (8 * b)) & 0xFF;
                    CELL(S->bytes[k].col_b0 + b, row) = v;
                    CELL(S->mult_col, v) += 1;
                    CELL(S->mult_col, 0) -= 1;
                }
            }
        for (int k = 0; k < S->n_linears; k++)
            if (S->linears[k].row == rt) {
                uint64_t acc = S->linears[k].constant;
                for (int t = 0; t < S->linears[k].n_terms; t++) {
                    const rc_cf_lin_term *tm = &S->lin_terms[S->linears[k].term0 + t];
                    acc = orc_gl_add(acc, orc_gl_mul(tm->coef, CELL(tm->col, row)));
                }
                CELL(S->linears[k].col, row) = acc;
            }
        for (int k = 0; k < S->n_products; k++)
            if (S->products[k].row == rt) CELL(S->products[k].col, row) = orc_gl_mul(CELL(S->products[k].col_a, row), CELL(S->products[k].col_b, row));
        if (hook) hook(user, rt, trace, n_rows, row);
        if (S->is_poseidon[rt]) {
            uint64_t in[12], slots[130];
            for (int k = 0; k < 12; k++) in[k] = CELL(k, row);
            orc_poseidon2_flattened(in, slots);
            for (int k = 0; k < 130; k++) CELL(k, row) = slots[k];
        }
    }
#undef ROW_OF
}

/* ---- RAMPermutation (type 8) ---- */
CF_SPEC(RC);

/* the value rows: bytes of limbs 5..7 into the lookup cells (and the multiplicity column: +1 for the byte, -1 for the zero the cell held);
   VIN also the encoding elements es3..es6 of the limbs (memory_query.rs:60-110) */
static void ram_value_row(uint64_t *trace, size_t n_rows, size_t row, int v0, int b0, int e3) {
    uint64_t v[8];
    for (int k = 0; k < 8; k++) v[k] = CELL(v0 + k, row);
    for (int l = 5; l < 8; l++)
        for (int k = 0; k < 4; k++) {
            const uint64_t b = (v[l] >> (8 * k)) & 0xFF;
            CELL(b0 + 4 * (l - 5) + k, row) = b;
            CELL(RC_MULT_COL, b) += 1;
            CELL(RC_MULT_COL, 0) -= 1;
        }
    if (e3 < 0) return;
    zkw_mem_query q;
    memset(&q, 0, sizeof q);
    for (int k = 0; k < 8; k++) q.value[k] = (uint32_t)v[k];
    uint64_t e[8];
    orc_encode_memory_query(&q, e);
    for (int k = 0; k < 4; k++) CELL(e3 + k, row) = e[3 + k];
}
static void ram_hook(void *user, int rt, uint64_t *trace, size_t n_rows, size_t row) {
    (void)user;
    if (rt == RC_ROW_VIN) ram_value_row(trace, n_rows, row, RC_VIN_VIN_v0, RC_VIN_VIN_v5_b0, RC_VIN_VIN_e3);
    if (rt == RC_ROW_VOUT) ram_value_row(trace, n_rows, row, RC_VOUT_VOUT_v0, RC_VOUT_VOUT_v5_b0, -1);
}

/* the closed-form section and the PI row of a synthesized RAM trace (after orc_ram_synthesize). `first` = the block's first instance */
void orc_ram_fill_public_input(const zkw_ram_instance *first, const zkw_ram_instance *in, uint32_t capacity, size_t n_rows, uint64_t *trace) {
    uint64_t oi[ORC_RAM_INPUT_ENC_LEN], fi[ORC_RAM_FSM_ENC_LEN], fo[ORC_RAM_FSM_ENC_LEN], flags[2] = {in->start_flag ? 1u : 0u, in->completion_flag ? 1u : 0u};
    orc_ram_encode_observable_input(first, oi);
    orc_ram_encode_fsm(&in->hidden_fsm_input, fi);
    orc_ram_encode_fsm(&in->hidden_fsm_output, fo);
    const uint64_t *src[5] = {oi, fi, fo, flags, NULL};
    cf_fill(&CF_RC, trace, n_rows, (size_t)RC_BOUNDARY_ROW(capacity), src, ram_hook, NULL);
}

/* a lookup cell of a section row takes a byte: the multiplicity column counted a zero there (the register fills count every cell of the
   lookup columns before the section is filled) */
static void put_byte(uint64_t *trace, size_t n_rows, int mult_col, int col, size_t row, uint64_t b) {
    CELL(col, row) = b;
    CELL(mult_col, b) += 1;
    CELL(mult_col, 0) -= 1;
}

/* ---- CodeDecommittmentsSorter (type 2) ---- */
CF_SPEC(DS);
/* GIN: the encoding of the open group's first request (previous_record.hash, page, first_encountered_timestamp, fresh) from the FSM words the
   row copied (decommit query encoding, oracle.c orc_encode_decommit_queries) */
static void ds_hook(void *user, int rt, uint64_t *trace, size_t n_rows, size_t row) {
    (void)user;
    if (rt != DS_ROW_GIN) return;
    zkw_decommit_query g;
    memset(&g, 0, sizeof g);
    for (int k = 0; k < 8; k++) g.hash[k] = (uint32_t)CELL(DS_GIN_gh0 + k, row);
    g.memory_page = (uint32_t)CELL(DS_GIN_gpage, row);
    g.timestamp = (uint32_t)CELL(DS_GIN_gfts, row);
    g.is_fresh = 1;
    for (int k = 0; k < 4; k++) {
        put_byte(trace, n_rows, DS_MULT_COL, DS_GIN_gpage_b0 + k, row, (g.memory_page >> (8 * k)) & 0xFF);
        put_byte(trace, n_rows, DS_MULT_COL, DS_GIN_gfts_b0 + k, row, (g.timestamp >> (8 * k)) & 0xFF);
    }
    uint64_t e[8];
    orc_encode_decommit_queries(&g, 1, e);
    for (int k = 0; k < 3; k++) CELL(DS_GIN_gge0 + k, row) = e[k];
}
void orc_ds_fill_closed_form(const zkw_decommit_sorter_instance *first, const zkw_decommit_sorter_instance *in, uint32_t capacity, size_t n_rows,
                             uint64_t *trace) {
    uint64_t oi[50], oo[25], fi[ORC_DS_FSM_ENC_LEN], fo[ORC_DS_FSM_ENC_LEN], flags[2] = {in->start_flag ? 1u : 0u, in->completion_flag ? 1u : 0u};
    orc_put_queue12(&first->initial_queue_state, oi);
    orc_put_queue12(&first->sorted_queue_initial_state, oi + 25);
    orc_put_queue12(&in->final_queue_state, oo);
    orc_ds_encode_fsm(&in->hidden_fsm_input, fi);
    orc_ds_encode_fsm(&in->hidden_fsm_output, fo);
    const uint64_t *src[5] = {oi, fi, fo, flags, oo};
    cf_fill(&CF_DS, trace, n_rows, (size_t)DS_BOUNDARY_ROW(capacity), src, ds_hook, NULL);
}

/* ---- EventsSorter / L1MessagesSorter (types 11 / 12) ---- */
CF_SPEC(ES);
void orc_es_fill_closed_form(const zkw_events_sorter_instance *first, const zkw_events_sorter_instance *in, uint32_t capacity, size_t n_rows,
                             uint64_t *trace) {
    uint64_t oi[18], oo[9], fi[80], fo[80], flags[2] = {in->start_flag ? 1u : 0u, in->completion_flag ? 1u : 0u};
    orc_put_queue4(&first->initial_log_queue_state, oi);
    orc_put_queue4(&first->intermediate_sorted_queue_state, oi + 9);
    orc_put_queue4(&in->final_queue_state, oo);
    orc_es_fsm(&in->hidden_fsm_input, fi);
    orc_es_fsm(&in->hidden_fsm_output, fo);
    const uint64_t *src[5] = {oi, fi, fo, flags, oo};
    cf_fill(&CF_ES, trace, n_rows, (size_t)ES_BOUNDARY_ROW(capacity), src, NULL, NULL);
}

/* ---- LogDemuxer (type 4) ---- */
CF_SPEC(LD);
void orc_ld_fill_closed_form(const zkw_log_demux_instance *first, const zkw_log_demux_instance *in, uint32_t capacity, size_t n_rows, uint64_t *trace) {
    uint64_t oi[9], oo[64], fi[64], fo[64], flags[2] = {in->start_flag ? 1u : 0u, in->completion_flag ? 1u : 0u};
    orc_put_queue4(&first->initial_log_queue_state, oi);
    size_t m = 0;
    for (int c = 0; c < ZKW_DEMUX_NUM_QUEUES; c++) m += orc_put_queue4(&in->output_queue_state[c], oo + m);
    orc_ld_fsm(&in->hidden_fsm_input, fi);
    orc_ld_fsm(&in->hidden_fsm_output, fo);
    const uint64_t *src[5] = {oi, fi, fo, flags, oo};
    cf_fill(&CF_LD, trace, n_rows, (size_t)LD_BOUNDARY_ROW(capacity), src, NULL, NULL);
}

/* ---- StorageSorter (type 9) ---- */
CF_SPEC(SS);
void orc_ss_fill_closed_form(const zkw_storage_sorter_instance *first, const zkw_storage_sorter_instance *in, uint32_t capacity, size_t n_rows,
                             uint64_t *trace) {
    uint64_t oi[19], oo[9], fi[80], fo[80], flags[2] = {in->start_flag ? 1u : 0u, in->completion_flag ? 1u : 0u};
    oi[0] = first->shard_id_to_process;
    orc_put_queue4(&first->unsorted_log_queue_state, oi + 1);
    orc_put_queue4(&first->intermediate_sorted_queue_state, oi + 10);
    orc_put_queue4(&in->final_sorted_queue_state, oo);
    orc_ss_fsm(&in->hidden_fsm_input, fi);
    orc_ss_fsm(&in->hidden_fsm_output, fo);
    const uint64_t *src[5] = {oi, fi, fo, flags, oo};
    cf_fill(&CF_SS, trace, n_rows, (size_t)SS_BOUNDARY_ROW(capacity), src, NULL, NULL);
}
