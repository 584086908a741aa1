/* zkw_netlist.h — "zkw trace v4": the record types of the generic netlist format that the bit-gate-heavy base-layer circuits
 * (Sha256RoundFunction 6, CodeDecommitter 3, Keccak256RoundFunction 5, L1MessagesHasher 13) are emitted in. The format and the
 * reference geometry / table sets it reproduces are described in tools/netlist.py; the per-circuit specs are the generated
 * include/zkw_*_circuit_spec.h. Shared by the HIP kernels (csrc/netlist_kernels.cuh), the host side (setup: selectors, copy
 * permutation) and the test oracle (oracle/netlist_circuit.c): record types and layout macros only, no semantics. Plain C.
 *
 * Trace of an instance with `capacity` cycles, column-major u64[cols][n_rows]:
 *   columns [0, G)                 general-purpose (copy-permutation) columns: step headers, gates, boundary rows
 *   columns [G, G + W * R)         R lookups of width W per row, ONE table per row (the reference's share_table_id): slot s of a row
 *                                  = columns G + W * s ..: the table's inputs, then its outputs, zero-padded to W
 *   column  G + W * R              the ONE multiplicity column over the stacked tables (row = table offset + key)
 *   rows: cycle c at c * rows_per_cycle; inside, its steps back to back; step = header row [reset, idle, m0, m1] + lookup rows /
 *   gate rows side by side; after the last cycle: BND_IN (the state before cycle 0: `state` elements over ceil(state / G) rows),
 *   BND_OUT (the state after the last cycle), PI (columns 0..3); zero below. */
#ifndef ZKW_NETLIST_H
#define ZKW_NETLIST_H
#include <stdint.h>

#define NL_REF_HDR 0xC000u
#define NL_REF_PREV 0xC100u
#define NL_REF_CYC 0xC200u
#define NL_REF_FREE 0xC300u
#define NL_REF_RC 0xC400u
#define NL_REF_CONST 0xC500u
#define NL_ORDER_FUSED 0x4000u /* + hint index: the hint and, in the same item, the lookup it keys (nl_hint.fused_slot) */
#define NL_ORDER_GATE 0x8000u
#define NL_ORDER_HINT 0xC000u
#define NL_HDR_RESET 0
#define NL_HDR_IDLE 1
#define NL_HDR_M0 2
#define NL_HDR_M1 3
#define NL_HDR_FIELDS 4

enum { NL_FN_XOR8 = 1, NL_FN_AND8 = 2, NL_FN_BYTESPLIT = 3, NL_FN_TRIXOR4 = 4, NL_FN_CH4 = 5, NL_FN_MAJ4 = 6, NL_FN_SPLIT4 = 7 };

typedef struct nl_table { uint8_t fn, param, n_in, in_bits, n_out; uint32_t rows, offset; } nl_table;
typedef struct nl_op { uint16_t table, in[3], out; } nl_op;
typedef struct nl_gate { uint32_t first_term; uint16_t n_known, n_new; uint32_t constant; uint16_t row, col; } nl_gate;
#define NL_TERM_LATE 0x100 /* a known cell that is part of the constraint but not of the evaluation of the gate's NEW cells */
typedef struct nl_term { uint16_t ref; uint16_t code; /* shift | 0x80: negative coefficient | NL_TERM_LATE */ } nl_term;
typedef struct nl_hint { uint16_t value, ref_a; uint8_t lo_a, n_a; uint16_t ref_b; uint8_t lo_b, n_b; uint16_t fused_slot; /* 0xFFFF: evaluated on its own */ } nl_hint;
typedef struct nl_home { uint16_t kind, item, cell; } nl_home;
typedef struct nl_step_type {
    uint32_t op0, n_ops, gate0, n_gates, term0, n_terms, hint0, n_hints, order0, level0, n_levels, home0, n_values, rows, lookup_rows,
        gate_rows, n_free, rowend0;
} nl_step_type;
typedef struct nl_cycle_step { uint16_t type, row0; uint8_t rc[8]; } nl_cycle_step;

typedef struct nl_spec {
    uint32_t g, w, r, cols, mult_col, n_tables, total_table_rows, state, n_step_types, steps_per_cycle, rows_per_cycle;
    int32_t masks[4]; /* header: m0 = masks[0] + masks[1] * reset, m1 = masks[2] + masks[3] * idle */
    const nl_table *tables;
    const nl_step_type *step_types;
    const nl_op *ops;
    const nl_gate *gates;
    const nl_term *terms;
    const nl_hint *hints;
    const uint16_t *out;         /* [n_step_types][state]: the state a step leaves, as references */
    const uint16_t *order;
    const uint16_t *level_start;
    const nl_home *homes;
    const nl_cycle_step *cycle;  /* [steps_per_cycle] */
    const uint16_t *gate_row_end; /* [step type's rowend0 + row]: general-purpose columns [0, end) of the row are in use */
    uint32_t n_ops, n_gates, n_terms, n_hints, n_values, n_order, n_level_starts, max_values, max_free;
    uint32_t free_per_cycle;     /* sum over the cycle's steps of their n_free */
} nl_spec;

#define NL_BND_ROWS(spec) (((spec)->state + (spec)->g - 1) / (spec)->g)
#define NL_BOUNDARY_ROW(spec, capacity) ((uint64_t)(capacity) * (spec)->rows_per_cycle)
#define NL_PI_ROW(spec, capacity) (NL_BOUNDARY_ROW(spec, capacity) + 2 * NL_BND_ROWS(spec))
#define NL_USED_ROWS(spec, capacity) (NL_PI_ROW(spec, capacity) + 1)

/* What a table computes (boojum's create_*_table contents) is NOT in this header: the library's evaluator is
 * era_zkevm_test_harness_amd/csrc/netlist_eval.cuh, the test oracle enumerates the tables' rows itself (oracle/netlist_tables.c). */

/* a spec struct from the generated macros of one circuit: NL_DEFINE_SPEC(sc, SC) defines `static const nl_spec sc_spec` */
#define NL_DEFINE_SPEC(name, P)                                                                                                   \
    static const nl_table name##_tables[] = P##_TABLES_INIT;                                                                     \
    static const nl_step_type name##_step_types[] = P##_STEP_TYPES_INIT;                                                         \
    static const nl_op name##_ops[] = P##_OPS_INIT;                                                                              \
    static const nl_gate name##_gates[] = P##_GATES_INIT;                                                                        \
    static const nl_term name##_terms[] = P##_TERMS_INIT;                                                                        \
    static const nl_hint name##_hints[] = P##_HINTS_INIT;                                                                        \
    static const uint16_t name##_out[] = P##_OUT_INIT;                                                                           \
    static const uint16_t name##_order[] = P##_ORDER_INIT;                                                                       \
    static const uint16_t name##_level_start[] = P##_LEVEL_START_INIT;                                                           \
    static const nl_home name##_homes[] = P##_VAL_HOME_INIT;                                                                     \
    static const nl_cycle_step name##_cycle[] = P##_CYCLE_INIT;                                                                  \
    static const uint16_t name##_gate_row_end[] = P##_GATE_ROW_END_INIT;                                                         \
    static const nl_spec name##_spec = {P##_G, P##_W, P##_R, P##_COLS, P##_MULT_COL, P##_NUM_TABLES, P##_TOTAL_TABLE_ROWS, P##_STATE,  \
                                        P##_NUM_STEP_TYPES, P##_STEPS_PER_CYCLE, P##_ROWS_PER_CYCLE, P##_MASKS_INIT, name##_tables, \
                                        name##_step_types, name##_ops, name##_gates, name##_terms, name##_hints, name##_out,     \
                                        name##_order, name##_level_start, name##_homes, name##_cycle, name##_gate_row_end, P##_NUM_OPS, P##_NUM_GATES, \
                                        P##_NUM_TERMS, P##_NUM_HINTS, P##_NUM_VALUES, P##_NUM_ORDER, P##_NUM_LEVEL_STARTS,       \
                                        P##_MAX_VALUES, P##_MAX_FREE, P##_FREE_PER_CYCLE}
#endif /* ZKW_NETLIST_H */
