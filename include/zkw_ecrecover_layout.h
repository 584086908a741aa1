/* zkw_ecrecover_layout.h — the FORMAT of the generated spec of the ECRecover circuit's EC section (include/zkw_ecrecover_ec_spec.h,
 * written by tools/gen_ecrecover_circuit.py): record types, the encoding of items and references, where a tape value's home cell
 * is, which table a row looks up. Data layout only — what an item COMPUTES or STATES is not here: the library's evaluator and
 * relations are include/zkw_ecrecover.h (HIP + host), the test oracle has its own in oracle/ecrecover_eval.c and includes only this
 * file. Plain C.
 *
 * Rows of cycle c: EC_FIRST_ROW + c * EC_ROWS_PER_CYCLE + run.row0 + instance * type.n_rows + row; a row = 80 general-purpose
 * cells + 16 lookup slots of width 3 of ONE table. Items (32-bit words, w0 = kind | row << 4 | col << 16 | aux << 24):
 *   LIN    aux = known cells; n_new; const lo, hi; known x {ref, coef (i32)}; new x {tape index, shift | width << 8}
 *   SEL    refs b, x, y; out tape index                      cells [b, x, y, o]
 *   FMA    aux = 1: d is NEW; refs a, b, c; d (tape index / ref)  cells [a, b, c, d]
 *   MUL    aux = modulus (0 P, 1 N); refs a0, b0, r0 (limb i = ref + i); q tape0, carry tape0   cells a 0.., b 16.., q 32.., r 48.., c 64..78
 *   HINT   aux = kind; arguments (no cells)
 *   LOOKUP row, col = slot; aux = inputs; table (| EC_ROWTAB_PER_INSTANCE); in0, in1; out tape0   cells [in.., out..] at 80 + 3 slot
 * References: kind << 28 | payload — 0 TAPE t, 1 PREV k (state element k of the previous segment), 2 GLOB k, 3 GLOBJ (base | stride << 16:
 * global base + stride * instance), 4 CONST v, 5 BIG (idx << 4 | limb), 6 IN k (input byte), 0xFFFFFFFF none. */
#ifndef ZKW_ECRECOVER_LAYOUT_H
#define ZKW_ECRECOVER_LAYOUT_H
#include <stdint.h>
#include <stddef.h>
#include "zkw_ecrecover_ec_spec.h"

#if defined(__HIPCC__)
#define EC_HD __host__ __device__ __forceinline__ static
#else
#define EC_HD static inline
#endif

/* The spec's arrays through the CONSTANT address space and a tape through the GLOBAL one on the device: a generic pointer makes every access
   a flat load (85 of them in the item walk of a segment's leaves, each a round trip of its own), while a constant-space load at a uniform
   index — the item words, the tables a reference decodes through: the same for all lanes of a wave — is a scalar load. Plain pointers on a host. */
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uint32_t *ec_cw;
typedef const __attribute__((address_space(4))) uint16_t *ec_ch;
typedef __attribute__((address_space(1))) uint64_t *ec_tp;
typedef const __attribute__((address_space(1))) uint64_t *ec_ctp;
#define EC_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x))) /* a value every lane of the wave holds alike: into a scalar register */
#else
#define EC_UNIFORM(x) ((uint32_t)(x))
typedef const uint32_t *ec_cw;
typedef const uint16_t *ec_ch;
typedef uint64_t *ec_tp;
typedef const uint64_t *ec_ctp;
#endif

enum { EC_I_LIN = 1, EC_I_SEL = 2, EC_I_FMA = 3, EC_I_MUL = 4, EC_I_HINT = 5, EC_I_LOOKUP = 6 };
enum { EC_H_MULSUB = 1, EC_H_DIV = 2, EC_H_SQRT = 3, EC_H_ISZERO = 4, EC_H_GE = 5 };
enum { EC_K_TAPE = 0, EC_K_PREV = 1, EC_K_GLOB = 2, EC_K_GLOBJ = 3, EC_K_CONST = 4, EC_K_BIG = 5, EC_K_IN = 6 };
#define EC_NONE 0xFFFFFFFFu
#define EC_GL_P 0xFFFFFFFF00000001ull
#define EC_STATE 32
#define EC_FIXED_WORDS (256 * 256 * 2) /* the 256 FixedBaseMul tables: [8 C + i][byte] -> {x word i, y word i} */

typedef struct ec_seg_type { uint32_t n_rows, n_tape, item0, n_items, index0, cell0, home0, out0, rowtab0; } ec_seg_type;
typedef struct ec_run { uint32_t type, count, row0, tape0; } ec_run;
typedef struct ec_spec {
    const ec_seg_type *types;
    const ec_run *runs;
    const uint32_t *items, *item_index, *cells, *homes, *outs;
    const uint16_t *rowtab;
    const uint32_t *globs, *bigs;
    const uint16_t *in_home;
    const uint32_t *key_byte;
    const uint32_t *fixed; /* EC_FIXED_WORDS, the 256 FixedBaseMul tables, built by the includer */
} ec_spec;

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) ec_seg_type *ec_ctype;
#else
typedef const ec_seg_type *ec_ctype;
#endif

#define EC_DEFINE_SPEC(name)                                                                                            \
    static const ec_seg_type name##_types[] = EC_TYPES_INIT;                                                            \
    static const ec_run name##_runs[] = EC_RUNS_INIT;                                                                   \
    static const uint32_t name##_items[] = EC_ITEMS_INIT;                                                               \
    static const uint32_t name##_item_index[] = EC_ITEM_INDEX_INIT;                                                     \
    static const uint32_t name##_cells[] = EC_CELLS_INIT;                                                               \
    static const uint32_t name##_homes[] = EC_HOME_INIT;                                                                \
    static const uint32_t name##_outs[] = EC_OUT_INIT;                                                                  \
    static const uint16_t name##_rowtab[] = EC_ROWTAB_INIT;                                                             \
    static const uint32_t name##_globs[] = EC_GLOB_INIT;                                                                \
    static const uint32_t name##_bigs[] = EC_BIG_INIT;                                                                  \
    static const uint16_t name##_in_home[] = EC_IN_HOME_INIT;                                                           \
    static const uint32_t name##_key_byte[] = EC_KEY_BYTE_INIT

/* absolute tape index a reference names, or EC_NONE for the kinds that are not tape values */
EC_HD uint32_t ec_ref_tape(const ec_spec *S, uint32_t ref, uint32_t base, uint32_t prev_base, uint32_t prev_type, uint32_t inst) {
    const uint32_t kind = ref >> 28, a = ref & 0x0FFFFFFFu;
    switch (kind) {
        case EC_K_TAPE: return base + a;
        case EC_K_PREV: return prev_base + ((ec_cw)S->outs)[((ec_ctype)S->types)[prev_type].out0 + a];
        case EC_K_GLOB: return ((ec_cw)S->globs)[a];
        case EC_K_GLOBJ: return ((ec_cw)S->globs)[(a & 0xFFFFu) + (uint32_t)((int32_t)(int8_t)(a >> 16) * (int32_t)inst)];
        default: return EC_NONE;
    }
}
EC_HD uint64_t ec_ref_const(const ec_spec *S, uint32_t ref, const uint8_t *in) {
    const uint32_t kind = ref >> 28, a = ref & 0x0FFFFFFFu;
    if (kind == EC_K_CONST) return a;
    if (kind == EC_K_BIG) return ((ec_cw)S->bigs)[(a >> 4) * 16 + (a & 15)];
    return in[a]; /* EC_K_IN */
}
EC_HD uint32_t ec_item_words_of(uint32_t w0, uint32_t w1) {
    const uint32_t kind = w0 & 15, aux = w0 >> 24;
    switch (kind) {
        case EC_I_LIN: return 4 + 2 * aux + 2 * w1;
        case EC_I_SEL: case EC_I_FMA: case EC_I_LOOKUP: return 5;
        case EC_I_MUL: return 6;
        default: return aux == EC_H_MULSUB ? 7 : aux == EC_H_DIV ? 5 : aux == EC_H_SQRT ? 4 : aux == EC_H_ISZERO ? 3 : 4;
    }
}
EC_HD uint32_t ec_item_words(const uint32_t *w) {
    const uint32_t kind = w[0] & 15, aux = w[0] >> 24;
    switch (kind) {
        case EC_I_LIN: return 4 + 2 * aux + 2 * w[1];
        case EC_I_SEL: case EC_I_FMA: case EC_I_LOOKUP: return 5;
        case EC_I_MUL: return 6;
        default: return aux == EC_H_MULSUB ? 7 : aux == EC_H_DIV ? 5 : aux == EC_H_SQRT ? 4 : aux == EC_H_ISZERO ? 3 : 4;
    }
}

/* ---- layout -------------------------------------------------------------------------------------------------------------- */
/* segment instance that holds row `r` (< EC_ROWS_PER_CYCLE) of a cycle */
EC_HD void ec_locate_row(const ec_spec *S, uint32_t r, uint32_t *run, uint32_t *inst, uint32_t *row) {
    uint32_t k = 0;
    while (k + 1 < EC_NUM_RUNS && S->runs[k + 1].row0 <= r) k++;
    const uint32_t nr = S->types[S->runs[k].type].n_rows;
    *run = k;
    *inst = (r - S->runs[k].row0) / nr;
    *row = (r - S->runs[k].row0) % nr;
}
/* the (run, instance) before (run, inst); run 0 has none */
EC_HD void ec_prev_segment(const ec_spec *S, uint32_t run, uint32_t inst, uint32_t *prun, uint32_t *pinst) {
    if (inst) { *prun = run; *pinst = inst - 1; }
    else { *prun = run ? run - 1 : 0; *pinst = run ? S->runs[run - 1].count - 1 : 0; }
}
/* value of a cell reference seen from segment (run, inst) of a cycle with this tape */
EC_HD uint64_t ec_cell_value(const ec_spec *S, const uint64_t *tape, const uint8_t *in, uint32_t run, uint32_t inst, uint32_t ref) {
    if (ref == EC_NONE) return 0;
    uint32_t prun, pinst;
    ec_prev_segment(S, run, inst, &prun, &pinst);
    const uint32_t base = S->runs[run].tape0 + inst * S->types[S->runs[run].type].n_tape;
    const uint32_t pbase = S->runs[prun].tape0 + pinst * S->types[S->runs[prun].type].n_tape;
    const uint32_t t = ec_ref_tape(S, ref, base, pbase, S->runs[prun].type, inst);
    return t != EC_NONE ? tape[t] : ec_ref_const(S, ref, in);
}
/* row (within the cycle) and column of the HOME cell of absolute tape index t */
EC_HD void ec_home_of_tape(const ec_spec *S, uint32_t t, uint32_t *row, uint32_t *col) {
    uint32_t k = 0;
    while (k + 1 < EC_NUM_RUNS && S->runs[k + 1].tape0 <= t) k++;
    const ec_seg_type *T = &S->types[S->runs[k].type];
    const uint32_t inst = (t - S->runs[k].tape0) / T->n_tape, idx = (t - S->runs[k].tape0) % T->n_tape;
    const uint32_t h = S->homes[T->home0 + idx];
    *row = S->runs[k].row0 + inst * T->n_rows + (h >> 8);
    *col = h & 0xFF;
}
/* table id of a row of segment (run, inst), 0 = the row has no lookups */
EC_HD uint32_t ec_row_table(const ec_spec *S, uint32_t run, uint32_t inst, uint32_t row) {
    const uint32_t t = S->rowtab[S->types[S->runs[run].type].rowtab0 + row];
    return (t & EC_ROWTAB_PER_INSTANCE) ? (t & 0x7FFFu) + 8 * inst : t;
}
/* stacked-table row (= row of the multiplicity column) a lookup of table id `tb` with these inputs hits: Xor8 at 0, And8 at 65 536,
   FixedBaseMul<i, C> (id 3 + 8 C + i) at 131 072 + 256 (id - 3) */
EC_HD uint32_t ec_table_key(uint32_t tb, uint64_t a, uint64_t b) {
    return tb == EC_T_XOR8 ? (uint32_t)(a | (b << 8)) : 131072u + 256u * (tb - EC_T_FIXED0) + (uint32_t)a;
}

#endif /* ZKW_ECRECOVER_LAYOUT_H */
