/* The LIBRARY's evaluator of the ECRecover EC section (include/zkw_ecrecover.h, the code the kernels run) compiled for the host, for
 * tests/test_ec_library_evaluator_host.py: a cycle's tape in program order, and in the order the fast form of the kernels walks it
 * (PRE's MAIN items; the other segments; PRE's remaining parts last). Test infrastructure. */
#include <stdlib.h>
#include <string.h>
#include "zkw_ecrecover.h"

EC_DEFINE_SPEC(ecs);
static uint32_t *g_fixed;

static ec_spec spec(void) {
    if (!g_fixed) {
        g_fixed = (uint32_t *)malloc(sizeof(uint32_t) * EC_FIXED_WORDS);
        ec_build_fixed_tables(g_fixed);
    }
    const ec_spec S = {ecs_types, ecs_runs, ecs_items, ecs_item_index, ecs_cells, ecs_homes, ecs_outs, ecs_rowtab, ecs_globs, ecs_bigs, ecs_in_home, ecs_key_byte, g_fixed};
    return S;
}

uint32_t lib_ec_tape_per_cycle(void) { return EC_TAPE_PER_CYCLE; }
uint32_t lib_ec_ws_bytes(void) { return (uint32_t)sizeof(ec_ws); }

/* order 0: ec_eval_cycle. order 1: the MAIN part of every segment in program order, the MULS parts, then the lists of LEAVES (as small items only), last first. `ts`: the stride between consecutive values
   of the tape (the kernels interleave the cycles of an instance: value t at tape[t * ts]). Returns 0 or the evaluator's failure code */
uint32_t lib_ec_eval_cycle(const uint8_t *in, uint64_t *tape, int order, uint32_t ts) {
    const ec_spec S = spec();
    ec_ws W;
    memset(&W, 0xA5, sizeof W);
    if (order == 0) return ec_eval_cycle_strided(&S, in, tape, ts, &W);
    static const uint32_t parts[EC_NUM_TYPES][EC_MAX_PARTS] = EC_PART_ITEMS_INIT;
    ec_eval_ctx E;
    E.S = &S; E.tape = tape; E.ts = ts; E.in = in; E.W = &W;
    for (int phase = 0; phase < 3; phase++) /* the MAIN parts in program order, the MULS parts, then the lists of LEAVES, last first */
        for (uint32_t rr = 0; rr < EC_NUM_RUNS; rr++) {
            const uint32_t r = phase == 2 ? EC_NUM_RUNS - 1 - rr : rr, type = S.runs[r].type;
            for (uint32_t jj = 0; jj < S.runs[r].count; jj++) {
                const uint32_t j = phase == 2 ? S.runs[r].count - 1 - jj : jj;
                uint32_t prun, pinst;
                ec_prev_segment(&S, r, j, &prun, &pinst);
                E.base = S.runs[r].tape0 + j * S.types[type].n_tape;
                E.prev_base = S.runs[prun].tape0 + pinst * S.types[S.runs[prun].type].n_tape;
                E.prev_type = S.runs[prun].type;
                E.inst = j;
                for (int pp = 0; pp < EC_MAX_PARTS; pp++) {
                    const int p = phase == 2 ? EC_MAX_PARTS - 1 - pp : pp;
                    if ((p < 2 ? p : 2) != phase || parts[type][p] == 0) continue;
                    uint32_t first = 0;
                    for (int k = 0; k < p; k++) first += parts[type][k];
                    const int bad = ec_eval_items_of(&E, type, first, parts[type][p], phase == 2);
                    if (bad) return (r << 24) | (j << 12) | (uint32_t)bad;
                }
            }
        }
    return 0;
}
