/* examples/ram_block.c — the drop-in boundary used from plain C (no Python, no torch): one block's memory queue ->
 * RAMPermutation instance witnesses (compute_ram_circuit_snapshots, src/witness/individual_circuits/ram_permutation.rs:
 * 26-470) -> every instance synthesized into a 2^N-row trace (ZkSyncBaseLayerCircuit::synthesis, base_layer/mod.rs:
 * 286-323) -> check_if_satisfied (src/tests/mod.rs:130-259) -> public inputs and the recursion queue
 * (postprocessing/mod.rs:353-405). A Rust host does the same through the binding of INTEGRATION.md.
 *
 *   gcc -O2 -Iinclude examples/ram_block.c -o ram_block -Lera_zkevm_test_harness_amd -lzkw -Wl,-rpath,$PWD/era_zkevm_test_harness_amd
 *   ./ram_block [n_queries] [capacity] [log2_rows]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkw.h"

#define TRY(call) do { int _rc = (call); if (_rc != ZKW_OK) { fprintf(stderr, "%s: %s\n", #call, zkw_last_error()); return 1; } } while (0)

static uint64_t splitmix(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* a valid memory trace: the first access of a cell is a write, later ones read the current value 70 % of the time */
static void make_trace(zkw_mem_query *q, size_t n) {
    enum { PAGES = 16, CELLS = 64 };
    static uint32_t cur[PAGES][CELLS][8];
    static unsigned char live[PAGES][CELLS];
    uint64_t s = 7;
    memset(live, 0, sizeof live);
    for (size_t i = 0; i < n; i++) {
        const uint32_t page = (uint32_t)(splitmix(&s) % PAGES), idx = (uint32_t)(splitmix(&s) % CELLS);
        zkw_mem_query *m = q + i;
        memset(m, 0, sizeof *m);
        m->timestamp = (uint32_t)(i + 1);
        m->page = 8 + page;
        m->index = idx;
        const int read = live[page][idx] && splitmix(&s) % 10 < 7;
        if (!read) {
            for (int k = 0; k < 8; k++) cur[page][idx][k] = (uint32_t)splitmix(&s);
            live[page][idx] = 1;
        }
        m->rw_flag = read ? 0 : 1;
        memcpy(m->value, cur[page][idx], sizeof m->value);
    }
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 5000;
    const uint32_t capacity = argc > 2 ? (uint32_t)strtoul(argv[2], NULL, 10) : 1024;
    const size_t n_rows = (size_t)1 << (argc > 3 ? atoi(argv[3]) : 14);
    zkw_ctx *ctx = zkw_create(0); /* NULL (ZKW_ERR_NO_DEVICE) without a gfx950 GPU: there is no CPU fallback */
    if (!ctx) { fprintf(stderr, "zkw_create: %s\n", zkw_last_error()); return 1; }
    zkw_mem_query *q = (zkw_mem_query *)malloc(n * sizeof *q);
    make_trace(q, n);

    zkw_ram_witness *w = NULL;
    TRY(zkw_ram_build_instances(ctx, q, n, capacity, 0, &w));
    const size_t n_inst = zkw_ram_witness_num_instances(w);
    zkw_ram_instance *inst = (zkw_ram_instance *)malloc(n_inst * sizeof *inst);
    uint64_t *pi = (uint64_t *)malloc(n_inst * 4 * sizeof *pi);
    TRY(zkw_ram_witness_get(w, ZKW_RAM_INSTANCES, inst, n_inst * sizeof *inst));
    TRY(zkw_ram_witness_get(w, ZKW_RAM_PUBLIC_INPUTS, pi, n_inst * 4 * sizeof *pi));
    printf("%zu queries -> %zu RAMPermutation instances of capacity %u\n", n, n_inst, capacity);

    zkw_trace *t = NULL;
    TRY(zkw_trace_create(ctx, n_rows, 1, &t));
    for (size_t i = 0; i < n_inst; i++) {
        uint64_t bad = 0, first = 0;
        TRY(zkw_synthesize(ctx, ZKW_CIRCUIT_RAM_PERMUTATION, w, i, 1, t, 0)); /* ZkSyncBaseLayerCircuit::synthesis: one entry point over the circuit types */
        TRY(zkw_check_satisfied(ctx, ZKW_CIRCUIT_RAM_PERMUTATION, t, 0, capacity, &bad, &first));
        printf("  instance %zu: items [%llu, +%llu) start %u completion %u  public input %016llx..  %s\n", i,
               (unsigned long long)inst[i].first_item, (unsigned long long)inst[i].num_items, inst[i].start_flag,
               inst[i].completion_flag, (unsigned long long)pi[4 * i], bad ? "NOT SATISFIED" : "satisfied");
        if (bad) return 2;
    }
    /* the recursion queue of circuit type 8 (RecursionQueueSimulator pushes, postprocessing/mod.rs:393-400) */
    uint64_t *enc = (uint64_t *)malloc(n_inst * 8 * sizeof *enc), *tails = (uint64_t *)malloc(n_inst * 12 * sizeof *tails);
    TRY(zkw_encode_recursion_requests(ctx, 8, pi, n_inst, enc));
    TRY(zkw_queue_push_chain_full(ctx, enc, n_inst, NULL, tails));
    printf("recursion queue tail after %zu pushes: %016llx %016llx %016llx %016llx ..\n", n_inst,
           (unsigned long long)tails[12 * (n_inst - 1)], (unsigned long long)tails[12 * (n_inst - 1) + 1],
           (unsigned long long)tails[12 * (n_inst - 1) + 2], (unsigned long long)tails[12 * (n_inst - 1) + 3]);
    zkw_trace_free(t);
    zkw_ram_witness_free(w);
    zkw_destroy(ctx);
    free(q); free(inst); free(pi); free(enc); free(tails);
    printf("ok\n");
    return 0;
}
