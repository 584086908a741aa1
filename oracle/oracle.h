/* oracle.h — TEST INFRASTRUCTURE. CPU restatement (plain C11, gcc) of the reference's hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libzkw / csrc) never includes, links or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"): the reference is Rust and cannot be built here; the field,
 * Poseidon2 and sponge helpers live in the absent crate era-boojum. Each function below cites the
 * reference file:line it follows. Pinned against reference fixtures: SHA-256/Keccak/Blake2s (public
 * KATs) and the round-constant table (Poseidon-Goldilocks KAT). The Poseidon2 permutation's linear
 * layers are "parity unpinned" (no fixture in the reference reaches them; the Merkle-path KAT
 * harvested from test_proofs/ is kept as tests/golden/merkle_kat_*.json and reported by
 * tests/test_reference_fixtures.py).
 */
#ifndef ZKW_ORACLE_H
#define ZKW_ORACLE_H
#include "../include/zkw_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Goldilocks field (boojum GoldilocksField; call sites circuit_encodings/src/lib.rs:664-713) */
/* field ops: any u64 in, canonical out. The *_ref forms are the one-`%` definitions the fast forms are tested against. */
uint64_t orc_gl_reduce128(unsigned __int128 w);
uint64_t orc_gl_add_ref(uint64_t a, uint64_t b);
uint64_t orc_gl_sub_ref(uint64_t a, uint64_t b);
uint64_t orc_gl_mul_ref(uint64_t a, uint64_t b);
uint64_t orc_gl_add(uint64_t a, uint64_t b);
uint64_t orc_gl_sub(uint64_t a, uint64_t b);
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_pow(uint64_t a, uint64_t e);
uint64_t orc_gl_inv(uint64_t a);
/* The same three ops as static inline functions: inside the oracle's own translation units calls to orc_gl_add / _sub /
 * _mul resolve to these (a call through the PLT per field operation costs more than the operation). */
#define ORC_GL_P 0xFFFFFFFF00000001ULL
#define ORC_GL_EPS 0xFFFFFFFFULL
static inline uint64_t orc_gl_reduce128_inl(unsigned __int128 w) {
    const uint64_t lo = (uint64_t)w, hi = (uint64_t)(w >> 64), hh = hi >> 32, hl = hi & ORC_GL_EPS;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= ORC_GL_EPS; /* wrapped by 2^64 = EPS */
    const uint64_t t1 = hl * ORC_GL_EPS;
    uint64_t r = t0 + t1;
    if (r < t1) r += ORC_GL_EPS;
    return r >= ORC_GL_P ? r - ORC_GL_P : r;
}
static inline uint64_t orc_gl_add_inl(uint64_t a, uint64_t b) { return orc_gl_reduce128_inl((unsigned __int128)a + b); }
static inline uint64_t orc_gl_sub_inl(uint64_t a, uint64_t b) {
    if (a >= ORC_GL_P) a -= ORC_GL_P;
    if (b >= ORC_GL_P) b -= ORC_GL_P;
    return a >= b ? a - b : a + (ORC_GL_P - b);
}
static inline uint64_t orc_gl_mul_inl(uint64_t a, uint64_t b) { return orc_gl_reduce128_inl((unsigned __int128)a * b); }
#ifndef ORC_FIELD_EXPORTS
#define orc_gl_reduce128 orc_gl_reduce128_inl
#define orc_gl_add orc_gl_add_inl
#define orc_gl_sub orc_gl_sub_inl
#define orc_gl_mul orc_gl_mul_inl
#endif


/* ---- Poseidon2Goldilocks as AlgebraicRoundFunction<F, 8, 12, 4> (lib.rs:12-15) */
void orc_poseidon2_permutation(uint64_t state[12]);
/* the same permutation in its obvious form (one reduction per field operation): cross-check of the fast form */
void orc_poseidon2_permutation_ref(uint64_t state[12]);
/* Poseidon (original, Plonky2-compatible) — used only to pin the shared round-constant table */
/* absorb_multiple_rounds::<AbsorptionModeOverwrite> (call sites lib.rs:198-203, 405-409):
   for each 8-chunk: state[0..8] = chunk; permutation; record the state. `states_out` may be NULL. */
void orc_absorb_multiple_rounds(uint64_t state[12], const uint64_t *to_absorb, size_t n_rounds,
                                uint64_t *states_out /* n_rounds*12 */);
/* Merkle helpers matching GoldilocksPoseidon2Sponge<AbsorptionModeOverwrite> as TreeHasher
   (src/prover_utils.rs:43) — used by the reference-fixture KAT only. */
void orc_poseidon2_hash_node(const uint64_t left[4], const uint64_t right[4], uint64_t out[4]);
void orc_poseidon2_hash_leaf(const uint64_t *elems, size_t n, uint64_t out[4]);

/* ---- encodings */
/* circuit_encodings/src/memory_query.rs:24-118 */
void orc_encode_memory_query(const zkw_mem_query *q, uint64_t out[8]);
void orc_encode_memory_queries(const zkw_mem_query *q, size_t n, uint64_t *out /* n*8 */);

/* circuit_encodings/src/log_query.rs:102-396; ext_ts != NULL adds the extended timestamp
   (LogQueryWithExtendedEnumeration, log_query.rs:400-427). out: n*20 */
void orc_encode_log_queries(const zkw_log_query *q, size_t n, const uint32_t *ext_ts, uint64_t *out);
/* circuit_encodings/src/decommittment_request.rs:9-74. out: n*8 */
void orc_encode_decommit_queries(const zkw_decommit_query *q, size_t n, uint64_t *out);
/* circuit_encodings/src/recursion_request.rs:13-28: [circuit_type, pi0..pi3, 0, 0, 0] */
void orc_encode_recursion_request(uint64_t circuit_type, const uint64_t pi[4], uint64_t out[8]);

/* ---- queue simulators */
/* FullWidthQueueSimulator::push_and_output_intermediate_data, lib.rs:391-429: tails[i] = state after
   absorbing enc[i] (rate 8, overwrite) into tails[i-1] (tail_in for i = 0). */
void orc_queue_push_chain_full(const uint64_t *enc /* n*8 */, size_t n, const uint64_t tail_in[12],
                               uint64_t *tails /* n*12 */);
/* QueueSimulator::push_and_output_intermediate_data, lib.rs:179-221: new_tail = first 4 words of the
   3-round sponge over enc(20) || old_tail(4) from the zero state. old_tails[i] is what the reference
   stores in `witness` (lib.rs:204). */
void orc_queue_push_chain_log(const uint64_t *enc /* n*20 */, size_t n, const uint64_t tail_in[4],
                              uint64_t *old_tails /* n*4, may be NULL */, uint64_t *new_tails /* n*4 */);

/* ---- Fiat-Shamir challenges, src/witness/utils.rs:498-550.
   state_w = N (12 for RAM/decommit sorter, 4 for storage/events); out = 2 repetitions x n_chal. */
void orc_fs_challenges(const uint64_t *tail_u, uint32_t len_u, const uint64_t *tail_s, uint32_t len_s,
                       int state_w, int n_chal, uint64_t *out /* 2*n_chal */);

/* ---- grand product chains, src/witness/utils.rs:554-697 (chunked by 2^16 exactly like the rayon
   version, then folded). Returns 0, or -1 if the final lhs and rhs products differ (utils.rs:685-696). */
int orc_grand_product_chains(const uint64_t *lhs, const uint64_t *rhs, size_t n, int width,
                             const uint64_t *challenges /* width+1 */, uint64_t *lhs_z, uint64_t *rhs_z);
/* multi-threaded variant (pthreads; mirrors rayon par_chunks) used by bench.py's cpu_baseline */
int orc_grand_product_chains_mt(const uint64_t *lhs, const uint64_t *rhs, size_t n, int width,
                                const uint64_t *challenges, uint64_t *lhs_z, uint64_t *rhs_z, int threads);

/* ---- RAM permutation builder, src/witness/individual_circuits/ram_permutation.rs:26-470.
   Inputs: the block's memory queries in queue order. Outputs (caller-allocated):
     sorted_q[n], unsorted_enc[n*8], sorted_enc[n*8], unsorted_tails[n*12], sorted_tails[n*12],
     challenges[2*9], lhs_z[2*n], rhs_z[2*n] (repetition-major), instances[ceil(n/capacity)].
   Returns the number of instances, or <0 on a failed self-check. */
int64_t orc_ram_build_instances(const zkw_mem_query *q, size_t n, uint32_t capacity,
                                uint32_t num_non_deterministic_heap_queries, zkw_mem_query *sorted_q,
                                uint64_t *unsorted_enc, uint64_t *sorted_enc, uint64_t *unsorted_tails,
                                uint64_t *sorted_tails, uint64_t *challenges, uint64_t *lhs_z,
                                uint64_t *rhs_z, zkw_ram_instance *instances);

/* ---- decommit sorter builder, src/witness/individual_circuits/sort_decommit_requests.rs:20-420.
   q: the block's decommit requests in queue order (n > 0). dedup_in: state of the deduplicated queue
   before the call (NULL = empty). Outputs (caller-allocated): sorted_q[n], unsorted/sorted enc [n*8] and
   tails [n*12], dedup_q/dedup_enc/dedup_tails sized for n (first *n_dedup used), challenges [2*9],
   lhs_z/rhs_z [2*n], instances [ceil(n/capacity)]. Returns the number of instances or <0 when one of the
   reference's asserts fails. */
int64_t orc_decommit_sorter_build(const zkw_decommit_query *q, size_t n, uint32_t capacity,
                                  const zkw_queue_state12 *dedup_in, zkw_decommit_query *sorted_q,
                                  uint64_t *unsorted_enc, uint64_t *sorted_enc, uint64_t *unsorted_tails,
                                  uint64_t *sorted_tails, zkw_decommit_query *dedup_q, uint64_t *dedup_enc,
                                  uint64_t *dedup_tails, uint64_t *n_dedup, uint64_t *challenges,
                                  uint64_t *lhs_z, uint64_t *rhs_z, zkw_decommit_sorter_instance *instances);

/* ---- events / L1-messages sorter builder, src/witness/individual_circuits/events_sort_dedup.rs:16-580.
   q: the demuxed event (or L1 message) log queue in queue order; result_in: state of the result queue
   before the call (NULL = empty). Outputs sized for n (n_result entries used in result_*). For n == 0 one
   dummy instance is produced (events_sort_dedup.rs:27-76). Returns the number of instances or <0. */
int64_t orc_events_sorter_build(const zkw_log_query *q, size_t n, uint32_t capacity, const zkw_queue_state4 *result_in,
                                zkw_log_query *sorted_q, uint64_t *unsorted_enc, uint64_t *sorted_enc,
                                uint64_t *unsorted_old_tails, uint64_t *unsorted_new_tails,
                                uint64_t *sorted_old_tails, uint64_t *sorted_new_tails, zkw_log_query *result_q,
                                uint64_t *result_enc, uint64_t *result_new_tails, uint64_t *n_result,
                                uint64_t *challenges /* [2][21] */, uint64_t *lhs_z, uint64_t *rhs_z,
                                zkw_events_sorter_instance *instances);

/* ---- log demuxer builder, src/witness/individual_circuits/log_demux.rs:20-388.
   q: the original (forward-applied) log queue in order. Outputs: in_enc [n][20], in_old/new tails [n][4];
   the six demuxed queues back to back in route order 0..5: out_q / out_enc / out_old_tails / out_new_tails
   sized for n, queue k occupying [out_offsets[k], out_offsets[k+1]) (out_offsets: 7 entries);
   instances [max(1, ceil(n/capacity))]. Returns the number of instances or <0. */
int64_t orc_log_demux_build(const zkw_log_query *q, size_t n, uint32_t capacity, const zkw_demux_params *params,
                            uint64_t *in_enc, uint64_t *in_old_tails, uint64_t *in_new_tails, zkw_log_query *out_q,
                            uint64_t *out_enc, uint64_t *out_old_tails, uint64_t *out_new_tails, uint64_t *out_offsets,
                            zkw_log_demux_instance *instances);

/* ---- storage sorter builder: sort_storage_access_queries (circuit_sequencer_api/src/sort_storage_access.rs:19-260)
   + compute_storage_dedup_and_sort (src/witness/individual_circuits/storage_sort_dedup.rs:12-703).
   q: the demuxed rollup storage queue in order. Outputs sized for n: sorted_q / sorted_ext_ts (= position in
   the unsorted queue), unsorted_enc (plain, what the queue hashes), lhs_enc (with extended timestamp, the
   permutation-argument side), sorted_enc (with extended timestamp), old/new tails of both queues, the
   deduplicated queries + encodings + new tails (n_result used), challenges [2][21], chains, instances
   [max(1, ceil(n/capacity))]. Returns the number of instances or <0 when an assert of the reference fails. */
int64_t orc_storage_sorter_build(const zkw_log_query *q, size_t n, uint32_t capacity, zkw_log_query *sorted_q,
                                 uint32_t *sorted_ext_ts, uint64_t *unsorted_enc, uint64_t *lhs_enc, uint64_t *sorted_enc,
                                 uint64_t *unsorted_old_tails, uint64_t *unsorted_new_tails, uint64_t *sorted_old_tails,
                                 uint64_t *sorted_new_tails, zkw_log_query *result_q, uint64_t *result_enc,
                                 uint64_t *result_new_tails, uint64_t *n_result, uint64_t *challenges, uint64_t *lhs_z,
                                 uint64_t *rhs_z, zkw_storage_sorter_instance *instances);

/* ---- code decommitter builder, src/witness/individual_circuits/decommit_code.rs:20-439.
   requests: the deduplicated (fresh) decommit requests in queue order with their deduplicated-queue tails
   (dedup_tails[k] = queue state after request k, as produced by the decommit sorter from an EMPTY queue);
   words: all bytecode words back to back (8 LE u32 limbs each), request k owning [word_offsets[k],
   word_offsets[k+1]); mem_in: state of the global memory queue before the call. Outputs: mem_q / mem_enc /
   mem_tails (one memory query per word, appended to the memory queue), round_states [total_rounds][8],
   instances [ceil(total_rounds/capacity)]. Returns the number of instances or <0 (e.g. -5 = a bytecode does
   not hash to its request, decommit_code.rs:323-337). */
int64_t orc_decommitter_build(const zkw_decommit_query *requests, const uint64_t *dedup_tails, size_t n_requests,
                              const uint32_t *words, const uint64_t *word_offsets, uint32_t capacity,
                              const zkw_queue_state12 *mem_in, zkw_mem_query *mem_q, uint64_t *mem_enc,
                              uint64_t *mem_tails, uint32_t *round_states, zkw_decommitter_instance *instances);
/* helper for tests/benches: SHA-256 based versioned hash of a bytecode the way the decommitter checks it:
   digest of the words (each big-endian) with the 4 most significant bytes replaced by `top4` */
void orc_bytecode_hash(const uint32_t *words, size_t n_words, uint32_t top_limb, uint32_t hash_out[8]);

/* ---- CodeDecommittmentsSorter synthesis + check (a21, second circuit type), see decommit_sorter_circuit.c */
/* sorted_q / *_enc: the block-wide arrays of orc_decommit_sorter_build; the fill carries every register itself, so
   the trace's BND_OUT row is an independent re-derivation of the instance's hidden_fsm_output */
void orc_poseidon2_flattened(const uint64_t in[12], uint64_t slots[130]);
int orc_decommit_sorter_synthesize(const zkw_decommit_sorter_instance *inst, const zkw_decommit_query *sorted_q,
                                   const uint64_t *unsorted_enc, const uint64_t *sorted_enc, const uint64_t *challenges,
                                   const uint64_t *rq_tail_in, uint32_t rq_len_in, const uint64_t *public_input,
                                   uint32_t capacity, size_t n_rows, uint64_t *trace);
uint64_t orc_decommit_sorter_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);

/* ---- EventsSorter / L1MessagesSorter synthesis + check (a21, circuit types 11 and 12), see events_sorter_circuit.c */
int orc_events_sorter_synthesize(const zkw_events_sorter_instance *inst, const zkw_log_query *sorted_q, const uint64_t *unsorted_enc,
                                 const uint64_t *sorted_enc, const uint64_t *challenges, const uint64_t *rq_tail_in,
                                 uint32_t rq_len_in, const uint64_t *public_input, uint32_t capacity, size_t n_rows, uint64_t *trace);
uint64_t orc_events_sorter_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
/* log_demux_circuit.c: LogDemuxer synthesis (circuit type 4) */
int orc_log_demux_synthesize(const zkw_log_demux_instance *inst, const uint64_t *in_enc, const uint64_t *public_input,
                             uint32_t capacity, size_t n_rows, uint64_t *trace);
void orc_log_demux_public_inputs(const zkw_log_demux_instance *inst, size_t n, uint64_t *compact, uint64_t *pi);
void orc_events_sorter_public_inputs(const zkw_events_sorter_instance *inst, size_t n, uint64_t *compact, uint64_t *pi);
void orc_storage_sorter_public_inputs(const zkw_storage_sorter_instance *inst, size_t n, uint64_t *compact, uint64_t *pi);
/* closed-form commitments of the circuits 3, 5, 6, 7, 10, 13 (instances: the type's zkw_*_instance records); -1 = unknown type */
#define ORC_CF_MAX_FSM_LEN 448
int orc_closed_form_public_inputs(int circuit_type, const void *instances, size_t n, uint64_t *compact, uint64_t *pi);
int orc_cf_encode(int circuit_type, const void *instances, size_t i, uint64_t *w /* 4 x ORC_CF_MAX_FSM_LEN */, size_t n[4], int flags[2]);
/* the closed-form section of the netlist circuits (netlist_closed_form.c; include/zkw_netlist_closed_form.h) */
int orc_nlcf_fill(int circuit_type, const void *instances, size_t i, uint32_t cycles, size_t n_rows, uint64_t *trace);
int orc_nlcf_standalone(int circuit_type, uint32_t cycles, size_t n_rows, uint64_t *trace);
uint64_t orc_nlcf_check(int circuit_type, const uint64_t *trace, uint32_t cycles, size_t n_rows, uint64_t *first_bad);
void orc_nlcf_geometry(int circuit_type, uint32_t cycles, uint64_t out[9]);
int orc_nlcf_cell(int circuit_type, uint32_t cycles, int what, uint32_t k, uint64_t out[2]);
uint64_t orc_log_demux_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
/* storage_sorter_circuit.c: StorageSorter synthesis (circuit type 9) */
int orc_storage_sorter_synthesize(const zkw_storage_sorter_instance *inst, const uint64_t *unsorted_enc, const uint64_t *sorted_enc,
                                  const uint64_t *challenges, const uint64_t *public_input, uint32_t capacity, size_t n_rows,
                                  uint64_t *trace);
uint64_t orc_storage_sorter_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);

/* ---- sparse storage tree + StorageApplication builder (a17), see storage_application.c */
typedef struct orc_tree orc_tree;
orc_tree *orc_tree_new(void);
void orc_tree_free(orc_tree *t);
void orc_tree_root(const orc_tree *t, uint8_t out[32]);
uint64_t orc_tree_next_enumeration_index(const orc_tree *t);
void orc_tree_get_leaf(orc_tree *t, const uint8_t key[32], uint64_t *leaf_index, uint8_t value[32], uint8_t *path);
uint64_t orc_tree_insert_leaf(orc_tree *t, const uint8_t key[32], const uint8_t value[32], uint8_t *path);
int orc_tree_verify_inclusion(const uint8_t root[32], const uint8_t key[32], uint64_t leaf_index, const uint8_t value[32],
                              const uint8_t *path);
void orc_derive_final_address(const zkw_log_query *q, uint8_t out[32]);
void orc_state_diff_encode(const zkw_log_query *q, const uint8_t derived_key[32], uint64_t enumeration_index, uint8_t out[156]);
int64_t orc_storage_application_build(orc_tree *tree, const zkw_log_query *queries, const uint64_t *query_tails, size_t n,
                                      uint32_t capacity, uint8_t *derived_keys, uint8_t *merkle_paths, uint64_t *leaf_indexes,
                                      uint8_t *roots, zkw_storage_application_instance *instances);

/* ---- keccak256 / sha256 / ecrecover round-function builders (a16), see precompiles.c */
int64_t orc_precompile_build(int kind, const zkw_log_query *requests, const uint64_t *req_tails, size_t n_req,
                             const zkw_mem_query *mem_q, size_t n_q, uint32_t capacity, const zkw_queue_state12 *mem_in,
                             uint64_t *mem_enc, uint64_t *mem_tails, zkw_precompile_instance *instances);
int64_t orc_precompile_build_ex(int kind, const zkw_log_query *requests, const uint64_t *req_tails, size_t n_req,
                                const zkw_mem_query *mem_q, size_t n_q, uint32_t capacity, const zkw_queue_state12 *mem_in,
                                uint64_t *mem_enc, uint64_t *mem_tails, zkw_precompile_instance *instances,
                                zkw_keccak_round_record *keccak_rounds);
/* ---- the netlist circuits ("zkw trace v4", include/zkw_netlist.h), netlist_circuit.c: one fill, one checker, four specs */
#include "../include/zkw_netlist.h"
const nl_spec *orc_nl_spec(int circuit_type); /* 6, 3, 5, 13, 10 */
/* netlist_tables.c: the tables' contents enumerated by the oracle itself (no code shared with the library's evaluator) */
void orc_nl_lookup(const nl_table *t, const uint32_t a[3], uint32_t out[3]);
uint32_t orc_nl_multiplicity_row(const nl_table *t, const uint32_t a[3]);
int orc_nl_synthesize(const nl_spec *sp, uint32_t capacity, const uint8_t *hdr_bits, const uint8_t *free_elems, const uint8_t *state_before,
                      const uint64_t pi[4], size_t n_rows, uint64_t *trace);
uint64_t orc_nl_check(const nl_spec *sp, const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
void orc_nl_geometry(int circuit_type, uint32_t out[6]);
void orc_nl_slots_per_cycle(int circuit_type, uint32_t *out);
int orc_linear_hasher_round_synthesize(const uint8_t state_in[200], const zkw_keccak_round_record *rounds, uint32_t n_active,
                                       uint32_t capacity, const uint64_t pi[4], size_t n_rows, uint64_t *trace);
uint64_t orc_linear_hasher_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
int orc_keccak_round_synthesize(const uint8_t state_in[200], const zkw_keccak_round_record *rounds, uint32_t n_active,
                                uint32_t capacity, const uint64_t pi[4], size_t n_rows, uint64_t *trace);
uint64_t orc_keccak_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
/* ---- Sha256RoundFunction circuit (include/zkw_sha256_circuit_spec.h), sha256_circuit.c */
void orc_precompile_set_sha256_rounds(zkw_sha256_round_record *r);
int orc_sha256_round_synthesize(const uint8_t state_in[32], const zkw_sha256_round_record *rounds, uint32_t n_active,
                                uint32_t capacity, const uint64_t pi[4], size_t n_rows, uint64_t *trace);
uint64_t orc_sha256_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
/* ---- CodeDecommitter circuit (include/zkw_code_decommitter_circuit_spec.h), code_decommitter_circuit.c (generated) */
void orc_decommitter_set_sha256_rounds(zkw_sha256_round_record *r);
int orc_code_decommitter_round_synthesize(const uint8_t state_in[32], const zkw_sha256_round_record *rounds, uint32_t n_active,
                                          uint32_t capacity, const uint64_t pi[4], size_t n_rows, uint64_t *trace);
uint64_t orc_code_decommitter_round_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
int orc_storage_application_synthesize(const zkw_log_query *items, size_t n_items, const uint8_t *keys, const uint8_t *paths,
                                       const uint64_t *read_indexes, uint64_t next_enumeration_index, uint32_t capacity,
                                       const uint64_t pi[4], size_t n_rows, uint64_t *trace, const uint8_t *idle_root);
uint64_t orc_storage_application_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
size_t orc_linear_hasher_rounds(const zkw_log_query *q, size_t n, zkw_keccak_round_record *records);

/* ---- callstack (a3 / a6), see callstack.c */
void orc_encode_callstack_entry(const zkw_callstack_entry *e, uint64_t out[32]);
void orc_encode_callstack_entries(const zkw_callstack_entry *e, size_t n, uint64_t *out /* n*32 */);
int orc_callstack_simulate(const uint8_t *is_push, size_t n_ops, const zkw_callstack_entry *pushed, size_t n_pushed,
                           uint64_t *previous_state, uint64_t *new_state, uint32_t *depth, uint64_t *round_states,
                           uint32_t *entry_index);

/* ---- public-input commitment (a20), see public_input.c */
#define ORC_RAM_INPUT_ENC_LEN 51
#define ORC_RAM_FSM_ENC_LEN 69
void orc_commit_var_length(const uint64_t *enc, size_t n, uint64_t out[4]);
/* ---- recursion layer witnesses, recursion.c (src/witness/recursive_aggregation.rs) */
void orc_vk_commitment(const uint64_t *cap, size_t cap_size, uint64_t out[4]);
void orc_leaf_params(uint8_t circuit_type, const uint64_t *base_cap, const uint64_t *leaf_cap, size_t cap_size, zkw_leaf_params *out);
void orc_leaf_vks_and_params_commitment(const zkw_leaf_params *p, uint64_t out[4]);
void orc_leaf_public_input(const zkw_leaf_params *params, const zkw_queue_state12 *queue_state, uint64_t out[4]);
int orc_node_witness(uint8_t branch_circuit_type, const zkw_leaf_params *leaf_layer_params, const uint64_t node_vk_commitment[4],
                     const zkw_queue_state12 *chunks, size_t n_chunks, zkw_queue_state12 *node_state, zkw_queue_tail12 *split_points,
                     uint64_t public_input[4]);
size_t orc_ram_encode_observable_input(const zkw_ram_instance *in, uint64_t out[ORC_RAM_INPUT_ENC_LEN]);
size_t orc_ram_encode_fsm(const zkw_ram_fsm *f, uint64_t out[ORC_RAM_FSM_ENC_LEN]);
void orc_ram_public_input(const zkw_ram_instance *first, const zkw_ram_instance *in, uint64_t compact[18],
                          uint64_t pi[4]);
void orc_ram_public_inputs(const zkw_ram_instance *inst, size_t n, uint64_t *compact, uint64_t *pi);
void orc_ram_fill_public_input(const zkw_ram_instance *first, const zkw_ram_instance *in, uint32_t capacity,
                               size_t n_rows, uint64_t *trace);
/* the closed-form sections of the other queue circuits (closed_form_fill.c): after orc_*_synthesize; `first` = the block's first instance */
size_t orc_put_queue12(const zkw_queue_state12 *q, uint64_t *o);
size_t orc_put_queue4(const zkw_queue_state4 *q, uint64_t *o);
size_t orc_es_fsm(const zkw_events_sorter_fsm *f, uint64_t *o); /* 68 words */
size_t orc_ss_fsm(const zkw_storage_sorter_fsm *f, uint64_t *o); /* 77 words */
size_t orc_ld_fsm(const zkw_log_demux_fsm *f, uint64_t *o);      /* 63 words */
void orc_ld_fill_closed_form(const zkw_log_demux_instance *first, const zkw_log_demux_instance *in, uint32_t capacity, size_t n_rows,
                             uint64_t *trace);
void orc_ss_fill_closed_form(const zkw_storage_sorter_instance *first, const zkw_storage_sorter_instance *in, uint32_t capacity,
                             size_t n_rows, uint64_t *trace);
void orc_es_fill_closed_form(const zkw_events_sorter_instance *first, const zkw_events_sorter_instance *in, uint32_t capacity,
                             size_t n_rows, uint64_t *trace);
void orc_ds_fill_closed_form(const zkw_decommit_sorter_instance *first, const zkw_decommit_sorter_instance *in, uint32_t capacity,
                             size_t n_rows, uint64_t *trace);
#define ORC_DS_FSM_ENC_LEN 100
size_t orc_ds_encode_fsm(const zkw_decommit_sorter_fsm *f, uint64_t out[ORC_DS_FSM_ENC_LEN]);
void orc_ds_public_inputs(const zkw_decommit_sorter_instance *inst, size_t n, uint64_t *compact, uint64_t *pi);
void orc_recursion_queue(uint64_t circuit_type, const uint64_t *pi, size_t n, const uint64_t tail_in[12],
                         uint64_t *enc, uint64_t *tails);

/* ---- L1 messages hasher, src/witness/individual_circuits/data_hasher_and_merklizer.rs:8-67:
   Keccak256 over the concatenated 88-byte serialisations (circuit_encodings/src/log_query.rs:503-534:
   shard | is_service | tx_number BE u16 | address 20 BE | key 32 BE | written_value 32 BE) */
void orc_serialize_l1_message(const zkw_log_query *q, uint8_t out[88]);
void orc_linear_keccak256(const zkw_log_query *q, size_t n, uint8_t hash_out[32]);

/* ---- the queue section of the netlist circuits (netlist_queue.c; format: include/zkw_netlist_queue.h) */
/* ---- ECRecover circuit (type 7), ecrecover_circuit.c ---- */
struct ec_spec;
const struct ec_spec *orc_ec_spec(void);
uint64_t orc_ec_first_row(uint32_t capacity);
uint64_t orc_ec_used_rows(uint32_t capacity);
void orc_ec_geometry(uint32_t capacity, uint64_t out[8]);
uint32_t orc_ec_eval_cycle(const uint8_t in[128], uint64_t *tape);
void orc_ec_outputs(const uint64_t *tape, uint8_t out[66]);
int orc_ec_cell(int what, uint32_t k, uint32_t out[2]);
int orc_ecrecover_synthesize(const uint8_t *inputs, uint32_t n_active, uint32_t capacity, const uint64_t pi[4], size_t n_rows, uint64_t *trace);
uint64_t orc_ec_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint32_t *hist, uint64_t *first_bad);
uint64_t orc_ecrecover_check(const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
struct nlq_feed;
void orc_ecrecover_queue_feed(size_t first_round, uint32_t n_active, uint32_t capacity, struct nlq_feed *feed);
int orc_nl_free_home(const nl_spec *sp, uint32_t free_index, uint32_t *row_in_cycle, uint32_t *col);

typedef struct orc_nlq_queue {
    const void *items;      /* zkw_log_query / zkw_mem_query / zkw_decommit_query [n_items] */
    const uint64_t *states; /* [n_items][width]: the queue state after item i (tail after the push / head after the pop) */
    const uint64_t *init;   /* [width]: the state before item 0; NULL = zeros */
    size_t n_items;
} orc_nlq_queue;
struct nlq_feed;
int orc_nlq_synthesize(int circuit_type, uint32_t capacity, const struct nlq_feed *feed, const orc_nlq_queue *queues, size_t n_rows, uint64_t *trace);
uint64_t orc_nlq_check(int circuit_type, const uint64_t *trace, uint32_t capacity, size_t n_rows, uint64_t *first_bad);
int orc_nlq_standalone(int circuit_type, const zkw_sha256_round_record *rounds, uint32_t n_active, uint32_t capacity, size_t n_rows, uint64_t *trace);
int orc_nlq_standalone_keccak(int circuit_type, const zkw_keccak_round_record *rounds, uint32_t n_active, uint32_t capacity, size_t n_rows, uint64_t *trace);
void orc_keccak_queue_feed(const zkw_log_query *requests, size_t n_req, size_t first_round, uint32_t n_active, uint32_t capacity, struct nlq_feed *feed);
void orc_linear_hasher_queue_feed(size_t n_messages, uint32_t cycles, struct nlq_feed *feed);
int orc_linear_hasher_queue_section(const zkw_log_query *messages, size_t n, const uint64_t *head, uint32_t cycles, size_t n_rows, uint64_t *trace);
void orc_sha256_queue_feed(const zkw_sha256_round_record *rounds, size_t total_rounds, size_t first_round, uint32_t n_active, uint32_t capacity, struct nlq_feed *feed);
void orc_code_decommitter_queue_feed(const zkw_sha256_round_record *rounds, size_t total_rounds, const uint64_t *word_offsets, size_t first_round,
                                     uint32_t n_active, uint32_t capacity, struct nlq_feed *feed);

#ifdef __cplusplus
}
#endif
/* MainVM instance slicing (src/witness/oracle.rs:1229-1469, src/witness/utils.rs:428-496); host pointers */
int orc_vm_slice_instances(const zkw_vm_tracer_streams *s, zkw_vm_instance *out, uint32_t *read_index, uint32_t *write_index,
                           uint64_t *n_reads, uint64_t *n_writes);

/* ---- the setup side as field elements (commit.c): NTT, LDE, Merkle tree with a cap */
uint64_t orc_root_of_unity(uint32_t log_n);
void orc_gl_powers(uint64_t base, size_t n, uint64_t *out);
void orc_ntt(uint64_t *x, uint32_t log_n, int inverse);
void orc_lde(const uint64_t *values, uint32_t log_n, size_t n_cols, uint32_t lde_factor, uint64_t *out);
uint64_t orc_poly_eval(const uint64_t *coeffs, size_t n, uint64_t x);
void orc_merkle_tree_with_cap(const uint64_t *leaf_cols, size_t n_sets, size_t n_cols, size_t n, uint32_t cap_size, uint64_t *tree);

#endif
