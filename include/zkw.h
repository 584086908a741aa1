/* zkw.h — C ABI of libzkw, the MI355X-native witness-generation / constraint-synthesis engine for the
 * zkSync Era base-layer circuits (hot path of matter-labs/era-zkevm_test_harness).
 *
 * This is the drop-in boundary: a Rust host keeps `circuit_sequencer_api`, `external_calls::run` and
 * `prover_utils` and binds these symbols with `extern "C"`. Every entry point
 * names the reference function it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - status: 0 = ok, < 0 = error; text via zkw_last_error() (thread-local). Nothing panics or throws
 *     across the boundary (the reference panics/asserts: src/witness/utils.rs:654-696).
 *   - field elements: canonical (< p) little-endian uint64_t, p = 2^64 - 2^32 + 1.
 *   - bulk pointers are HOST pointers by default; after zkw_set_pointer_mode(ctx, ZKW_PTR_DEVICE) they
 *     are device (HBM) pointers and no staging copy is made. Small descriptor arrays documented as
 *     "host" are always host pointers.
 *   - all work is enqueued on the context's HIP stream (zkw_set_stream); calls that return data to
 *     host memory synchronise that stream before returning, device-mode calls do not.
 *   - there is no CPU fallback: every call fails with ZKW_ERR_NO_DEVICE when no gfx950 device is usable.
 *   - lifetimes: every witness / trace / block object keeps a reference to the context it was created from (its
 *     accessors and its free function use the context's device and stream). zkw_destroy on a context with outstanding
 *     objects synchronises and marks it; the context is released by the last zkw_*_free. The context must not be
 *     passed to any other call after zkw_destroy. A zkw_*_free synchronises its OWN context's stream; work that another
 *     context (or the host's own streams) still has in flight on the object must be synchronised by the caller first.
 *   - memory: device and pinned buffers and the library's streams are recycled through per-device
 *     caches instead of going back to the HIP runtime (hipFree / hipHostFree / hipStreamDestroy wait for the whole
 *     device, which stalls every other context); zkw_trim_caches() hands everything idle back, an allocation failure
 *     does so by itself and retries, ZKW_ALLOC_CACHE=0 in the environment turns the caches off.
 */
#ifndef ZKW_H
#define ZKW_H
#include "zkw_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkw_ctx zkw_ctx;
typedef struct zkw_ram_witness zkw_ram_witness;
typedef struct zkw_trace zkw_trace;

enum {
    ZKW_OK = 0,
    ZKW_ERR_INVALID = -1,
    ZKW_ERR_NO_DEVICE = -2,
    ZKW_ERR_HIP = -3,
    ZKW_ERR_OOM = -4,
    ZKW_ERR_CHECK_FAILED = -5 /* a self-check of the reference failed (e.g. lhs != rhs grand product) */
};

enum { ZKW_PTR_HOST = 0, ZKW_PTR_DEVICE = 1 };

/* ---- context ------------------------------------------------------------------------------------ */
/* Optional, once per process, BEFORE the process's first HIP call: process-wide settings the block sequencer's scheduling
   relies on (today: GPU_MAX_HW_QUEUES=8 unless the host exported a value; csrc/zkw_block.hip explains the number). The
   library changes nothing in the process environment unless this is called. */
int zkw_process_init(void);
/* device_id >= 0. Returns NULL on failure (see zkw_last_error). */
zkw_ctx *zkw_create(int device_id);
/* deferred while witnesses / traces created from ctx are outstanding (see "lifetimes" above) */
void zkw_destroy(zkw_ctx *ctx);
const char *zkw_last_error(void);
/* hip_stream: a hipStream_t (NULL = the HIP null stream, which is what torch's default stream is), or
   ZKW_STREAM_OWN for the non-blocking stream the context created for itself (the default). */
#define ZKW_STREAM_OWN ((void *)(intptr_t)-1)
int zkw_set_stream(zkw_ctx *ctx, void *hip_stream);
/* optional: run the queue-chain kernels on this HIP stream (ordered against the context's stream by events); lets the
   host give the latency-bound chains their own CUs with a stream created by hipExtStreamCreateWithCUMask. NULL = off. */
int zkw_set_chain_stream(zkw_ctx *ctx, void *hip_stream);
int zkw_set_pointer_mode(zkw_ctx *ctx, int mode);
int zkw_synchronize(zkw_ctx *ctx);
/* tuning knob: lanes that cooperate on one Poseidon2 state in the queue-chain kernel: 16 (4 chains per wave,
   row DPP, lowest latency), 4 (16 chains per wave, quad DPP), 2 (32 chains per wave, a pair of lanes per state), 1 (64 chains per wave, the whole state in one lane:
   fewest instructions per permutation, for launches of tens of thousands of queues) or 0 = choose by the number of
   chains in the launch (default). Results are identical. */
int zkw_set_chain_form(zkw_ctx *ctx, int lanes_per_state);
/* tuning knob: how the netlist circuits (types 3, 5, 6, 10, 13) are filled. 0 (default): a wave owns a cycle and walks its
   netlist level by level (k_nl_fill). 1: a lane owns a cycle — 64 cycles run one resolved instruction stream side by side,
   cells go through a byte-sized scratch tile and a transposing pass (k_nl_walk + k_nl_expand). Results are identical; form 1
   is the slower one on every circuit measured so far and is kept as a tested alternative. */
int zkw_set_netlist_fill_form(zkw_ctx *ctx, int form);
/* Opt this context into the device's chain service: its queue-chain jobs are no longer launched on its own stream but
   handed to a per-device worker that merges the jobs of ALL opted-in contexts arriving within a short window (0.4 - 4 ms)
   into ONE launch on a high-priority stream, and the call waits for that launch. For hosts that keep many contexts busy
   at once (zkw_blocks_run): K concurrent builders cost one chain pass instead of K serialised ones. Results are identical. */
int zkw_set_chain_service(zkw_ctx *ctx, int on);
/* Stages. A host that runs the SAME graph of builders for many blocks at once (zkw_blocks_run) gives every branch of the graph a
   tag (> 0) on its context: the n-th chain submission of the contexts tagged t is stage (t, n), and the service batches a stage of
   all blocks into ONE launch — its chains are equally long, so the launch costs what one costs. zkw_chain_service_expect(+K) says
   how many blocks are in flight on the device: a stage with K submitters leaves at once, otherwise after 100 ms of silence (a block
   may skip a stage); (-K) when they are done. Without a tag or an expectation the service batches by arrival time as before.
   Measured at 96 production-capacity blocks in flight: DESIGN.md, review table item 4. */
int zkw_set_chain_tag(zkw_ctx *ctx, int tag);
int zkw_chain_service_expect(int device_id, int delta);
/* Buffers and streams from the library's caches (see "memory" above), for hosts that build graphs of contexts the way
   zkw_block_run does. pinned_host = 0: device memory on ctx's device, 1: pinned host memory. Contents are unspecified.
   zkw_stream_release synchronises the stream. */
int zkw_buffer_alloc(zkw_ctx *ctx, int pinned_host, size_t bytes, void **out);
void zkw_buffer_free(int pinned_host, void *p);
int zkw_stream_acquire(zkw_ctx *ctx, void **hip_stream);
void zkw_stream_release(zkw_ctx *ctx, void *hip_stream);
void zkw_trim_caches(void);
/* library/ABI version and the kernels' target ISA ("gfx950") */
const char *zkw_version(void);

/* ZkSyncBaseLayerCircuit::{numeric_circuit_type, geometry, size_hint} + the per-type capacity of GeometryConfig
   (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:268-394, 442-476; wrappers under base_layer/;
   circuit_sequencer_api/src/geometry_config.rs:5-20). circuit_type = BaseLayerCircuitType as u8 (1 = MainVM ...
   13 = L1MessagesHasher). trace_len_log2 = 20 for every type. Returns ZKW_ERR_INVALID for an unknown type. */
typedef struct zkw_circuit_geometry {
    uint32_t num_columns_under_copy_permutation;
    uint32_t num_witness_columns;
    uint32_t num_constant_columns;
    uint32_t max_allowed_constraint_degree;
    uint32_t lookup_width;          /* LookupParameters: width x num_repetitions, table id as constant */
    uint32_t lookup_repetitions;
    uint32_t capacity;              /* cycles / items per instance (geometry_config.rs) */
    uint32_t trace_len_log2;
    uint64_t size_hint_variables;   /* size_hint().1: 2^26, or 2^26 + 2^25 for RAM / sorters / demuxer */
} zkw_circuit_geometry;
int zkw_circuit_geometry_of(uint8_t circuit_type, zkw_circuit_geometry *out);

/* Where this library's own trace layout of a circuit type puts things — the counterpart, for the "zkw trace v2" layouts,
   of what the reference stores per circuit in FinalizationHintsForProver (setup/base_layer/finalization_hint_N.json:
   `public_inputs` locations, `nop_gates_to_add`, `final_trace_len`; src/prover_utils.rs:48-197 produces it,
   base_layer/mod.rs:302-312 consumes it). capacity = 0: geometry_config.rs default. synthesizable = 0 for the circuit
   types without a layout yet (the other fields are then 0 except capacity / trace_len). No GPU needed.
   These layouts are NOT the reference's gate placement: rows_used differs from its hints and the traces do not pair with
   its vk_N.json. */
typedef struct zkw_circuit_layout {
    uint32_t synthesizable;
    uint32_t fits;              /* rows_used <= trace_len */
    uint32_t capacity;
    uint32_t num_columns;       /* copy-permutation + lookup + multiplicity columns of a trace slot */
    uint32_t rows_per_cycle;    /* row types repeated once per cycle, region-major */
    uint32_t total_table_rows;  /* rows of the stacked lookup tables = the reference's `total_tables_len` (vk_N.json) */
    uint64_t region_stride;     /* rows between two regions (capacity rounded up to 64); 0: cycle-major (the netlist circuits 3, 5, 6, 13) */
    uint64_t rows_used;         /* cycle regions + boundary rows */
    uint64_t nop_rows;          /* trace_len - rows_used: zero padding, the reference's nop_gates_to_add */
    uint64_t trace_len;         /* 2^20 */
    uint32_t public_input_column[4];
    uint64_t public_input_row[4];
    /* the queue section of a netlist circuit (types 6 and 3; include/zkw_netlist_queue.h): request-queue pops and memory-queue pushes
       as Poseidon2 rows below the netlist. Row queue_first_row holds the queue states before / after the instance; row r of the
       operations of cycle c is queue_first_row + 1 + r * cycles + c. 0 rows per cycle: the type has no section. */
    uint64_t queue_first_row;
    uint32_t queue_rows_per_cycle;
    /* the EC section of the ECRecover circuit (type 7; include/zkw_ecrecover.h): row r of cycle c is ec_first_row + c * ec_rows_per_cycle + r */
    uint32_t ec_rows_per_cycle;
    uint64_t ec_first_row;
    /* the closed-form section of a netlist circuit (types 3, 5, 6, 7, 10, 13; include/zkw_netlist_closed_form.h): flags, the words of the
       closed-form input, their ties to the trace's registers, then the flattened Poseidon2 rows of the commitments, the compact form and
       the public input — closed_form_rows rows from closed_form_first_row, the last of rows_used */
    uint64_t closed_form_first_row;
    uint32_t closed_form_rows;
    uint32_t closed_form_header_rows; /* rows of flags / words / ties; the rest are Poseidon2 rows */
} zkw_circuit_layout;
int zkw_circuit_layout_of(uint8_t circuit_type, uint32_t capacity, zkw_circuit_layout *out);
/* Setup side, selectors: out[r] (host, n_rows bytes) says which gate set applies to row r of a trace of this library's
   layout — what the reference's setup keeps in its constant columns (gate selectors / the lookup table id of a row).
   Queue circuits (2, 4, 8, 9, 11, 12): the row type of the type's spec header (0 .. NUM_ROW_TYPES - 1: the per-cycle row types
   in region order, then the boundary rows). Netlist circuits (3, 5, 6, 13): the lookup table id of the row (0: none), |
   ZKW_ROW_HAS_GATES where ADD gates sit in the general-purpose columns, ZKW_ROW_HEADER for a cycle's first row, boundary rows
   ZKW_ROW_BOUNDARY + k. ZKW_ROW_PADDING: the row holds nothing (all cells zero). capacity 0 = the type's default.
   No GPU needed. Copy-permutation (sigma) columns are not produced yet. */
/* bytes one synthesis call writes into one slot (the cells its fill kernels store x 8): *warm when the slot already holds this layout
   (same circuit, capacity and row count: the slot keeps every cell that is zero in all traces of the layout — padding rows, the columns
   a row type does not use, region gaps, multiplicity rows >= 256; see zkw_trace_device_ptr for what resets the tag), *cold for any other
   slot (every cell of the slot's columns). What measured circuits/s are multiplied by to get bytes/s. */
int zkw_circuit_fill_bytes(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint64_t *warm, uint64_t *cold);
#define ZKW_ROW_HAS_GATES 0x40
#define ZKW_ROW_HEADER 0x80
#define ZKW_ROW_BOUNDARY 0xC0
#define ZKW_ROW_QUEUE_BOUNDARY 0xE0 /* queue section: the queue states before / after the instance */
#define ZKW_ROW_QUEUE_ENCODING 0xE1 /* queue section: an item's fields, its encoding (linear gates), old / new state (selection gates) */
#define ZKW_ROW_QUEUE_POSEIDON2 0xE2 /* queue section: a (folded) flattened Poseidon2 gate */
#define ZKW_ROW_CLOSED_FORM_WORDS 0xE4 /* closed-form section: flags, words, ties (selection / recomposition gates) */
#define ZKW_ROW_CLOSED_FORM_POSEIDON2 0xE5 /* closed-form section: a (folded) flattened Poseidon2 gate of a commitment sponge */
#define ZKW_ROW_EC_GATES 0xF0      /* EC section of the ECRecover circuit: LIN / SEL / FMA gates in the general-purpose columns, no lookups */
#define ZKW_ROW_EC_XOR8 0xF1       /* ... its 16 lookup slots are Xor8 (byte range checks) */
#define ZKW_ROW_EC_FIXED_BASE 0xF2 /* ... its lookup slots are a FixedBaseMul table (which one: the row's segment instance) */
#define ZKW_ROW_EC_MUL 0x04        /* | on an EC row: the general-purpose columns are ONE non-native multiplication gate */
#define ZKW_ROW_PADDING 0xFF
int zkw_setup_row_selectors(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint8_t *out);
/* Setup side, copy permutation of the ten synthesized types: sigma[c * n_rows + r] (host, *n_columns x n_rows words;
   *n_columns = every column but the multiplicity column(s)) = the cell c' * n_rows + r' that follows cell (c, r) in its copy
   cycle, itself when the cell is under no copy constraint. Built by a host union-find — from the spec's link table for the
   queue circuits, from the netlist's operand references for the netlist circuits (seconds and ~1.1 GB at production size;
   sigma = NULL only reports *n_columns). zkw_check_copy_permutation: trace[cell] == trace[sigma[cell]] on the GPU for every
   cell of the first n_columns columns of a slot (sigma: host or device pointer per the context's pointer mode); violations
   are reported as kind 4 with (column, row). */
int zkw_setup_copy_permutation(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint64_t *sigma, uint32_t *n_columns);
int zkw_check_copy_permutation(zkw_ctx *ctx, const zkw_trace *t, size_t slot, const uint64_t *sigma, uint32_t n_columns,
                               uint64_t *n_violations, uint64_t *first_bad);

/* ---- Setup side as field elements: NTT, low-degree extension, Merkle tree with a cap ------------------------------------
   What the reference does between a synthesized circuit and its verification key (src/prover_utils.rs:48-197
   create_base_layer_setup_data: SetupBaseStorage -> SetupStorage in monomial form and as an LDE of factor fri_lde_factor ->
   MerkleTreeWithCap over the LDE (cap 16) -> VerificationKey.setup_merkle_tree_cap; bodies in boojum, absent). Pinned by the
   reference's data: leaf = overwrite-mode Poseidon2 sponge over a position's values (eight at a time, zero padded), node =
   permutation of (left || right || 0000), ONE binary tree over all LDE positions cut at the cap: every Merkle path of the
   committed proofs and setup_merkle_tree_cap of the vks reproduce (tests/golden/reference_merkle_paths_kat.json). The domain:
   multiplicative generator 7, w_(2^32) = 7^((p-1)/2^32) = 0x185629dcda58878c (boojum's constant). NOT pinned (bodies absent):
   which point a leaf index stands for — here leaf c * n + i is the point 7 * w_(lde n)^c * w_n^i, natural order — and the coset
   representatives of the sigma polynomials (here k_j = 7^j). Pointers follow the context's pointer mode; arrays are column-major
   like the traces ([column][position]). log_n <= 20.
   zkw_ntt: out[c][k] = sum_j in[c][j] w_n^(jk) (inverse: w^-1 and 1/n), natural order both sides, out may be in.
   zkw_lde: values of n_cols polynomials on the domain -> out[coset][column][i], lde_factor cosets (a power of two <= 8).
   zkw_merkle_tree_with_cap: leaf_cols[set][column][i] (leaf index = set * n + i) -> cap[cap_size][4]; tree (optional, NULL to skip):
     every level, leaves first, zkw_merkle_tree_words(n_sets * n, cap_size) words — the cap is its last level.
   zkw_setup_num_columns / zkw_setup_columns: the setup columns of one of this library's layouts as field elements
     ([n_columns][2^log_n]): the sigma columns of zkw_setup_copy_permutation (cell (c', r') as k_c' * w^r'), the selector
     column of zkw_setup_row_selectors, then the lookup-table columns of zkw_setup_lookup_tables. zkw_setup_commit: those columns -> monomial form -> LDE -> tree -> cap[cap_size][4]
     (host pointer in host mode), all on the device. */
int zkw_ntt(zkw_ctx *ctx, const uint64_t *in, uint64_t *out, uint32_t log_n, size_t n_cols, int inverse);
int zkw_lde(zkw_ctx *ctx, const uint64_t *values, uint32_t log_n, size_t n_cols, uint32_t lde_factor, uint64_t *out);
size_t zkw_merkle_tree_words(size_t n_leaves, uint32_t cap_size);
int zkw_merkle_tree_with_cap(zkw_ctx *ctx, const uint64_t *leaf_cols, size_t n_sets, size_t n_cols, size_t n, uint32_t cap_size,
                             uint64_t *cap, uint64_t *tree);
/* the lookup tables of a layout as columns (host; what the wrappers' add_tables put into the setup): cols[c][t] for row t of the stacked table
   (= row t of the multiplicity column): the table's cells (inputs, then outputs: width columns), then a table-id column (1-based index in the
   circuit's table list, 0 = not a table row). Queue circuits: the 8-bit range table (width 1). cols = NULL only reports *n_columns. */
int zkw_setup_lookup_tables(uint8_t circuit_type, size_t n_rows, uint64_t *cols, uint32_t *n_columns);
int zkw_setup_num_columns(uint8_t circuit_type, uint32_t *n_columns);
int zkw_setup_columns(zkw_ctx *ctx, uint8_t circuit_type, uint32_t capacity, uint32_t log_n, uint64_t *columns);
int zkw_setup_commit(zkw_ctx *ctx, uint8_t circuit_type, uint32_t capacity, uint32_t log_n, uint32_t lde_factor, uint32_t cap_size,
                     uint64_t *cap);

/* ---- per-kernel timing -------------------------------------------------------------------------- */
/* When enabled, every kernel launch (and library sort) of this context is bracketed by HIP events
   recorded on the context's stream; totals are keyed by kernel name ("k_chain_full", "k_gp_local",
   "radix_sort", ...). zkw_profile_get synchronises the stream. Used by bench.py for the roofline. */
int zkw_profile_enable(zkw_ctx *ctx, int on);
int zkw_profile_reset(zkw_ctx *ctx);
int zkw_profile_get(zkw_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches);
/* comma-separated list of the kernel names seen so far */
int zkw_profile_names(zkw_ctx *ctx, char *buf, size_t buf_bytes);

/* ---- L1 primitives ------------------------------------------------------------------------------ */
/* MemoryQuery::encoding_witness, circuit_encodings/src/memory_query.rs:24-118. enc: [n][8]. */
int zkw_encode_memory_queries(zkw_ctx *ctx, const zkw_mem_query *q, size_t n, uint64_t *enc);

/* FullWidthQueueSimulator::push_and_output_intermediate_data applied to n items in order,
   circuit_encodings/src/lib.rs:391-429 (memory queue: src/witness/oracle.rs:894-903).
   tails[i] = sponge state after item i; tail_in = state before item 0 (NULL = empty queue). */
int zkw_queue_push_chain_full(zkw_ctx *ctx, const uint64_t *enc /* [n][8] */, size_t n,
                              const uint64_t tail_in[12], uint64_t *tails /* [n][12] */);
/* n_queues independent queues in one launch: queue k owns items [offsets[k], offsets[k+1]).
   offsets: host, n_queues+1 entries. tails_in: [n_queues][12] or NULL. */
int zkw_queue_push_chain_full_batch(zkw_ctx *ctx, const uint64_t *enc, const uint64_t *offsets,
                                    size_t n_queues, const uint64_t *tails_in, uint64_t *tails);

/* LogQuery::encoding_witness, circuit_encodings/src/log_query.rs:102-396; with ext_ts != NULL the
   LogQueryWithExtendedEnumeration variant (log_query.rs:400-427). enc: [n][20]. */
int zkw_encode_log_queries(zkw_ctx *ctx, const zkw_log_query *q, size_t n, const uint32_t *ext_ts, uint64_t *enc);
/* DecommittmentQuery::encoding_witness, circuit_encodings/src/decommittment_request.rs:9-74. enc: [n][8]
   (feeds zkw_queue_push_chain_full: the decommit queue is a full-width queue). */
int zkw_encode_decommit_queries(zkw_ctx *ctx, const zkw_decommit_query *q, size_t n, uint64_t *enc);

/* QueueSimulator::push_and_output_intermediate_data applied to the items of n_queues independent
   4-wide queues, circuit_encodings/src/lib.rs:179-221 (LogQueueSimulator: storage, events, L1 messages,
   precompile requests). enc: [total][20]; offsets: host, n_queues+1; tails_in: [n_queues][4] or NULL;
   old_tails: [total][4] or NULL (the tail BEFORE each push, what the reference keeps as queue witness,
   lib.rs:204); new_tails: [total][4]. */
int zkw_queue_push_chain_log_batch(zkw_ctx *ctx, const uint64_t *enc, const uint64_t *offsets, size_t n_queues,
                                   const uint64_t *tails_in, uint64_t *old_tails, uint64_t *new_tails);
int zkw_queue_push_chain_log(zkw_ctx *ctx, const uint64_t *enc, size_t n, const uint64_t tail_in[4],
                             uint64_t *old_tails, uint64_t *new_tails);

/* produce_fs_challenges, src/witness/utils.rs:498-550. state_w = 12 (RAM, decommit sorter) or 4
   (storage / events sorters); out: [2 repetitions][n_chal], out[r][0] = 1. */
int zkw_fs_challenges(zkw_ctx *ctx, const uint64_t *tail_u, uint32_t len_u, const uint64_t *tail_s,
                      uint32_t len_s, int state_w, int n_chal, uint64_t *out);

/* compute_grand_product_chains, src/witness/utils.rs:554-697, for n_reps (1 or 2) challenge sets in
   one pass over the rows. lhs, rhs: [n][width] (width 8 or 20); challenges: [n_reps][width+1];
   lhs_z, rhs_z: [n_reps][n]. Returns ZKW_ERR_CHECK_FAILED when a final lhs product differs from the
   rhs one (utils.rs:685-696) — only checked in host pointer mode (device mode never syncs). */
int zkw_grand_product_chains(zkw_ctx *ctx, const uint64_t *lhs, const uint64_t *rhs, size_t n, int width,
                             const uint64_t *challenges, int n_reps, uint64_t *lhs_z, uint64_t *rhs_z);

/* ---- RAM permutation witness builder ------------------------------------------------------------ */
/* compute_ram_circuit_snapshots, src/witness/individual_circuits/ram_permutation.rs:26-470, for one
   block's memory queue (q in queue order). *out must be NULL or a witness previously returned for the
   same shape (its buffers are reused). In device-pointer mode the witness keeps a reference to q (the kernels
   re-encode the queries instead of storing their encodings): q must stay valid and unchanged until the witness
   has been synthesized / read or is freed. In host-pointer mode the witness owns a device copy. */
int zkw_ram_build_instances(zkw_ctx *ctx, const zkw_mem_query *q, size_t n, uint32_t capacity,
                            uint32_t num_non_deterministic_heap_queries, zkw_ram_witness **out);
/* The same for n_blocks independent memory queues (one per block being proven) in one pass:
   block b owns q[block_offsets[b] .. block_offsets[b+1]). block_offsets / n_nondet: host arrays. */
int zkw_ram_build_instances_batch(zkw_ctx *ctx, const zkw_mem_query *q, const uint64_t *block_offsets,
                                  size_t n_blocks, uint32_t capacity, const uint32_t *n_nondet,
                                  zkw_ram_witness **out);

enum {
    ZKW_RAM_SORTED_QUERIES = 0, /* zkw_mem_query[total]: gathered on first access (the builder keeps the permutation) */
    ZKW_RAM_UNSORTED_ENC = 1,   /* uint64_t[total][8]: materialised on first access (the builder encodes the 48-byte */
    ZKW_RAM_SORTED_ENC = 2,     /* queries on the fly in every kernel instead of keeping 2 x 64 bytes per query)     */
    ZKW_RAM_UNSORTED_TAILS = 3, /* uint64_t[total][12]: expanded on first access (the builder keeps only the   */
    ZKW_RAM_SORTED_TAILS = 4,   /* capacity words + the tails at instance ends: 64 instead of 192 bytes per query) */
    ZKW_RAM_CHALLENGES = 5,     /* uint64_t[n_blocks][2][9]       */
    ZKW_RAM_LHS_Z = 6,          /* per block b: uint64_t[2][n_b] at element offset 2*block_offsets[b]; computed on first
                                   access (builder and synthesis recompute the chains in a window) */
    ZKW_RAM_RHS_Z = 7,          /* idem                           */
    ZKW_RAM_INSTANCES = 8,      /* zkw_ram_instance[n_instances], blocks in order */
    /* a20, CircuitMaker::process src/witness/postprocessing/mod.rs:353-405 (every instance of a block shares
       the first one's observable input): ClosedFormInputCompactForm = [start, completion, C(observable_input),
       C(observable_output), C(hidden_fsm_input), C(hidden_fsm_output)], and the public input C(compact form)
       that simulate_public_input_value_from_witness (src/witness/utils.rs:269-306) returns. */
    ZKW_RAM_COMPACT_FORMS = 9,  /* uint64_t[n_instances][18]      */
    ZKW_RAM_PUBLIC_INPUTS = 10  /* uint64_t[n_instances][4]       */
};
size_t zkw_ram_witness_num_instances(const zkw_ram_witness *w);
size_t zkw_ram_witness_num_items(const zkw_ram_witness *w);
/* size in bytes of one of the arrays above */
size_t zkw_ram_witness_bytes(const zkw_ram_witness *w, int what);
/* device (HBM) address of the array; valid until the witness is freed or rebuilt */
const void *zkw_ram_witness_device_ptr(const zkw_ram_witness *w, int what);
/* copy an array out (to host memory, or to a device buffer in ZKW_PTR_DEVICE mode) */
int zkw_ram_witness_get(const zkw_ram_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_ram_witness_free(zkw_ram_witness *w);

/* ---- CodeDecommittmentsSorter witness builder --------------------------------------------------- */
typedef struct zkw_decommit_witness zkw_decommit_witness;
/* compute_decommitts_sorter_circuit_snapshots, src/witness/individual_circuits/sort_decommit_requests.rs:20-420.
   q: the block's decommit requests in queue order (n > 0); dedup_in (host, may be NULL = empty) is the
   state of the deduplicated queue the results are appended to (the reference passes the simulator in).
   Returns ZKW_ERR_CHECK_FAILED when the reference's ordering self-check (:99-114) or the grand-product
   check fails. */
int zkw_decommit_sorter_build(zkw_ctx *ctx, const zkw_decommit_query *q, size_t n, uint32_t capacity,
                              const zkw_queue_state12 *dedup_in, zkw_decommit_witness **out);
/* The same builder in two phases, so that a block sequencer can overlap the hash chains of independent builders:
   _prepare does everything that depends on the requests' CONTENTS only (encodings, the stable sort, the deduplicated
   queue: ZKW_DEC_SORTED_QUERIES / _ENC and ZKW_DEC_DEDUP_QUERIES are valid when it returns and it has synchronised),
   _finish hashes the three queues in one launch and derives challenges, grand products, instance records and public
   inputs. No other decommit-sorter call may be made on ctx between the two. zkw_decommit_sorter_build = both. */
int zkw_decommit_sorter_prepare(zkw_ctx *ctx, const zkw_decommit_query *q, size_t n, uint32_t capacity,
                                const zkw_queue_state12 *dedup_in, zkw_decommit_witness **out);
int zkw_decommit_sorter_finish(zkw_ctx *ctx, zkw_decommit_witness *w);
enum {
    ZKW_DEC_SORTED_QUERIES = 0, /* zkw_decommit_query[n]  */
    ZKW_DEC_UNSORTED_ENC = 1,   /* uint64_t[n][8]         */
    ZKW_DEC_SORTED_ENC = 2,
    ZKW_DEC_UNSORTED_TAILS = 3, /* uint64_t[n][12]        */
    ZKW_DEC_SORTED_TAILS = 4,
    ZKW_DEC_DEDUP_QUERIES = 5,  /* zkw_decommit_query[n_dedup] : deduplicated_decommit_requests */
    ZKW_DEC_DEDUP_TAILS = 6,    /* uint64_t[n_dedup][12]   : deduplicated_decommittment_queue_states */
    ZKW_DEC_CHALLENGES = 7,     /* uint64_t[2][9]          */
    ZKW_DEC_LHS_Z = 8,          /* uint64_t[2][n]          */
    ZKW_DEC_RHS_Z = 9,
    ZKW_DEC_INSTANCES = 10,     /* zkw_decommit_sorter_instance[ceil(n/capacity)] */
    ZKW_DEC_COMPACT_FORMS = 11, /* uint64_t[n_instances][18]: see ZKW_RAM_COMPACT_FORMS */
    ZKW_DEC_PUBLIC_INPUTS = 12  /* uint64_t[n_instances][4] */
};
size_t zkw_decommit_witness_num_instances(const zkw_decommit_witness *w);
size_t zkw_decommit_witness_num_dedup(const zkw_decommit_witness *w);
size_t zkw_decommit_witness_bytes(const zkw_decommit_witness *w, int what);
const void *zkw_decommit_witness_device_ptr(const zkw_decommit_witness *w, int what);
int zkw_decommit_witness_get(const zkw_decommit_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_decommit_witness_free(zkw_decommit_witness *w);

/* ---- Events / L1-messages sorter witness builder ------------------------------------------------ */
typedef struct zkw_events_witness zkw_events_witness;
/* compute_events_dedup_and_sort (+ sort_and_dedup_events_log), src/witness/individual_circuits/
   events_sort_dedup.rs:16-580 — used twice by the reference (events and L2->L1 messages, oracle.rs:1069-1088).
   q: the demuxed log queue in queue order (n == 0 yields the reference's single dummy instance);
   result_in (host, NULL = empty): state of the result queue the net events are appended to.
   PRECONDITION (a deliberate difference from the reference): the queue must be WELL-FORMED — per timestamp at most
   one forward record, optionally followed, somewhere later in the queue, by its own rollback twin (same timestamp and payload,
   rollback = 1). The reference's comparator (events_sort_dedup.rs:81-90) is not a strict weak order on other queues, so its result
   there depends on the sorting algorithm; this library sorts stably by (timestamp, rollback) — which is what the reference's
   comparator does on well-formed queues — and REJECTS everything else with ZKW_ERR_CHECK_FAILED (the number of violations in
   zkw_last_error()) instead of reproducing an order nobody specified. ZKW_ERR_CHECK_FAILED also when one of the reference's own
   asserts on the queue's shape fails (events_sort_dedup.rs:344-356, 512-533). */
int zkw_events_sorter_build(zkw_ctx *ctx, const zkw_log_query *q, size_t n, uint32_t capacity,
                            const zkw_queue_state4 *result_in, zkw_events_witness **out);
enum {
    ZKW_EVT_SORTED_QUERIES = 0,     /* zkw_log_query[n]  */
    ZKW_EVT_UNSORTED_ENC = 1,       /* uint64_t[n][20]   */
    ZKW_EVT_SORTED_ENC = 2,
    ZKW_EVT_UNSORTED_OLD_TAILS = 3, /* uint64_t[n][4]: tail before each push = queue witness (lib.rs:204) */
    ZKW_EVT_UNSORTED_NEW_TAILS = 4,
    ZKW_EVT_SORTED_OLD_TAILS = 5,
    ZKW_EVT_SORTED_NEW_TAILS = 6,
    ZKW_EVT_RESULT_QUERIES = 7,     /* zkw_log_query[n_result]: sort_and_dedup_events_log output */
    ZKW_EVT_RESULT_NEW_TAILS = 8,   /* uint64_t[n_result][4] */
    ZKW_EVT_CHALLENGES = 9,         /* uint64_t[2][21]   */
    ZKW_EVT_LHS_Z = 10,             /* uint64_t[2][n]    */
    ZKW_EVT_RHS_Z = 11,
    ZKW_EVT_INSTANCES = 12,         /* zkw_events_sorter_instance[max(1, ceil(n/capacity))] */
    ZKW_EVT_COMPACT_FORMS = 13,     /* uint64_t[n_instances][18]: see ZKW_RAM_COMPACT_FORMS */
    ZKW_EVT_PUBLIC_INPUTS = 14      /* uint64_t[n_instances][4] */
};
size_t zkw_events_witness_num_instances(const zkw_events_witness *w);
size_t zkw_events_witness_num_results(const zkw_events_witness *w);
size_t zkw_events_witness_bytes(const zkw_events_witness *w, int what);
const void *zkw_events_witness_device_ptr(const zkw_events_witness *w, int what);
int zkw_events_witness_get(const zkw_events_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_events_witness_free(zkw_events_witness *w);

/* ---- LogDemuxer witness builder ------------------------------------------------------------------- */
typedef struct zkw_demux_witness zkw_demux_witness;
/* compute_logs_demux, src/witness/individual_circuits/log_demux.rs:20-388. q: the original log queue in order
   (n == 0 yields the reference's single placeholder instance); params: routing constants (NULL = defaults).
   The six demuxed queues (queries, encodings, old/new tails = LogQueue.simulator.witness / .states) are what
   the storage / events / L1-message sorters and the precompile circuits consume next.
   ZKW_ERR_CHECK_FAILED on an input the reference treats as unreachable!(). */
int zkw_log_demux_build(zkw_ctx *ctx, const zkw_log_query *q, size_t n, uint32_t capacity,
                        const zkw_demux_params *params, zkw_demux_witness **out);
enum {
    ZKW_DMX_IN_ENC = 0,        /* uint64_t[n][20]  */
    ZKW_DMX_IN_OLD_TAILS = 1,  /* uint64_t[n][4]   */
    ZKW_DMX_IN_NEW_TAILS = 2,
    ZKW_DMX_OUT_QUERIES = 3,   /* zkw_log_query[routed]: the six queues back to back, route order */
    ZKW_DMX_OUT_ENC = 4,       /* uint64_t[routed][20] */
    ZKW_DMX_OUT_OLD_TAILS = 5, /* uint64_t[routed][4]  */
    ZKW_DMX_OUT_NEW_TAILS = 6,
    ZKW_DMX_OUT_OFFSETS = 7,   /* uint64_t[7]: queue k = [offsets[k], offsets[k+1]) */
    ZKW_DMX_INSTANCES = 8,     /* zkw_log_demux_instance[max(1, ceil(n/capacity))] */
    ZKW_DMX_COMPACT_FORMS = 9, /* uint64_t[n_instances][18]: see ZKW_RAM_COMPACT_FORMS */
    ZKW_DMX_PUBLIC_INPUTS = 10 /* uint64_t[n_instances][4] */
};
size_t zkw_demux_witness_num_instances(const zkw_demux_witness *w);
size_t zkw_demux_witness_bytes(const zkw_demux_witness *w, int what);
const void *zkw_demux_witness_device_ptr(const zkw_demux_witness *w, int what);
int zkw_demux_witness_get(const zkw_demux_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_demux_witness_free(zkw_demux_witness *w);

/* ---- StorageSorter witness builder ---------------------------------------------------------------- */
typedef struct zkw_storage_witness zkw_storage_witness;
/* sort_storage_access_queries (circuit_sequencer_api/src/sort_storage_access.rs:19-260) +
   compute_storage_dedup_and_sort (src/witness/individual_circuits/storage_sort_dedup.rs:12-703).
   q: the demuxed rollup-storage queue in order (n == 0 yields the reference's dummy instance with its
   cycle_idx = 4 convention). ZKW_ERR_CHECK_FAILED when the log is not a consistent storage history
   (rollback without a pending write, first access of a cell being a rollback, ...). */
int zkw_storage_sorter_build(zkw_ctx *ctx, const zkw_log_query *q, size_t n, uint32_t capacity,
                             zkw_storage_witness **out);
enum {
    ZKW_STO_SORTED_QUERIES = 0,     /* zkw_log_query[n] */
    ZKW_STO_SORTED_EXT_TS = 1,      /* uint32_t[n]: extended timestamp = position in the unsorted queue */
    ZKW_STO_UNSORTED_ENC = 2,       /* uint64_t[n][20]: plain encodings hashed into the unsorted queue */
    ZKW_STO_LHS_ENC = 3,            /* uint64_t[n][20]: with extended timestamp (storage_sort_dedup.rs:128-143) */
    ZKW_STO_SORTED_ENC = 4,         /* uint64_t[n][20]: with extended timestamp */
    ZKW_STO_UNSORTED_OLD_TAILS = 5, /* uint64_t[n][4] */
    ZKW_STO_UNSORTED_NEW_TAILS = 6,
    ZKW_STO_SORTED_OLD_TAILS = 7,
    ZKW_STO_SORTED_NEW_TAILS = 8,
    ZKW_STO_RESULT_QUERIES = 9,     /* zkw_log_query[n_result]: deduplicated_rollup_storage_queries */
    ZKW_STO_RESULT_NEW_TAILS = 10,  /* uint64_t[n_result][4] */
    ZKW_STO_CHALLENGES = 11,        /* uint64_t[2][21] */
    ZKW_STO_LHS_Z = 12,             /* uint64_t[2][n] */
    ZKW_STO_RHS_Z = 13,
    ZKW_STO_INSTANCES = 14,         /* zkw_storage_sorter_instance[max(1, ceil(n/capacity))] */
    ZKW_STO_COMPACT_FORMS = 15,     /* uint64_t[n_instances][18]: see ZKW_RAM_COMPACT_FORMS */
    ZKW_STO_PUBLIC_INPUTS = 16      /* uint64_t[n_instances][4] */
};
size_t zkw_storage_witness_num_instances(const zkw_storage_witness *w);
size_t zkw_storage_witness_num_results(const zkw_storage_witness *w);
size_t zkw_storage_witness_bytes(const zkw_storage_witness *w, int what);
const void *zkw_storage_witness_device_ptr(const zkw_storage_witness *w, int what);
int zkw_storage_witness_get(const zkw_storage_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_storage_witness_free(zkw_storage_witness *w);

/* ---- CodeDecommitter witness builder --------------------------------------------------------------- */
typedef struct zkw_decommitter_witness zkw_decommitter_witness;
/* compute_decommitter_circuit_snapshots, src/witness/individual_circuits/decommit_code.rs:20-439.
   requests / dedup_tails: the deduplicated decommit requests and the deduplicated queue's states (outputs
   ZKW_DEC_DEDUP_QUERIES / ZKW_DEC_DEDUP_TAILS of zkw_decommit_sorter_build started from an empty queue);
   words: the bytecodes back to back, 8 little-endian u32 limbs per 32-byte word, request k owning words
   [word_offsets[k], word_offsets[k+1]) (word_offsets: host, n_requests + 1); capacity = SHA-256 rounds per
   instance; mem_in (host): state of the global memory queue before the call. The code words become memory
   writes appended to that queue (ZKW_DCM_MEM_*), which the RAM permutation consumes later.
   ZKW_ERR_CHECK_FAILED when a bytecode does not hash to its request (decommit_code.rs:323-337). */
int zkw_decommitter_build(zkw_ctx *ctx, const zkw_decommit_query *requests, const uint64_t *dedup_tails,
                          size_t n_requests, const uint32_t *words, const uint64_t *word_offsets, uint32_t capacity,
                          const zkw_queue_state12 *mem_in, zkw_decommitter_witness **out);
/* The same with the states of the memory queue over the code words given instead of hashed here:
   given_mem_tails[total_words][12] (NULL = hash them) — a slice of a memory queue the caller has already hashed as a
   whole (zkw_block_run hashes one block's memory queue once, inside the RAM-permutation builder). */
int zkw_decommitter_build_with_tails(zkw_ctx *ctx, const zkw_decommit_query *requests, const uint64_t *dedup_tails,
                                     size_t n_requests, const uint32_t *words, const uint64_t *word_offsets,
                                     uint32_t capacity, const zkw_queue_state12 *mem_in, const uint64_t *given_mem_tails,
                                     zkw_decommitter_witness **out);
/* Only the memory writes the code words become (decommit_code.rs:47-78), in the order of `requests`:
   out[total_words]. Lets a sequencer assemble the block's whole memory queue before anything is hashed. */
int zkw_decommitter_memory_queries(zkw_ctx *ctx, const zkw_decommit_query *requests, size_t n_requests,
                                   const uint32_t *words, const uint64_t *word_offsets, zkw_mem_query *out);
enum {
    ZKW_DCM_MEM_QUERIES = 0,  /* zkw_mem_query[total_words]     */
    ZKW_DCM_MEM_ENC = 1,      /* uint64_t[total_words][8]       */
    ZKW_DCM_MEM_TAILS = 2,    /* uint64_t[total_words][12]      */
    ZKW_DCM_ROUND_STATES = 3, /* uint32_t[total_rounds][8]: SHA-256 state after every round */
    ZKW_DCM_INSTANCES = 4,    /* zkw_decommitter_instance[ceil(total_rounds/capacity)] */
    ZKW_DCM_SHA256_ROUNDS = 5 /* zkw_sha256_round_record[total_rounds]: the cycles of the type-3 circuit */
};
size_t zkw_decommitter_witness_num_instances(const zkw_decommitter_witness *w);
size_t zkw_decommitter_witness_bytes(const zkw_decommitter_witness *w, int what);
const void *zkw_decommitter_witness_device_ptr(const zkw_decommitter_witness *w, int what);
int zkw_decommitter_witness_get(const zkw_decommitter_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_decommitter_witness_free(zkw_decommitter_witness *w);

/* ---- CodeDecommittmentsSorter synthesis (a21, circuit type 2) ------------------------------------------ */
/* Counterpart of ZkSyncBaseLayerCircuit::CodeDecommittmentsSorter(..).synthesis (base_layer/mod.rs:286-323, wrapper
   base_layer/sort_code_decommits.rs:28-39): fills the traces of instances [first_instance, first_instance + n) of a
   zkw_decommit_witness into consecutive slots of `t` (geometry 130 + 18 + 1 = 149 columns, layout
   include/zkw_decommit_sorter_circuit_spec.h, "zkw trace v2"; production capacity 117 500 needs n_rows = 2^20).
   Requires DS_MIN_ROWS(capacity) <= n_rows. */
int zkw_decommit_sorter_synthesize(zkw_ctx *ctx, const zkw_decommit_witness *w, size_t first_instance, size_t n_instances,
                                   zkw_trace *t, size_t first_slot);
int zkw_decommit_sorter_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                        uint64_t *n_violations, uint64_t *first_bad);

/* ---- EventsSorter / L1MessagesSorter synthesis (a21, circuit types 11 and 12) --------------------------- */
/* Counterpart of ZkSyncBaseLayerCircuit::{EventsSorter, L1MessagesSorter}(..).synthesis (base_layer/mod.rs:286-323,
   wrapper base_layer/events_sort_dedup.rs:28-39) on a zkw_events_witness: geometry 130 + 8 + 1 = 139 columns (the
   first 139 of a trace slot), layout include/zkw_events_sorter_circuit_spec.h ("zkw trace v2", 22 rows per cycle;
   production capacity 31 287 needs n_rows = 2^20). Requires ES_MIN_ROWS(capacity) <= n_rows. */
int zkw_events_sorter_synthesize(zkw_ctx *ctx, const zkw_events_witness *w, size_t first_instance, size_t n_instances,
                                 zkw_trace *t, size_t first_slot);
int zkw_events_sorter_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                      uint64_t *n_violations, uint64_t *first_bad);

/* ---- LogDemuxer synthesis (a21, circuit type 4) ----------------------------------------------------------- */
/* Counterpart of ZkSyncBaseLayerCircuit::LogDemuxer(..).synthesis (base_layer/mod.rs:286-323, wrapper
   base_layer/log_demux.rs:27-38) on a zkw_demux_witness: geometry 136 + 14 + 1 = 151 columns (create the trace with
   zkw_trace_create_with_columns(.., 151, ..)), layout include/zkw_log_demux_circuit_spec.h ("zkw trace v2", 12 rows
   per cycle; production capacity 58 750 needs n_rows = 2^20). Requires LD_MIN_ROWS(capacity) <= n_rows and a witness
   built with ZKW_DEMUX_PARAMS_DEFAULT (the routing constants are constants of the circuit). */
int zkw_log_demux_synthesize(zkw_ctx *ctx, const zkw_demux_witness *w, size_t first_instance, size_t n_instances,
                             zkw_trace *t, size_t first_slot);
int zkw_log_demux_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                  uint64_t *n_violations, uint64_t *first_bad);

/* ---- StorageSorter synthesis (a21, circuit type 9) -------------------------------------------------------- */
/* Counterpart of ZkSyncBaseLayerCircuit::StorageSorter(..).synthesis (base_layer/mod.rs:286-323, wrapper
   base_layer/storage_sort_dedup.rs:29-40) on a zkw_storage_witness: geometry 132 + 16 + 1 = 149 columns, layout
   include/zkw_storage_sorter_circuit_spec.h ("zkw trace v2", 22 rows per cycle; production capacity 46 921 needs
   n_rows = 2^20). Requires SS_MIN_ROWS(capacity) <= n_rows. */
int zkw_storage_sorter_synthesize(zkw_ctx *ctx, const zkw_storage_witness *w, size_t first_instance, size_t n_instances,
                                  zkw_trace *t, size_t first_slot);
int zkw_storage_sorter_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                       uint64_t *n_violations, uint64_t *first_bad);

/* ---- StorageApplication witness builder (a17) --------------------------------------------------------- */
typedef struct zkw_storage_application_witness zkw_storage_application_witness;
/* decompose_into_storage_application_witnesses, src/witness/individual_circuits/storage_application.rs:31-361.
   queries / query_tails: the deduplicated rollup storage queries and their queue states (ZKW_STO_RESULT_QUERIES /
   ZKW_STO_RESULT_NEW_TAILS of the storage sorter, pushed from the empty queue); every slot occurs once.
   The reference walks a `BinarySparseStorageTree` (src/witness/tree/mod.rs:41-98) query by query. This entry
   takes what that tree answers for the state BEFORE the block, per query i: init_leaf_indexes[i] = get_leaf(slot)
   .leaf.current_index() (0 = empty leaf), init_merkle_paths[i][256][32] = its merkle_path (level 0 = the leaf's
   sibling), plus root() and next_enumeration_index(); paths, roots and enumeration indices as the sequential
   walk sees them are rebuilt on the device. capacity = cycles_per_storage_application. With n == 0 one dummy
   instance is produced. The caller applies the writes to its own tree (new root: last instance's output).
   ZKW_ERR_CHECK_FAILED when a pre-state proof does not lead to initial_root, a read value differs from the leaf
   (storage_application.rs:221, 276) or a slot repeats. */
int zkw_storage_application_build(zkw_ctx *ctx, const zkw_log_query *queries, const uint64_t *query_tails, size_t n,
                                  const uint64_t *init_leaf_indexes, const uint8_t *init_merkle_paths,
                                  const uint8_t initial_root[32], uint64_t initial_next_enumeration_index,
                                  uint32_t capacity, zkw_storage_application_witness **out);
enum {
    ZKW_SAP_DERIVED_KEYS = 0, /* uint8_t[n][32]: LogQuery::derive_final_address */
    ZKW_SAP_MERKLE_PATHS = 1, /* uint8_t[n][256][32]: merkle_paths as seen when query i is applied */
    ZKW_SAP_LEAF_INDEXES = 2, /* uint64_t[n]: leaf_indexes_for_reads (the index BEFORE a write) */
    ZKW_SAP_ROOTS = 3,        /* uint8_t[n][32]: tree root after query i */
    ZKW_SAP_INSTANCES = 4     /* zkw_storage_application_instance[n_instances] */
};
size_t zkw_storage_application_witness_num_instances(const zkw_storage_application_witness *w);
size_t zkw_storage_application_witness_bytes(const zkw_storage_application_witness *w, int what);
const void *zkw_storage_application_witness_device_ptr(const zkw_storage_application_witness *w, int what);
int zkw_storage_application_witness_get(const zkw_storage_application_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_storage_application_witness_free(zkw_storage_application_witness *w);

/* ---- keccak256 / sha256 / ecrecover round-function witness builders (a16) ---------------------------- */
typedef struct zkw_precompile_witness zkw_precompile_witness;
/* kind = ZKW_PRECOMPILE_KECCAK256: keccak256_decompose_into_per_circuit_witness,
                                    src/witness/individual_circuits/keccak256_round_function.rs:23-528
          ZKW_PRECOMPILE_SHA256:    sha256_decompose_into_per_circuit_witness, .../sha256_round_function.rs:23-406
          ZKW_PRECOMPILE_ECRECOVER: ecrecover_decompose_into_per_circuit_witness, .../ecrecover.rs:12-262
   requests / request_tails: the demuxed precompile queue and its states (ZKW_DMX_OUT_QUERIES / _NEW_TAILS of
   that queue, pushed from the empty queue). mem_queries: the memory accesses of all requests back to back in
   the order the reference flattens the VM's round witnesses (keccak :44-65, sha256 :45-59, ecrecover :31-41):
   per request its reads round by round, then its write(s); n_queries must equal what the requests' ABIs imply.
   capacity = rounds per instance; mem_in (host): state of the global memory queue before the call. The queries
   are appended to that queue (ZKW_PRC_MEM_*). With n_requests == 0 one dummy instance is produced.
   ZKW_ERR_CHECK_FAILED when a query contradicts its request (read/write flag, keccak word index). */
int zkw_precompile_build(zkw_ctx *ctx, int kind, const zkw_log_query *requests, const uint64_t *request_tails,
                         size_t n_requests, const zkw_mem_query *mem_queries, size_t n_queries, uint32_t capacity,
                         const zkw_queue_state12 *mem_in, zkw_precompile_witness **out);
/* given_mem_tails[n_queries][12] (NULL = hash them): see zkw_decommitter_build_with_tails */
int zkw_precompile_build_with_tails(zkw_ctx *ctx, int kind, const zkw_log_query *requests, const uint64_t *request_tails,
                                    size_t n_requests, const zkw_mem_query *mem_queries, size_t n_queries,
                                    uint32_t capacity, const zkw_queue_state12 *mem_in, const uint64_t *given_mem_tails,
                                    zkw_precompile_witness **out);
enum {
    ZKW_PRC_MEM_ENC = 0,   /* uint64_t[n_queries][8]  */
    ZKW_PRC_MEM_TAILS = 1, /* uint64_t[n_queries][12] */
    ZKW_PRC_INSTANCES = 2, /* zkw_precompile_instance[max(1, ceil(total_rounds/capacity))] */
    ZKW_PRC_KECCAK_ROUNDS = 3, /* keccak256 only: zkw_keccak_round_record[total_rounds], the cycles of the type-5 circuit */
    ZKW_PRC_SHA256_ROUNDS = 4  /* sha256 only: zkw_sha256_round_record[total_rounds], the cycles of the type-6 circuit */
};
size_t zkw_precompile_witness_num_instances(const zkw_precompile_witness *w);
size_t zkw_precompile_witness_num_rounds(const zkw_precompile_witness *w);
size_t zkw_precompile_witness_bytes(const zkw_precompile_witness *w, int what);
const void *zkw_precompile_witness_device_ptr(const zkw_precompile_witness *w, int what);
int zkw_precompile_witness_get(const zkw_precompile_witness *w, int what, void *dst, size_t dst_bytes);
void zkw_precompile_witness_free(zkw_precompile_witness *w);

/* ---- callstack (a3 / a6) ------------------------------------------------------------------------------ */
/* ExtendedCallstackEntry::encoding_witness, circuit_encodings/src/callstack_entry.rs:36-179. enc: n*32 */
int zkw_encode_callstack_entries(zkw_ctx *ctx, const zkw_callstack_entry *entries, size_t n, uint64_t *enc);
/* FullWidthStackSimulator::{push,pop}_and_output_intermediate_data (circuit_encodings/src/lib.rs:558-644) for a
   whole sequence of operations on an initially empty CallstackSimulator: is_push[i] != 0 pushes the next entry
   of `pushed` (in order), 0 pops the top. Outputs per operation i (FullWidthStackIntermediateStates,
   lib.rs:511-521): previous_state[i][12], new_state[i][12], depth[i] (= num_items after the operation),
   round_states[i][4][12] (the outputs of round_function_execution_pairs; their inputs are the previous output
   with the rate replaced by the encoding chunk), entry_index[i] = index in `pushed` of the element pushed /
   returned by the pop. ZKW_ERR_INVALID on a pop from the empty stack (the reference's unwrap at lib.rs:619)
   or when pushes outnumber n_pushed. */
int zkw_callstack_simulate(zkw_ctx *ctx, const uint8_t *is_push, size_t n_ops, const zkw_callstack_entry *pushed,
                           size_t n_pushed, uint64_t *previous_state, uint64_t *new_state, uint32_t *depth,
                           uint64_t *round_states, uint32_t *entry_index);

/* ---- Keccak256RoundFunction circuit (type 5) ----------------------------------------------------------------
   ZkSyncBaseLayerCircuit::synthesis for the keccak256 round function on the wrapper's geometry and table set (circuit_definitions/src/
   circuit_definitions/base_layer/keccak256_round_function.rs:28-39,120-140: 86 copy columns, width-3 lookups x 14 per row with one
   table id per row, Xor8 / And8 / ByteSplit<1..4> = 132 096 table rows = vk_5.json's total_tables_len, ONE multiplicity column; 2^20
   rows, capacity 293). Trace "zkw trace v4" (include/zkw_netlist.h, include/zkw_keccak_circuit_spec.h): 129 columns
   (zkw_trace_create_with_columns(.., 129, ..)), per cycle a netlist stating out = idle ? prev : Keccak-f[1600]((reset ? 0 : prev) ^
   block); cycles = the instance's rounds (ZKW_PRC_KECCAK_ROUNDS), idle up to the capacity the witness was built with. Below the PI row:
   the QUEUE SECTION (include/zkw_netlist_queue.h) — per cycle the pop of the precompile call, up to six memory reads and the digest
   write as Poseidon2 rows, the queues chained through the trace, the written value's bytes copies of the sponge state. n_rows >=
   max(262 144, zkw_circuit_layout_of(5, capacity).rows_used). The check re-derives every relation from the cells: table membership,
   copy constraints, gates, headers, boundary rows, multiplicities, the section's encodings / permutations / selections / chains.
   w must be a keccak256 witness (zkw_precompile_build(ctx, ZKW_PRECOMPILE_KECCAK256, ..)) whose write queries hold the digests. */
/* compact closed-form inputs [n_instances][18] and public inputs [n_instances][4] of a precompile witness (any kind), DEVICE
   pointers valid until the witness is freed; computed on ctx's stream at the first call and kept with the witness (the
   type-5 synthesis writes them into the PI rows). Either out pointer may be NULL. */
int zkw_precompile_closed_forms(zkw_ctx *ctx, zkw_precompile_witness *w, const uint64_t **compact, const uint64_t **public_inputs);
int zkw_keccak_round_synthesize(zkw_ctx *ctx, zkw_precompile_witness *w, size_t first_instance, size_t n_instances,
                                zkw_trace *t, size_t first_slot);
int zkw_keccak_round_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                     uint64_t *n_violations, uint64_t *first_bad);

/* ---- ECRecover circuit (type 7) ----------------------------------------------------------------------------
   ZkSyncBaseLayerCircuit::synthesis for the ecrecover precompile on the wrapper's geometry and table set (circuit_definitions/src/
   circuit_definitions/base_layer/ecrecover.rs:30-41: 80 copy columns, width-3 lookups x 16 per row; :138-176: Xor8, And8, the 8 x 32
   FixedBaseMulTable<i, C>, ByteSplit<1..4> = 197 632 table rows = vk_7.json's total_tables_len; 2^20 rows, capacity 7 requests). One
   cycle per request (src/witness/individual_circuits/ecrecover.rs:143-178: 4 reads, 2 writes). Trace = three parts
   (tools/gen_ecrecover_circuit.py): the Keccak-f netlist over the recovered public key ("zkw trace v4",
   include/zkw_ecrecover_circuit_spec.h), the queue section (pop of the call, four reads, two writes: include/zkw_netlist_queue.h) and the
   EC SECTION (include/zkw_ecrecover.h, zkw_ecrecover_ec_spec.h): secp256k1 over 16-bit limbs in field-element-valued rows — r / s range
   flags, the square root of x^3 + 7, u1 = h / r and u2 = s / r, 256 double-and-add steps over the bits of u2, 32 table additions over
   the bytes of u1 (FixedBaseMul lookups), normalisation; the read values are the section's inputs, the written values the netlist's
   masked digest and the success flag, all by copy constraints. A request whose accumulator meets x1 == x2 in the (incomplete, affine)
   addition has no witness: ZKW_ERR_CHECK_FAILED. w must be an ecrecover witness. n_rows >= zkw_circuit_layout_of(7, capacity).rows_used
   (>= 197 632: the tables). Context scratch: 4 MB of value tape per request of the call (the cycles of an instance interleaved: capacity
   rounded up to 8 tapes per instance), 27 KB of Jacobian points per request, 240 KB of lookup keys per request. The call uses the
   context's side stream (the EC section's rows are written beside the netlist's fill) and returns with both streams' work queued and the
   status read: the slots are the caller's when it returns. */
int zkw_ecrecover_synthesize(zkw_ctx *ctx, zkw_precompile_witness *w, size_t first_instance, size_t n_instances, zkw_trace *t,
                             size_t first_slot);
/* Every instance of every witness (e.g. one ECRecover witness per block), in order, into slots first_slot ..: ONE launch of every EC
   kernel over all of them. The witnesses may belong to other contexts of ctx's device; their builders must have finished. */
int zkw_ecrecover_synthesize_multi(zkw_ctx *ctx, zkw_precompile_witness *const *witnesses, size_t n_witnesses, zkw_trace *t,
                                   size_t first_slot);
int zkw_ecrecover_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity, uint64_t *n_violations,
                                  uint64_t *first_bad);

/* ---- Sha256RoundFunction circuit (type 6) ------------------------------------------------------------------
   ZkSyncBaseLayerCircuit::synthesis for the sha256 round function on the wrapper's geometry and table set (circuit_definitions/src/
   circuit_definitions/base_layer/sha256_round_function.rs:28-39,121-134: 116 copy columns, width-4 lookups x 9 per row, TriXor4 / Ch4 /
   Maj4 / Split4BitChunk<1,2> = 12 320 table rows = vk_6.json's total_tables_len; 2^20 rows, capacity 2206). Trace "zkw trace v4",
   include/zkw_sha256_circuit_spec.h: 153 columns, per cycle a netlist on 4-bit chunks stating out = idle ? prev :
   sha256_compress(reset ? IV : prev, block); cycles = the instance's rounds (ZKW_PRC_SHA256_ROUNDS), idle up to the witness's
   capacity. Below the PI row the QUEUE SECTION (include/zkw_netlist_queue.h): per cycle the pop of the precompile call (first round of
   a request), the two memory reads whose value nibbles are copies of the block's message nibbles, the write of the digest (last
   round; value nibbles = copies of the state after the cycle) as Poseidon2 rows. A write query that does not hold the digest makes
   the trace unsatisfied. n_rows >= zkw_circuit_layout_of(6, capacity).rows_used. w must be a sha256 witness.
   Context scratch per instance of the call (kept by the context for the next call): multiplicity histogram slices and 2 bytes per
   lookup of keys; the same holds for the type-5 / 13 / 3 / 10 calls. */
int zkw_sha256_round_synthesize(zkw_ctx *ctx, zkw_precompile_witness *w, size_t first_instance, size_t n_instances,
                                zkw_trace *t, size_t first_slot);
int zkw_sha256_round_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                     uint64_t *n_violations, uint64_t *first_bad);

/* ---- CodeDecommitter circuit (type 3) ----------------------------------------------------------------------
   ZkSyncBaseLayerCircuit::synthesis for the code decommitter on the wrapper's geometry (circuit_definitions/src/circuit_definitions/
   base_layer/code_decommitter.rs:28-39,121-134: 108 copy columns, width-4 lookups x 11 per row, the SHA-256 table set = 12 320 rows;
   2^20 rows, capacity 2845 SHA-256 rounds). Trace "zkw trace v4", include/zkw_code_decommitter_circuit_spec.h: the SHA-256 netlist of
   the type-6 trace at this geometry (153 columns); one cycle per round of the unpacked bytecodes (ZKW_DCM_SHA256_ROUNDS: block as
   hashed incl. the final padding, reset at a bytecode's first round, state after), idle beyond the instance's rounds; below the PI row
   the QUEUE SECTION (include/zkw_netlist_queue.h): per cycle the pop of the decommit request (a bytecode's first round) and the memory
   writes of the round's two code words (one in a bytecode's last round), value nibbles = copies of the block's message nibbles.
   w: the witness of zkw_decommitter_build. n_rows >= zkw_circuit_layout_of(3, capacity).rows_used. */
int zkw_code_decommitter_synthesize(zkw_ctx *ctx, zkw_decommitter_witness *w, size_t first_instance, size_t n_instances,
                                    zkw_trace *t, size_t first_slot);
int zkw_code_decommitter_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                         uint64_t *n_violations, uint64_t *first_bad);

/* ---- StorageApplication circuit (type 10) ----------------------------------------------------------------
   ZkSyncBaseLayerCircuit::synthesis for the storage application (wrapper circuit_definitions/src/circuit_definitions/
   base_layer/storage_apply.rs:28-39: 60 + 3 x 26 columns; :124-140: Xor8, And8, ByteSplit<1, 2, 3, 4, 7> = 132 352 table rows =
   vk_10.json's total_tables_len; 2^20 rows, capacity 33 tree queries). Trace "zkw trace v4",
   include/zkw_storage_application_circuit_spec.h: the Merkle walks of the instance's tree queries — a read is one walk, a write
   two (old leaf, new leaf), src/witness/individual_circuits/storage_application.rs:141-153 — as Blake2s-256 compressions on
   bytes, one cycle per compression, SA_CYCLES_PER_WALK = 257 cycles per walk (leaf hash, 256 levels; the key is shifted
   through the cycle state and its low bit picks the sibling's side), idle cycles beyond the instance's walks;
   122 rows per cycle = 33 walks in 2^20 rows. w: the witness of zkw_storage_application_build (its capacity). The public
   input is the instance's closed-form commitment (a20). n_rows >= 2^18 (the tables' 132 352 rows). */
int zkw_storage_application_synthesize(zkw_ctx *ctx, zkw_storage_application_witness *w, size_t first_instance,
                                       size_t n_instances, zkw_trace *t, size_t first_slot);
int zkw_storage_application_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                                            uint64_t *n_violations, uint64_t *first_bad);

/* ---- LinearHasher circuit (type 13) ---------------------------------------------------------------------
   ZkSyncBaseLayerCircuit::synthesis for the L1-messages hasher (wrapper base_layer/linear_hasher.rs:28-138, witness
   compute_linear_keccak256 data_hasher_and_merklizer.rs:8-67): the Keccak-f netlist of the type-5 trace over the sponge that
   hashes the n <= capacity serialized messages (host or device pointer per the pointer mode); cycles =
   ZKW_LINEAR_HASHER_CYCLES(capacity), idle beyond the message's rounds; BND_OUT's first 32 bytes are the pubdata hash.
   queue_state (host): the state of the deduplicated L1-messages queue = the closed form's observable input. record_out /
   public_input_out (host): the instance record and its public input [4] (may be NULL). Its own geometry (66 + 3 x 26 + 1
   columns: zkw_circuit_layout_of(13)); check with zkw_linear_hasher_check_satisfied(ctx, t, slot, capacity, ..). */
int zkw_linear_hasher_synthesize(zkw_ctx *ctx, const zkw_log_query *messages, size_t n, const zkw_queue_state4 *queue_state,
                                 uint32_t capacity, zkw_trace *t, size_t slot, zkw_linear_hasher_instance *record_out,
                                 uint64_t *public_input_out);
/* The same for n_queues independent message queues in ONE call (the L1-messages queues of several blocks): queue b = messages
   [message_offsets[b], message_offsets[b + 1]) -> slot first_slot + b, records_out[b], public_inputs_out[4 b ..]. The sponge of
   a queue is serial (one Keccak-f per 136 bytes); a batch runs the queues' sponges side by side and fills all traces with one
   launch. */
int zkw_linear_hasher_synthesize_batch(zkw_ctx *ctx, const zkw_log_query *messages, const uint64_t *message_offsets, size_t n_queues,
                                       const zkw_queue_state4 *queue_states, uint32_t capacity, zkw_trace *t, size_t first_slot,
                                       zkw_linear_hasher_instance *records_out, uint64_t *public_inputs_out);
/* The same with the queues' states given: message_tails [total][4] = the state of a queue after each of its messages was pushed (the
   events sorter's ZKW_EVT_RESULT_NEW_TAILS) — the heads the circuit's pops run through in the trace's queue section
   (include/zkw_netlist_queue.h: every message is popped as three Poseidon2 permutations below the netlist, the head chained from
   queue_states[b].head). NULL: hashed here, one serial Poseidon2 chain per queue (what zkw_linear_hasher_synthesize[_batch] do). */
int zkw_linear_hasher_synthesize_batch_with_tails(zkw_ctx *ctx, const zkw_log_query *messages, const uint64_t *message_offsets, size_t n_queues,
                                                  const zkw_queue_state4 *queue_states, const uint64_t *message_tails, uint32_t capacity, zkw_trace *t,
                                                  size_t first_slot, zkw_linear_hasher_instance *records_out, uint64_t *public_inputs_out);

/* ---- public inputs and the recursion queue (a20) ---------------------------------------------------- */
/* commit_variable_length_encodable_item as driven by simulate_public_input_value_from_witness
   (src/witness/utils.rs:269-306): n_items flat encodings of item_len elements each -> out[n_items][4]. */
int zkw_commit_encodings(zkw_ctx *ctx, const uint64_t *enc, size_t n_items, uint32_t item_len, uint64_t *out);
/* RecursionRequest::encoding_witness, circuit_encodings/src/recursion_request.rs:13-28: enc[i] =
   [circuit_type, pi[i][0..4], 0, 0, 0]; feed the result to zkw_queue_push_chain_full to obtain the
   RecursionQueueSimulator states CircuitMaker::process produces (postprocessing/mod.rs:393-400). */
int zkw_encode_recursion_requests(zkw_ctx *ctx, uint64_t circuit_type, const uint64_t *public_inputs, size_t n,
                                  uint64_t *enc /* n*8 */);

/* ClosedFormInputCompactForm::from_full_form + the public input (postprocessing/mod.rs:353-369) for the circuits whose
   builders return instance records without compact forms: circuit_type 3 (zkw_decommitter_instance), 5 / 6 / 7
   (zkw_precompile_instance of that kind), 10 (zkw_storage_application_instance), 13 (zkw_linear_hasher_instance). The
   observable input of instance i is the one of the latest instance <= i with start_flag set, as in the reference.
   compact: [n][18] = start, completion, 4 x 4 commitment words; public_inputs: [n][4]. Types 2, 4, 8, 9, 11, 12 carry
   theirs in the witness (ZKW_*_COMPACT_FORMS / ZKW_*_PUBLIC_INPUTS); the MainVM closed form needs the VM's local state,
   which this library does not see. */
int zkw_closed_form_public_inputs(zkw_ctx *ctx, uint8_t circuit_type, const void *instances, size_t n, uint64_t *compact,
                                  uint64_t *public_inputs);

/* The leaf layer's view of a recursion queue: RecursionQueueSimulator::split_by(RECURSION_ARITY = 32) as create_leaf_witnesses
   uses it (src/witness/recursive_aggregation.rs:98-117): states = the [n][12] queue states after each request (host; from
   zkw_queue_push_chain_full or zkw_block_recursion_states), leaf_states[k] = RecursionLeafInputWitness::queue_state of leaf k
   (head = the state before its first request, tail = the state after its last, length = its requests); the leaf's
   queue_witness elements are the requests' encodings with old_tail = the state before each. Host arithmetic only. */
#define ZKW_RECURSION_ARITY 32
int zkw_recursion_queue_split(const uint64_t *states, size_t n, uint32_t arity, zkw_queue_state12 *leaf_states, size_t max_leaves,
                              size_t *n_leaves);

/* ---- recursion layer witnesses (src/witness/recursive_aggregation.rs; BASELINE config #5, witness side) --------------
   All arrays are HOST arrays (a few hundred bytes per leaf): call these on a context in ZKW_PTR_HOST mode. The hashing runs
   in the library's kernels. The encodings are pinned by three committed leaf proofs of the reference
   (tests/golden/leaf_layer_kat.json). */
/* records: zkw_leaf_params, zkw_queue_tail12 (include/zkw_types.h) */
/* commitment of a verification key = compute_encodable_item_from_witness::<AllocatedVerificationKey> (:45-68; used at
   :184-206 and by compute_node_vk_commitment :242-268): setup_merkle_tree_cap[cap_size][4] of a vk_N.json -> out[4] */
int zkw_vk_commitment(zkw_ctx *ctx, const uint64_t *setup_merkle_tree_cap, size_t cap_size, uint64_t out[4]);
/* compute_leaf_params (:163-216): circuit_type = the base-layer type, its VK's cap and the cap of the leaf-layer VK that
   aggregates it (recursion type = base type + 2) */
int zkw_compute_leaf_params(zkw_ctx *ctx, uint8_t circuit_type, const uint64_t *base_layer_cap, const uint64_t *leaf_layer_cap,
                            size_t cap_size, zkw_leaf_params *out);
/* compute_leaf_vks_and_params_commitment (:218-240): leaf_params[13] in base circuit type order -> out[4] */
int zkw_leaf_vks_and_params_commitment(zkw_ctx *ctx, const zkw_leaf_params *leaf_params, uint64_t out[4]);
/* create_leaf_witnesses (:71-161) for one circuit type: public_inputs[n][4] of its base-layer instances in emission order
   -> the recursion queue (enc[n][8], states[n][12]: FullStateCircuitQueueRawWitness element i = (request i, old tail =
   states[i - 1] or queue_tail_in/zero)), split_by(32) into leaf_states[*n_leaves] and, if leaf_public_inputs != NULL, each
   leaf circuit's public input = commit(RecursionLeafInput{params, queue_state}) [*n_leaves][4]. queue_tail_in: NULL = empty. */
int zkw_create_leaf_witnesses(zkw_ctx *ctx, const zkw_leaf_params *params, const uint64_t *public_inputs, size_t n,
                              const uint64_t *queue_tail_in, uint64_t *enc, uint64_t *states, zkw_queue_state12 *leaf_states,
                              uint64_t *leaf_public_inputs, size_t max_leaves, size_t *n_leaves);
/* create_node_witnesses (:270-421) for one circuit type: chunks[n_chunks] = the queue states of the leaves (or nodes) below,
   in order; every 32 are merged into node_states[k] (heads and tails must chain), split_points[k][31] (tails of the first 31
   chunks, padded with (merged tail, 0)) and, if node_public_inputs != NULL, the node circuit's public input =
   commit(RecursionNodeInput{branch_circuit_type, leaf_layer_parameters[13], node_layer_vk_commitment, queue_state}). */
int zkw_create_node_witnesses(zkw_ctx *ctx, uint8_t branch_circuit_type, const zkw_leaf_params *leaf_layer_params,
                              const uint64_t node_layer_vk_commitment[4], const zkw_queue_state12 *chunks, size_t n_chunks,
                              zkw_queue_state12 *node_states, zkw_queue_tail12 *split_points, uint64_t *node_public_inputs,
                              size_t max_nodes, size_t *n_nodes);

int zkw_linear_hasher_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity, uint64_t *n_violations,
                                     uint64_t *first_bad);
/* ---- L1 messages hasher ------------------------------------------------------------------------------ */
/* compute_linear_keccak256, src/witness/individual_circuits/data_hasher_and_merklizer.rs:8-67: Keccak-256 of
   the concatenated 88-byte serialisations (circuit_encodings/src/log_query.rs:503-534) of the net L2->L1
   messages (ZKW_EVT_RESULT_QUERIES of the L1-messages sorter). hash_out: 32 bytes (host, or device in
   ZKW_PTR_DEVICE mode). The instance's queue witness/state is the sorter's result queue itself. */
int zkw_linear_keccak256(zkw_ctx *ctx, const zkw_log_query *messages, size_t n, uint8_t *hash_out);

/* ---- synthesis: filled traces ------------------------------------------------------------------- */
/* A zkw_trace owns n_slots trace buffers in HBM, each column-major uint64_t[n_cols][n_rows] (n_rows =
   2^20 at production geometry = TARGET_CIRCUIT_TRACE_LENGTH, base_layer/mod.rs:17). It plays the role of
   the reference's CSReferenceAssembly (output of `synthesis`, base_layer/mod.rs:315-323): variable
   columns, lookup columns and the lookup-multiplicity column, every cell written. Slots are a ring: a
   prover consumes a slot while later instances are synthesised into the others.
   A trace may be filled / checked through any context of the same device (the kernels run on THAT context's stream;
   ordering between streams that touch one slot is the caller's business). */
int zkw_trace_create(zkw_ctx *ctx, size_t n_rows, size_t n_slots, zkw_trace **out);
/* same with an explicit column count (zkw_trace_create = 149, the RAMPermutation geometry): a circuit type whose
   geometry is wider — the LogDemuxer's 136 + 14 + 1 = 151 — needs its own trace */
int zkw_trace_create_with_columns(zkw_ctx *ctx, size_t n_rows, size_t n_cols, size_t n_slots, zkw_trace **out);
void zkw_trace_free(zkw_trace *t);
size_t zkw_trace_num_rows(const zkw_trace *t);
size_t zkw_trace_num_cols(const zkw_trace *t); /* default 149 = 133 copy-permutation + 15 lookup + 1 multiplicity */
size_t zkw_trace_num_slots(const zkw_trace *t);
/* device address of slot's column 0; column c starts at + c * n_rows.
   CONTRACT: the library remembers which layout a slot last held (a synthesis of any circuit type into a slot that still holds the
   same circuit / capacity / row count stores only the cells that can differ between traces of that layout and relies on the rest
   still being zero); taking this pointer FORGETS that — the caller may write through it — so
   the next synthesis into the slot writes every cell again. A pointer kept from an earlier call must therefore not be written
   through after a later synthesis into the slot: take the pointer again (or read with zkw_trace_get, which keeps the slot's
   state). Reading through a kept pointer is always fine. */
const uint64_t *zkw_trace_device_ptr(const zkw_trace *t, size_t slot);
/* copy columns [first_col, first_col + n_cols) of a slot out (host, or device in ZKW_PTR_DEVICE mode) */
int zkw_trace_get(const zkw_trace *t, size_t slot, uint32_t first_col, uint32_t n_cols, uint64_t *dst);

/* ZkSyncBaseLayerCircuit::synthesis for RAMPermutation instances (base_layer/mod.rs:286-323 with
   base_layer/ram_permutation.rs:26-135): instances [first_instance, first_instance + n_instances) of a
   witness are materialised into slots (first_slot + k) % n_slots. Layout: include/zkw_ram_circuit_spec.h.
   Requires RC_MIN_ROWS(capacity) <= n_rows. */
int zkw_ram_synthesize(zkw_ctx *ctx, const zkw_ram_witness *w, size_t first_instance, size_t n_instances,
                       zkw_trace *t, size_t first_slot);
/* check_if_satisfied (src/tests/mod.rs:130-259) for one RAMPermutation trace: every row constraint, the
   Poseidon2 rows, copy links, lookup ranges and multiplicities, zero padding, canonical cells.
   n_violations = number of failed relations; first_bad = (kind << 56) | (index << 32) | row of the
   smallest failing code (kind 1 constraint, 2 poseidon row, 3 cell range, 4 copy link, 5 multiplicity,
   6 padding). Synchronises the stream. */
int zkw_ram_check_satisfied(zkw_ctx *ctx, const zkw_trace *t, size_t slot, uint32_t capacity,
                            uint64_t *n_violations, uint64_t *first_bad);

/* ---- type-dispatching synthesis (8b) ------------------------------------------------------------------------------ */
/* BaseLayerCircuitType numbering (ZkSyncBaseLayerCircuit::numeric_circuit_type, base_layer/mod.rs:442-476) */
enum {
    ZKW_CIRCUIT_MAIN_VM = 1,
    ZKW_CIRCUIT_CODE_DECOMMITTMENTS_SORTER = 2,
    ZKW_CIRCUIT_CODE_DECOMMITTER = 3,
    ZKW_CIRCUIT_LOG_DEMUXER = 4,
    ZKW_CIRCUIT_KECCAK256_ROUND_FUNCTION = 5,
    ZKW_CIRCUIT_SHA256_ROUND_FUNCTION = 6,
    ZKW_CIRCUIT_ECRECOVER = 7,
    ZKW_CIRCUIT_RAM_PERMUTATION = 8,
    ZKW_CIRCUIT_STORAGE_SORTER = 9,
    ZKW_CIRCUIT_STORAGE_APPLICATION = 10,
    ZKW_CIRCUIT_EVENTS_SORTER = 11,
    ZKW_CIRCUIT_L1_MESSAGES_SORTER = 12,
    ZKW_CIRCUIT_L1_MESSAGES_HASHER = 13
};
/* the "witness" of the L1MessagesHasher for zkw_synthesize: its instances are whole queues (one instance per queue,
   data_hasher_and_merklizer.rs:34-60), the arguments of zkw_linear_hasher_synthesize_batch_with_tails as one record */
typedef struct zkw_linear_hasher_witness {
    const zkw_log_query *messages;        /* all queues back to back (host or device per the pointer mode) */
    const uint64_t *message_offsets;      /* [n_queues + 1] (host) */
    size_t n_queues;
    const zkw_queue_state4 *queue_states; /* [n_queues] (host) */
    const uint64_t *message_tails;        /* [total][4] or NULL (see zkw_linear_hasher_synthesize_batch_with_tails) */
    uint32_t capacity;                    /* messages per instance */
    zkw_linear_hasher_instance *records_out; /* [n_queues] or NULL */
    uint64_t *public_inputs_out;             /* [n_queues][4] or NULL */
} zkw_linear_hasher_witness;
/* ZkSyncBaseLayerCircuit::synthesis (base_layer/mod.rs:286-323: one `match` over the enum's variants): instances
   [first_instance, first_instance + n_instances) of `witness` into slots first_slot.. of `t`. `witness` is the witness handle
   of the type: 2 zkw_decommit_witness, 3 zkw_decommitter_witness, 4 zkw_demux_witness, 5 / 6 / 7 zkw_precompile_witness (of that
   kind), 8 zkw_ram_witness, 9 zkw_storage_witness, 10 zkw_storage_application_witness, 11 / 12 zkw_events_witness, 13
   zkw_linear_hasher_witness* (a record of pointers, above) — exactly what zkw_block_witness(b, type) returns for 2..12. The
   trace must have the type's column count (zkw_circuit_layout_of). Type 1 (MainVM) has no synthesis here: ZKW_ERR_INVALID. */
int zkw_synthesize(zkw_ctx *ctx, uint8_t circuit_type, const void *witness, size_t first_instance, size_t n_instances, zkw_trace *t,
                   size_t first_slot);
/* check_if_satisfied of base_test_circuit (src/tests/mod.rs:130-259) for a slot that holds an instance of `circuit_type` at
   `capacity` (type 13: messages): n_violations / first_bad as the per-type checkers report them. */
int zkw_check_satisfied(zkw_ctx *ctx, uint8_t circuit_type, const zkw_trace *t, size_t slot, uint32_t capacity, uint64_t *n_violations,
                        uint64_t *first_bad);

/* ---- MainVM instance slicing (a19) ------------------------------------------------------------------------------ */
/* The loop of src/witness/oracle.rs:1229-1469 (`vm_snapshots.windows(2)`): every window [at_cycle_k, at_cycle_{k+1}) of
   the VM's snapshots becomes one MainVM instance whose VmWitnessOracle FIFOs are the elements of eight cycle-stamped
   streams that fall into the window, whose entry states are "the last state with cycle < from" of the memory queue,
   the decommit queue, the callstack sponge and the storage-log history (:1245-1273, :1359-1375), and whose
   auxilary_final_parameters are the next instance's initial ones (the global final states on the last one, :1414-1468);
   plus the flags and observable input / output of vm_instance_witness_to_circuit_formal_input (src/witness/utils.rs:
   428-496). One lane per (instance, stream) does the binary searches the reference does with take_while / partition_point.
   instances: [n_snapshots - 1]; memory_read_index / memory_write_index: [stream_len[ZKW_VMS_MEMORY]] capacity each (the
   memory stream's reads / writes in order; n_reads / n_writes receive the totals; may be NULL together). The VM and its
   tracer themselves (which produce the streams) are out of scope. */
int zkw_vm_slice_instances(zkw_ctx *ctx, const zkw_vm_tracer_streams *streams, zkw_vm_instance *instances,
                           uint32_t *memory_read_index, uint32_t *memory_write_index, uint64_t *n_reads, uint64_t *n_writes);

/* ---- a19, pre-builder half: the tracer's raw record -> log queue with rollbacks, callstack replay ----------------------
   Counterpart of src/witness/callstack_handler.rs:174-460 (forward / rollback queues per frame, glued on `ret`, the rollback
   queue appended in reverse to the parent's forward queue on a panic) and src/witness/oracle.rs:233-843 (the flat queue
   hashed through one LogQueueSimulator, marker positions, rollback tails of new frames, rollback head segments, the
   storage-log state per cycle, the callstack entries with their rollback segments pushed / popped through the
   CallstackSimulator). events: time order, events[0] = the bootloader frame's push (from_initial_callstack), every frame
   popped at the end. HOST arrays; ctx in ZKW_PTR_HOST mode. The reference's panics are ZKW_ERR_CHECK_FAILED. */
typedef struct zkw_vm_trace zkw_vm_trace;
int zkw_vm_trace_build(zkw_ctx *ctx, const zkw_vm_event *events, size_t n_events, const zkw_log_query *log_queries, size_t n_logs,
                       const zkw_callstack_entry *entries, size_t n_entries, zkw_vm_trace **out);
enum {
    ZKW_VMT_FLAT_QUERIES = 0,           /* zkw_log_query[n_flat]: the flat queue, rollback twins with rollback = 1; the first
                                           original_log_queue_length items are the block's log queue (the demuxer's input) */
    ZKW_VMT_FLAT_CYCLES = 1,            /* uint32_t[n_flat] */
    ZKW_VMT_FLAT_FRAMES = 2,            /* uint32_t[n_flat]: the frame that issued the query */
    ZKW_VMT_FLAT_OLD_TAILS = 3,         /* uint64_t[n_flat][4]: chain_of_states previous_tail */
    ZKW_VMT_FLAT_NEW_TAILS = 4,         /* uint64_t[n_flat][4]: chain_of_states tail */
    ZKW_VMT_NEW_FRAME_TAIL_CYCLES = 5,  /* rollback_queue_initial_tails_for_new_frames: uint32_t[n_frames] ...          */
    ZKW_VMT_NEW_FRAME_TAILS = 6,        /* ... uint64_t[n_frames][4]                                                   */
    ZKW_VMT_HEAD_SEGMENT_CYCLES = 7,    /* rollback_queue_head_segments: ascending cycles ... */
    ZKW_VMT_HEAD_SEGMENTS = 8,          /* ... uint64_t[][4] */
    ZKW_VMT_STORAGE_LOG_STATE_CYCLES = 9,   /* history_of_storage_log_states: strictly ascending cycles ... */
    ZKW_VMT_STORAGE_LOG_STATE_FRAMES = 10,  /* ... frame_idx ... */
    ZKW_VMT_STORAGE_LOG_STATES = 11,        /* ... zkw_storage_log_detailed_state[] */
    ZKW_VMT_CALLSTACK_WITNESS_CYCLES = 12,  /* callstack_values_witnesses: one per push and per pop ... */
    ZKW_VMT_CALLSTACK_WITNESS_IS_PUSH = 13, /* uint8_t[] */
    ZKW_VMT_CALLSTACK_WITNESS_ENTRIES = 14, /* zkw_callstack_entry[] (ExtendedCallstackEntry: with the rollback segment) */
    ZKW_VMT_CALLSTACK_WITNESS_PREVIOUS_STATES = 15, /* uint64_t[][12] */
    ZKW_VMT_CALLSTACK_WITNESS_NEW_STATES = 16,      /* uint64_t[][12] */
    ZKW_VMT_CALLSTACK_WITNESS_DEPTHS = 17,          /* uint32_t[] */
    ZKW_VMT_CALLSTACK_WITNESS_ROUND_STATES = 18,    /* uint64_t[][4][12] */
    ZKW_VMT_CALLSTACK_SPONGE_CYCLES = 19,   /* callstack_sponge_encoding_ranges: (0, zero) first ... */
    ZKW_VMT_CALLSTACK_SPONGE_STATES = 20,   /* ... uint64_t[][12] */
    ZKW_VMT_NEW_FRAME_CYCLES = 21,          /* flat_new_frames_history ... */
    ZKW_VMT_NEW_FRAME_ENTRIES = 22          /* ... zkw_callstack_entry[] */
};
size_t zkw_vm_trace_count(const zkw_vm_trace *t, int what);
const void *zkw_vm_trace_ptr(const zkw_vm_trace *t, int what); /* host memory, valid until zkw_vm_trace_free */
int zkw_vm_trace_get(const zkw_vm_trace *t, int what, void *dst, size_t dst_bytes);
int zkw_vm_trace_info(const zkw_vm_trace *t, zkw_vm_trace_summary *out);
/* points the four FIFOs (streams 4-7) and the callstack-sponge / storage-log-state histories of `s` at this trace's arrays
   (host): with the VM's own streams added, `s` is the input of zkw_vm_slice_instances — the VmWitnessOracle FIFOs end to end */
int zkw_vm_trace_streams(const zkw_vm_trace *t, zkw_vm_tracer_streams *s);
void zkw_vm_trace_free(zkw_vm_trace *t);

/* ---- multi-GPU (8e): shard plan and the one collective ------------------------------------------------------- */
/* One process per GPU. Instances are independent once the builders have fixed their hidden FSM inputs, so they are
   sharded with no data-path collective; the only exchange is the gather of the per-instance closed-form records to the
   root, which replays the order-sensitive recursion-queue pushes (src/witness/postprocessing/mod.rs:396-402,
   src/external_calls.rs:354-537). The transport is RCCL (xGMI inside a node), loaded with dlopen when a communicator
   of more than one rank is created; a TCP transport (zkw_comm_init_tcp) runs the same collective between processes that
   have no GPU (tests) or where RCCL cannot start. */
/* owner[i] = rank of instance i: longest-processing-time over the rows the reference's synthesis of each circuit type
   uses (setup/base_layer/finalization_hint_N.json). Deterministic; needs no GPU. circuit_types: 1..13. */
int zkw_shard_lpt(const uint8_t *circuit_types, size_t n, int world, uint32_t *owner);
typedef struct zkw_comm zkw_comm;
#define ZKW_COMM_ID_BYTES 128
/* rank 0 creates the id (ncclGetUniqueId) and hands it to the other ranks over the host's own channel */
int zkw_comm_unique_id(uint8_t id[ZKW_COMM_ID_BYTES]);
/* collective over all ranks (ncclCommInitRank); world == 1 needs no id and no RCCL. Work runs on ctx's stream. */
int zkw_comm_init(zkw_ctx *ctx, const uint8_t id[ZKW_COMM_ID_BYTES], int rank, int world, zkw_comm **out);
/* the RCCL transport whatever the world size (world == 1: ncclCommInitRank over one rank; `id` is required): what lets a
   single-GPU host run the transport's dlopen, symbol table, stream ordering and error paths (tests/test_gpu_comm_rccl.py) */
int zkw_comm_init_rccl(zkw_ctx *ctx, const uint8_t id[ZKW_COMM_ID_BYTES], int rank, int world, zkw_comm **out);
/* one grouped send of src[bytes] to `peer` + receive of dst[bytes] from it through the communicator's transport (peer may be
   the own rank on RCCL: a send to self matched by a receive from self). Pointers as in zkw_gather_closed_form_inputs, not
   overlapping. Enqueued on the stream (zkw_comm_synchronize). */
int zkw_comm_exchange(zkw_comm *c, const void *src, void *dst, size_t bytes, int peer);
/* the same communicator over a full mesh of TCP sockets on ONE host: every rank binds and connects on `address`
   (use 127.0.0.1), rank j connects to every rank i < j at address:(port + i); the hello carries the 64-bit job id of the
   environment variable ZKW_COMM_JOB_ID (0 when unset) and connections of another job are turned away. A test / fallback
   transport for one trusted host, not an authenticated channel.
   ctx == NULL: a host-memory communicator (records / recv of zkw_gather_closed_form_inputs are HOST pointers; needs no GPU);
   ctx != NULL: device pointers, staged through host memory. timeout_ms <= 0: 30 s. Collective over all ranks. */
int zkw_comm_init_tcp(zkw_ctx *ctx, const char *address, int port, int rank, int world, int timeout_ms, zkw_comm **out);
void zkw_comm_destroy(zkw_comm *c);
/* everything this communicator has enqueued so far has completed (its context's stream, for the device transports) */
int zkw_comm_synchronize(zkw_comm *c);
/* counts[r] (host, every rank passes the same array) records of record_bytes each from rank r, concatenated in rank
   order into recv on `root` (NULL elsewhere). records / recv: DEVICE pointers (HOST pointers on a context-less TCP
   communicator). Enqueued on the stream; not synchronised (zkw_comm_synchronize). */
int zkw_gather_closed_form_inputs(zkw_comm *c, const void *records, const uint64_t *counts, size_t record_bytes, int root,
                                  void *recv);

/* The gather as a sequencer uses it. Every rank passes the same owner[n] (zkw_shard_lpt over the ordered instance list) and
   `mine` = the records of ITS instances in list order (HOST memory, record_bytes each); the root receives all n records in
   list (= emission) order in out (HOST, n * record_bytes; NULL elsewhere). Synchronous on every rank. */
int zkw_gather_records(zkw_comm *c, const uint32_t *owner, size_t n, const void *mine, size_t record_bytes, int root, void *out);

/* ---- one block: the post-VM half of create_artifacts_from_tracer (a19) ----------------------------------------- */
/* Counterpart of src/witness/oracle.rs:928-1130 + 1494-1732 (everything `create_artifacts_from_tracer` does after the
   VM has run, except the MainVM instances): every per-circuit witness builder over what the VM left behind, the shared
   queues threaded through them, public inputs and one recursion queue per circuit type, then the synthesis of every
   instance in the reference's emission order. The sequencer lives in the library (csrc/zkw_block.hip, written against
   this header only) so that a Rust host makes ONE call per block and nothing round-trips through host memory.

   Scheduling (why this is not the reference's order): every queue hash chain is serial, ~10 us per item on one wave,
   but independent chains run concurrently for free. The builders are therefore run as a dependency graph on their own
   contexts / HIP streams / host threads: contents first (sorts, routing, deduplication), then all hash chains side by
   side; the block's memory queue (VM, code words, keccak256, sha256, ecrecover queries) is assembled up front and
   hashed ONCE, inside the RAM-permutation builder, and the decommitter / precompile builders take their slices of it.
   Results are identical to calling the builders one by one (tests/test_gpu_block.py). */
typedef struct zkw_block zkw_block;
/* the reference's `tree: impl BinarySparseStorageTree` (src/external_calls.rs:81): asked once, for the deduplicated
   rollup storage queries of the block in their final order: leaf_indexes[n] (0 = empty leaf) and merkle_paths[n][256][32]
   of the state BEFORE the block (see zkw_storage_application_build). Return 0 on success. Called on the thread that runs
   the storage branch. */
typedef int (*zkw_storage_tree_fn)(void *user, const zkw_log_query *dedup_queries, size_t n, uint64_t *leaf_indexes,
                                   uint8_t *merkle_paths);
typedef struct zkw_block_inputs {
    const zkw_mem_query *vm_memory_queries;      /* the VM's memory queue in order (oracle.rs:894-903) */
    size_t n_vm_memory_queries;
    const zkw_decommit_query *decommit_queries;  /* decommit requests in order (oracle.rs:928-945), n > 0 */
    size_t n_decommit_queries;
    /* the bytecodes behind the decommit hashes: code k has hash bytecode_hashes[k][8] (limbs as zkw_decommit_query.hash)
       and owns 32-byte words [bytecode_word_offsets[k], bytecode_word_offsets[k+1]) of bytecode_words[..][8] */
    const uint32_t *bytecode_hashes;
    const uint32_t *bytecode_words;
    const uint64_t *bytecode_word_offsets;
    size_t n_bytecodes;
    const zkw_log_query *log_queries;            /* forward-applied log queue (oracle.rs:308-350) */
    size_t n_log_queries;
    const zkw_mem_query *precompile_memory_queries[3]; /* keccak256, sha256, ecrecover (see zkw_precompile_build) */
    size_t n_precompile_memory_queries[3];
    uint32_t num_non_deterministic_heap_queries;
    zkw_storage_tree_fn storage_tree;            /* NULL = no StorageApplication instances are built */
    void *storage_tree_user;
    uint8_t storage_initial_root[32];
    uint64_t storage_initial_next_enumeration_index;
    uint32_t capacities[14];                     /* per BaseLayerCircuitType; 0 = geometry_config.rs default */
    /* optional (NULL = no MainVM records): the tracer's cycle-stamped vectors for the MainVM instance slicing
       (zkw_vm_slice_instances). HOST pointers. stream_len[ZKW_VMS_MEMORY] must equal n_vm_memory_queries and
       n_decommit_states must equal n_decommit_queries: the block supplies vm_memory_queries, memory_queue_tails and
       decommit_queue_tails itself, from the states it has just hashed (the fields of the same name are ignored). */
    const zkw_vm_tracer_streams *vm_tracer;
    /* nonzero: the four QUEUES — vm_memory_queries, decommit_queries, log_queries, precompile_memory_queries[] — are DEVICE pointers
       (a VM that runs next to the library leaves them in HBM); they are read, never written, and must stay valid until the block is
       freed. The bytecodes and vm_tracer stay HOST pointers. 0 (the default): everything is host memory. */
    uint32_t queues_on_device;
} zkw_block_inputs;
/* All pointers in `in` are HOST pointers (but see queues_on_device). Blocks until every builder has finished. */
int zkw_block_run(int device_id, const zkw_block_inputs *in, zkw_block **out);
/* n_blocks independent blocks at once (a witness-generation service's batch): the same graph per block, but no host thread and no
   stream per block — the blocks' builder branches run as fibers of the calling thread, and the launches they make of the same kernel
   leave as ONE launch over a job table, a stage's queue chains as one chain launch (csrc/zkw_batch.h): K blocks cost about one block's
   chain pass while SIMDs and memory last (~0.3 GB of HBM per production-capacity block). Each block's results are what zkw_block_run
   gives for it alone. out[k] receives block k; on failure every block is released and out[] is all NULL. */
int zkw_blocks_run(int device_id, const zkw_block_inputs *const *inputs, size_t n_blocks, zkw_block **out);
/* The same on one rank of a multi-GPU job, the BLOCKS sharded (the mode that scales: nothing is replicated): rank r builds the
   blocks k with zkw_blocks_owner(k, world) == r (round-robin) and leaves out[k] = NULL for the others. Every rank passes the same
   inputs array (pointers of blocks it does not own are only validated). No communication. */
int zkw_blocks_owner(size_t block, int world);
int zkw_blocks_run_sharded(int device_id, const zkw_block_inputs *const *inputs, size_t n_blocks, int rank, int world, zkw_block **out);
/* Collective over `comm` after zkw_blocks_run_sharded: the closed-form records of every block reach `root` in block order —
   out[n_blocks][1 + 24 * max_per_block] (host, root only): word 0 = the block's instance count n, then n records
   [circuit_type, instance, compact form (18), public input (4)] in emission order, zero padded — for the recursion-queue
   replay of each block (postprocessing/mod.rs:396-402). blocks[k] is read only for the blocks this rank owns. */
int zkw_blocks_gather_closed_form_inputs(zkw_block *const *blocks, size_t n_blocks, zkw_comm *comm, int rank, int world, int root,
                                         size_t max_per_block, uint64_t *out);
/* message of the last failed zkw_block_run on this thread (its builders run on worker threads, whose zkw_last_error
   is not the caller's) */
const char *zkw_block_last_error(void);
void zkw_block_free(zkw_block *b);
/* n_blocks blocks released on a few threads of the library (NULL entries are skipped); what zkw_block_free does for each. */
void zkw_blocks_free(zkw_block *const *blocks, size_t n_blocks);
/* witness of one circuit type, to be cast to its zkw_*_witness type (2 zkw_decommit_witness, 3 zkw_decommitter_witness,
   4 zkw_demux_witness, 5/6/7 zkw_precompile_witness, 8 zkw_ram_witness, 9 zkw_storage_witness, 10
   zkw_storage_application_witness, 11/12 zkw_events_witness); NULL for the others. The handles belong to the block and
   to the context zkw_block_context() returns for the type (device pointer mode). */
void *zkw_block_witness(const zkw_block *b, uint8_t circuit_type);
zkw_ctx *zkw_block_context(const zkw_block *b, uint8_t circuit_type);
size_t zkw_block_num_instances(const zkw_block *b, uint8_t circuit_type);
/* public inputs [n_instances][4] of every type but MainVM (1; its closed form needs the VM's local state; 10 only when the
   storage tree answers were given), the RecursionRequest encodings
   [n_instances][8] and the RecursionQueueSimulator states [n_instances][12] after each push (postprocessing/mod.rs:393-400).
   Host pointers valid until zkw_block_free; NULL when the type has none. */
const uint64_t *zkw_block_public_inputs(const zkw_block *b, uint8_t circuit_type);
const uint64_t *zkw_block_recursion_encodings(const zkw_block *b, uint8_t circuit_type);
const uint64_t *zkw_block_recursion_states(const zkw_block *b, uint8_t circuit_type);
/* MainVM instance records (zkw_vm_instance[zkw_block_num_instances(b, 1)], host) when vm_tracer was given, else NULL */
const zkw_vm_instance *zkw_block_vm_instances(const zkw_block *b);
/* the whole memory queue the RAM permutation saw, its final state, the six demuxed queue offsets, the L1 messages
   pubdata hash (compute_linear_keccak256) */
size_t zkw_block_memory_queue_length(const zkw_block *b);
const zkw_mem_query *zkw_block_memory_queue_device_ptr(const zkw_block *b);
int zkw_block_memory_queue_state(const zkw_block *b, zkw_queue_state12 *out);
int zkw_block_demuxed_offsets(const zkw_block *b, uint64_t out[7]);
int zkw_block_l1_messages_hash(const zkw_block *b, uint8_t out[32]);
/* the LinearHasher instance (type 13) over the net L2 -> L1 messages queue: its closed form is among zkw_block_public_inputs */
int zkw_block_linear_hasher_instance(const zkw_block *b, zkw_linear_hasher_instance *out);
/* wall-clock spans of the last run: names (comma separated), then start / end in ms since zkw_block_run was entered */
int zkw_block_timings(const zkw_block *b, char *names, size_t names_bytes, double *start_ms, double *end_ms, size_t max_spans,
                      size_t *n_spans);
/* ZkSyncBaseLayerCircuit::synthesis for every instance of the synthesizable types of the block, in the reference's
   emission order (oracle.rs:975-984 demuxer, 1039-1049 RAM, then CircuitMaker order 1494-1732: decommit sorter, keccak256
   round function, storage sorter, events, L1 messages, L1-messages hasher): each instance is filled into a slot of an
   internal trace ring (n_rows rows, 151 columns — a type with fewer columns uses the first of them, the rest of the slot
   is unspecified; `ring_slots` slots) and handed to `cb` — the counterpart of external_calls::run's circuit_callback
   (a prover consumes the slot before it is reused; the slot stays valid until cb returns). cb may be NULL.
   n_rows must hold every type at the block's capacities (the keccak netlist circuits need at least 65 536 rows).
   *n_done = number of instances synthesized. */
typedef int (*zkw_circuit_fn)(void *user, uint8_t circuit_type, size_t instance, const zkw_trace *trace, size_t slot,
                              const uint64_t public_input[4]);
int zkw_block_synthesize(zkw_block *b, size_t n_rows, size_t ring_slots, zkw_circuit_fn cb, void *user, size_t *n_done);
/* The same on one rank of a multi-GPU job: every rank has run zkw_block_run on the same inputs (the builders are
   deterministic and bounded by one serial hash chain: replicating them keeps every GPU's witnesses local) and synthesizes
   only the instances zkw_shard_lpt gives it (the plan covers the block's synthesizable instances in emission order). */
int zkw_block_synthesize_sharded(zkw_block *b, size_t n_rows, size_t ring_slots, int rank, int world, zkw_circuit_fn cb,
                                 void *user, size_t *n_done);
/* K blocks at once (after zkw_blocks_run): every synthesizable instance of every block, each trace cell for cell what zkw_block_synthesize
   hands out for the same (block, type, instance). How it differs from K calls of zkw_block_synthesize: (1) the ECRecover instances of ALL blocks
   are synthesized in joint calls (zkw_ecrecover_synthesize_multi: at most ec_chunk instances each, 0 = 16, at most 64) on two threads of the
   library with rings and priority streams of their own — a request's accumulator chain costs ~2.5 ms per call whatever the batch (round 5: 13 ms); (2) the other
   types run on a few workers (ZKW_SYNTH_THREADS, default 3), each with ONE ring of 16 x ring_slots slots (a ring belongs to a worker, not to a
   block: 1.28 GB a slot) and a contiguous share of the blocks: 16 fibers of the worker's thread own a slot each and go through their blocks TYPE BY
   TYPE, in step, so that a type's fills leave as one launch per kernel over 16 instances and a slot keeps its layout from call to call
   (csrc/zkw_batch.h). cb (may be NULL) is called from the workers' threads, possibly concurrently (one call at a time per worker), with the
   block's index; the slot is the callee's until it returns; per block the types arrive in the reference's emission order except that ECRecover
   instances arrive on their own, and between blocks the order is by type first. The callee must NOT use the blocks' contexts
   (zkw_block_context): they are at work on the blocks' other instances — a block's ECRecover instances arrive from another thread than its
   Keccak / SHA-256 / decommitter instances, which share the block's precompile context. A check (zkw_check_satisfied) or a read
   (zkw_trace_get) inside cb takes a context of the calling thread's own. */
typedef int (*zkw_blocks_circuit_fn)(void *user, size_t block, uint8_t circuit_type, size_t instance, const zkw_trace *trace, size_t slot,
                                     const uint64_t public_input[4]);
int zkw_blocks_synthesize(zkw_block *const *blocks, size_t n_blocks, size_t n_rows, size_t ring_slots, size_t ec_chunk,
                          zkw_blocks_circuit_fn cb, void *user, size_t *n_done);
/* The one collective of the multi-GPU path: per owned instance the record [circuit_type, instance, compact closed-form
   input (18), public input (4)] (24 words) is gathered to `root` over `comm` (RCCL), which receives them in emission
   order in out[n_records][24] (host) for the recursion-queue replay (postprocessing/mod.rs:396-402). Collective. */
int zkw_block_gather_closed_form_inputs(zkw_block *b, zkw_comm *comm, int rank, int world, int root, uint64_t *out,
                                        size_t max_records, size_t *n_records);

#ifdef __cplusplus
}
#endif
#endif /* ZKW_H */
