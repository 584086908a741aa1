/* zkw_types.h — plain-data records that cross the C ABI of libzkw (MI355X witness/synthesis engine).
 *
 * Every record mirrors a Rust type of the reference (paths relative to the reference root):
 *   zkw_mem_query      <- zk_evm::aux_structures::MemoryQuery as consumed by
 *                         circuit_encodings/src/memory_query.rs:24-118 (`encoding_witness`)
 *                         and reflected at memory_query.rs:131-144 (`reflect`).
 *   zkw_queue_state12  <- QueueStateWitness<F, FULL_SPONGE_QUEUE_STATE_WIDTH = 12>
 *                         built by src/witness/utils.rs:73-85 (`transform_sponge_like_queue_state`).
 *   zkw_ram_fsm        <- RamPermutationFSMInputOutputWitness, fields as filled at
 *                         src/witness/individual_circuits/ram_permutation.rs:385-406.
 *   zkw_ram_instance   <- RamPermutationCircuitInstanceWitness / ClosedFormInputWitness,
 *                         ram_permutation.rs:372-412.
 *
 * All field elements are Goldilocks (p = 2^64 - 2^32 + 1) stored as canonical (< p) little-endian
 * uint64_t. No pointers inside records; bulk arrays are passed next to them.
 */
#ifndef ZKW_TYPES_H
#define ZKW_TYPES_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKW_GOLDILOCKS_P 0xFFFFFFFF00000001ULL

#define ZKW_MEMORY_QUERY_PACKED_WIDTH 8   /* memory_query.rs:116 */
#define ZKW_LOG_QUERY_PACKED_WIDTH 20     /* log_query.rs:391-394 */
#define ZKW_DECOMMIT_QUERY_PACKED_WIDTH 8 /* decommittment_request.rs:72 */
#define ZKW_FULL_SPONGE_QUEUE_STATE_WIDTH 12
#define ZKW_QUEUE_STATE_WIDTH 4
#define ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS 2 /* ram_permutation.rs:85 */
#define ZKW_RAM_SORTING_KEY_LENGTH 3               /* memory_query.rs:6-14 */
#define ZKW_RAM_FULL_KEY_LENGTH 2                  /* memory_query.rs:16-20 */
/* zkevm_opcode_defs::BOOTLOADER_HEAP_PAGE (used at ram_permutation.rs:315). The crate is absent from
   the reference tree; value = heap_page_from_base(BOOTLOADER_BASE_PAGE = 8) = base + 2 in
   zkevm_opcode_defs v1.4.1 (inferred, see DESIGN.md "inferred constants"). */
#define ZKW_BOOTLOADER_HEAP_PAGE 10u

/* One memory access. 48 bytes, 16-byte aligned so a lane moves it as three 128-bit words. */
typedef struct zkw_mem_query {
    uint32_t timestamp;        /* query.timestamp.0 */
    uint32_t page;             /* query.location.page.0 */
    uint32_t index;            /* query.location.index.0 */
    uint8_t rw_flag;           /* 1 = write */
    uint8_t value_is_pointer;  /* 1 = fat pointer */
    uint8_t _pad[2];
    uint32_t value[8];         /* U256 as 8 little-endian u32 limbs (decompose_u256_as_u32x8) */
} zkw_mem_query;

/* One storage / event / L1-message / precompile log (zk_evm::aux_structures::LogQuery as consumed by
   circuit_encodings/src/log_query.rs:102-396). 128 bytes. 256-bit and 160-bit fields are little-endian
   u32 limbs (decompose_u256_as_u32x8 / decompose_address_as_u32x5). */
typedef struct zkw_log_query {
    uint32_t timestamp;
    uint16_t tx_number_in_block; /* u16 out of circuit, u32 in circuit (log_query.rs:367) */
    uint8_t aux_byte;
    uint8_t shard_id;
    uint32_t address[5];
    uint32_t key[8];
    uint32_t read_value[8];
    uint32_t written_value[8];
    uint8_t rw_flag;
    uint8_t rollback;
    uint8_t is_service;
    uint8_t _pad;
} zkw_log_query;

/* zk_evm::aux_structures::DecommittmentQuery as consumed by
   circuit_encodings/src/decommittment_request.rs:9-74. 48 bytes. */
typedef struct zkw_decommit_query {
    uint32_t hash[8]; /* U256, little-endian limbs */
    uint32_t timestamp;
    uint32_t memory_page;
    uint16_t decommitted_length;
    uint8_t is_fresh;
    uint8_t _pad[5];
} zkw_decommit_query;

/* QueueStateWitness<F, QUEUE_STATE_WIDTH = 4> (src/witness/utils.rs:59-71) */
typedef struct zkw_queue_state4 {
    uint64_t head[4];
    uint64_t tail[4];
    uint32_t length;
    uint32_t _pad;
} zkw_queue_state4;

/* zkevm_circuits::storage_validity_by_grand_product::EXTENDED_TIMESTAMP_ENCODING_{ELEMENT,OFFSET}
   (log_query.rs:400-427); the crate is absent, values inferred (DESIGN.md "inferred constants") */
#define ZKW_EXTENDED_TIMESTAMP_ENCODING_ELEMENT 19
#define ZKW_EXTENDED_TIMESTAMP_ENCODING_OFFSET 8

typedef struct zkw_queue_state12 {
    uint64_t head[12];
    uint64_t tail[12];
    uint32_t length;
    uint32_t _pad;
} zkw_queue_state12;

/* ---- recursion layer (src/witness/recursive_aggregation.rs) */
#define ZKW_NUM_BASE_LAYER_CIRCUITS 13
/* RecursionLeafParametersWitness (recursive_aggregation.rs:208-213) */
typedef struct zkw_leaf_params {
    uint64_t circuit_type;
    uint64_t basic_circuit_vk_commitment[4];
    uint64_t leaf_layer_vk_commitment[4];
} zkw_leaf_params;
/* QueueTailStateWitness: a node's split point (recursive_aggregation.rs:352-367) */
typedef struct zkw_queue_tail12 {
    uint64_t tail[12];
    uint32_t length;
    uint32_t _pad;
} zkw_queue_tail12;

typedef struct zkw_ram_fsm {
    uint64_t lhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    uint64_t rhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    zkw_queue_state12 current_unsorted_queue_state;
    zkw_queue_state12 current_sorted_queue_state;
    uint32_t previous_sorting_key[ZKW_RAM_SORTING_KEY_LENGTH]; /* [timestamp, index, page] */
    uint32_t previous_full_key[ZKW_RAM_FULL_KEY_LENGTH];       /* [index, page] */
    uint32_t previous_value[8];
    uint32_t previous_is_ptr;
    uint32_t num_nondeterministic_writes;
    uint32_t _pad;
} zkw_ram_fsm;

typedef struct zkw_ram_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    /* observable_input (RamPermutationInputDataWitness) */
    zkw_queue_state12 unsorted_queue_initial_state;
    zkw_queue_state12 sorted_queue_initial_state;
    uint32_t non_deterministic_bootloader_memory_snapshot_length;
    uint32_t _pad;
    zkw_ram_fsm hidden_fsm_input;
    zkw_ram_fsm hidden_fsm_output;
    /* the instance's slice of the two queue witnesses: items [first_item, first_item+num_items) of
       the block-wide unsorted / sorted arrays (queries, encodings, tails) */
    uint64_t first_item;
    uint64_t num_items;
} zkw_ram_instance;

/* ---- callstack (a3 / a6) -------------------------------------------------------------------------- */
#define ZKW_EXECUTION_CONTEXT_RECORD_ENCODING_WIDTH 32 /* callstack_entry.rs:33-36,179 */

/* ExtendedCallstackEntry (circuit_encodings/src/callstack_entry.rs:16-24) around zk_evm's CallStackEntry,
   only the fields encoding_witness (:36-179) reads. Addresses are u32x5 little-endian limbs
   (decompose_address_as_u32x5), context_u128_value is u32x4 (u128_as_u32_le, :6-13). is_kernel_mode is
   derived by the reference from this_address (CallStackEntry::is_kernel_mode in the absent zk_evm crate:
   the upper 18 bytes of the address are zero); the library derives it the same way. 176 bytes. */
typedef struct zkw_callstack_entry {
    uint64_t rollback_queue_head[4];
    uint64_t rollback_queue_tail[4];
    uint32_t rollback_queue_segment_length;
    uint32_t code_address[5];
    uint32_t this_address[5];
    uint32_t msg_sender[5];
    uint32_t context_u128_value[4];
    uint32_t code_page;
    uint32_t base_memory_page;
    uint32_t ergs_remaining;
    uint32_t heap_bound;
    uint32_t aux_heap_bound;
    uint16_t pc;
    uint16_t sp;
    uint16_t exception_handler_location;
    uint8_t this_shard_id;
    uint8_t caller_shard_id;
    uint8_t code_shard_id;
    uint8_t is_static;
    uint8_t is_local_frame;
    uint8_t _pad[1];
} zkw_callstack_entry;

/* ---- CodeDecommittmentsSorter (sort_decommit_requests.rs) ---------------------------------------- */
#define ZKW_DECOMMIT_PACKED_KEY_LENGTH 9 /* [timestamp, hash limbs 0..7], sort_decommit_requests.rs:422-435 */

/* CodeDecommittmentsDeduplicatorFSMInputOutputWitness, fields as filled at
   src/witness/individual_circuits/sort_decommit_requests.rs:402-414 */
typedef struct zkw_decommit_sorter_fsm {
    zkw_queue_state12 initial_queue_state; /* unsorted queue, remaining part */
    zkw_queue_state12 sorted_queue_state;
    zkw_queue_state12 final_queue_state;   /* deduplicated (output) queue */
    uint64_t lhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    uint64_t rhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    uint32_t previous_packed_key[ZKW_DECOMMIT_PACKED_KEY_LENGTH];
    uint32_t first_encountered_timestamp;
    uint32_t _pad[2];
    zkw_decommit_query previous_record; /* DecommitQueryWitness{code_hash, page, is_first, timestamp} */
} zkw_decommit_sorter_fsm;

/* CodeDecommittmentsDeduplicatorInstanceWitness, sort_decommit_requests.rs:345-420 */
typedef struct zkw_decommit_sorter_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    zkw_queue_state12 initial_queue_state;        /* observable_input */
    zkw_queue_state12 sorted_queue_initial_state; /* observable_input */
    zkw_queue_state12 final_queue_state;          /* observable_output (placeholder except on the last) */
    zkw_decommit_sorter_fsm hidden_fsm_input;
    zkw_decommit_sorter_fsm hidden_fsm_output;
    uint64_t first_item;
    uint64_t num_items;
} zkw_decommit_sorter_instance;

/* ---- Events / L1-messages sorter (events_sort_dedup.rs) ----------------------------------------- */
/* EventsDeduplicatorFSMInputOutputWitness, src/witness/individual_circuits/events_sort_dedup.rs:438-455 */
typedef struct zkw_events_sorter_fsm {
    uint64_t lhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    uint64_t rhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    zkw_queue_state4 initial_unsorted_queue_state;
    zkw_queue_state4 intermediate_sorted_queue_state;
    zkw_queue_state4 final_result_queue_state;
    uint32_t previous_key; /* timestamp */
    uint32_t _pad;
    zkw_log_query previous_item;
} zkw_events_sorter_fsm;

/* EventsDeduplicatorInstanceWitness, events_sort_dedup.rs:426-461 */
typedef struct zkw_events_sorter_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    zkw_queue_state4 initial_log_queue_state;         /* observable_input */
    zkw_queue_state4 intermediate_sorted_queue_state; /* observable_input */
    zkw_queue_state4 final_queue_state;               /* observable_output (placeholder except on the last) */
    zkw_events_sorter_fsm hidden_fsm_input;
    zkw_events_sorter_fsm hidden_fsm_output;
    uint64_t first_item;
    uint64_t num_items;
} zkw_events_sorter_instance;

/* ---- LogDemuxer (log_demux.rs) --------------------------------------------------------------------- */
#define ZKW_DEMUX_NUM_QUEUES 6
enum { ZKW_DEMUX_STORAGE = 0, ZKW_DEMUX_EVENTS = 1, ZKW_DEMUX_L1_MESSAGES = 2, ZKW_DEMUX_KECCAK256 = 3,
       ZKW_DEMUX_SHA256 = 4, ZKW_DEMUX_ECRECOVER = 5 };
/* Routing constants of zkevm_opcode_defs::system_params used at src/witness/individual_circuits/
   log_demux.rs:154-161 (crate absent: values inferred, DESIGN.md "inferred constants"). */
typedef struct zkw_demux_params {
    uint8_t storage_aux_byte;     /* STORAGE_AUX_BYTE = 0 */
    uint8_t event_aux_byte;       /* EVENT_AUX_BYTE = 1 */
    uint8_t l1_message_aux_byte;  /* L1_MESSAGE_AUX_BYTE = 2 */
    uint8_t precompile_aux_byte;  /* PRECOMPILE_AUX_BYTE = 3 */
    uint32_t keccak256_address;   /* KECCAK256_ROUND_FUNCTION_PRECOMPILE_ADDRESS = 0x8010 (as the low limb of H160) */
    uint32_t sha256_address;      /* SHA256_ROUND_FUNCTION_PRECOMPILE_ADDRESS = 0x02 */
    uint32_t ecrecover_address;   /* ECRECOVER_INNER_FUNCTION_PRECOMPILE_ADDRESS = 0x01 */
} zkw_demux_params;
#define ZKW_DEMUX_PARAMS_DEFAULT {0, 1, 2, 3, 0x8010u, 0x02u, 0x01u}

/* LogDemuxerFSMInputOutput, log_demux.rs:283-301 */
typedef struct zkw_log_demux_fsm {
    zkw_queue_state4 initial_log_queue_state;
    zkw_queue_state4 queue_state[ZKW_DEMUX_NUM_QUEUES]; /* storage, events, l1messages, keccak256, sha256, ecrecover */
} zkw_log_demux_fsm;

/* LogDemuxerCircuitInstanceWitness, log_demux.rs:303-375 */
typedef struct zkw_log_demux_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    zkw_queue_state4 initial_log_queue_state;                 /* observable_input */
    zkw_queue_state4 output_queue_state[ZKW_DEMUX_NUM_QUEUES]; /* observable_output (placeholder except on the last) */
    zkw_log_demux_fsm hidden_fsm_input;
    zkw_log_demux_fsm hidden_fsm_output;
    uint64_t first_item;
    uint64_t num_items;
} zkw_log_demux_instance;

/* ---- StorageSorter (storage_sort_dedup.rs + sort_storage_access.rs) -------------------------------- */
#define ZKW_STORAGE_PACKED_KEY_LENGTH 13 /* key limbs 0..7 then address limbs 0..4, log_query.rs:82-92 */

/* StorageDeduplicatorFSMInputOutputWitness, src/witness/individual_circuits/storage_sort_dedup.rs:563-596 */
typedef struct zkw_storage_sorter_fsm {
    uint64_t lhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    uint64_t rhs_accumulator[ZKW_NUM_PERMUTATION_ARGUMENT_REPETITIONS];
    zkw_queue_state4 current_unsorted_queue_state;
    zkw_queue_state4 current_intermediate_sorted_queue_state;
    zkw_queue_state4 current_final_sorted_queue_state;
    uint32_t cycle_idx;
    uint32_t previous_packed_key[ZKW_STORAGE_PACKED_KEY_LENGTH];
    uint32_t previous_key[8];
    uint32_t previous_address[5];
    uint32_t previous_timestamp;
    uint32_t this_cell_has_explicit_read_and_rollback_depth_zero;
    uint32_t this_cell_base_value[8];
    uint32_t this_cell_current_value[8];
    uint32_t this_cell_current_depth;
    uint32_t _pad[2];
} zkw_storage_sorter_fsm;

/* StorageDeduplicatorInstanceWitness, storage_sort_dedup.rs:552-603 */
typedef struct zkw_storage_sorter_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    uint32_t shard_id_to_process;                    /* observable_input */
    uint32_t _pad;
    zkw_queue_state4 unsorted_log_queue_state;        /* observable_input */
    zkw_queue_state4 intermediate_sorted_queue_state; /* observable_input */
    zkw_queue_state4 final_sorted_queue_state;        /* observable_output (placeholder except on the last) */
    zkw_storage_sorter_fsm hidden_fsm_input;
    zkw_storage_sorter_fsm hidden_fsm_output;
    uint64_t first_item;
    uint64_t num_items;
} zkw_storage_sorter_instance;

/* ---- CodeDecommitter (decommit_code.rs) ------------------------------------------------------------- */
/* CodeDecommitterFSMInputOutput incl. CodeDecommittmentFSM, src/witness/individual_circuits/decommit_code.rs:
   172-199 (input), 363-401 (output), internal fields set at :262-277 */
typedef struct zkw_decommitter_fsm {
    zkw_queue_state12 decommittment_requests_queue_state;
    zkw_queue_state12 memory_queue_state;
    uint32_t sha256_inner_state[8];
    uint32_t hash_to_compare_against[8]; /* U256 LE limbs; the 4 most significant bytes zeroed */
    uint32_t current_index;
    uint32_t current_page;
    uint32_t timestamp;
    uint32_t num_rounds_left; /* u16 in the reference */
    uint32_t length_in_bits;
    uint8_t state_get_from_queue;
    uint8_t state_decommit;
    uint8_t finished;
    uint8_t _pad;
} zkw_decommitter_fsm;

/* CodeDecommitterCircuitInstanceWitness, decommit_code.rs:139-420 */
typedef struct zkw_decommitter_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    zkw_queue_state12 sorted_requests_queue_initial_state; /* observable_input (first instance only) */
    zkw_queue_state12 memory_queue_initial_state;          /* observable_input (first instance only) */
    zkw_queue_state12 memory_queue_final_state;            /* observable_output (last instance only) */
    zkw_decommitter_fsm hidden_fsm_input;
    zkw_decommitter_fsm hidden_fsm_output;
    uint64_t first_round, num_rounds;     /* SHA-256 rounds (cycles) of this instance in the flattened sequence */
    uint64_t first_request, num_requests; /* requests popped in this instance = sorted_requests_queue_witness */
    uint64_t first_word, num_words;       /* code words consumed = code_words (flattened) */
} zkw_decommitter_instance;

/* ---- precompile round-function circuits: keccak256 / sha256 / ecrecover (a16) ------------------------ */
/* zk_evm_abstractions::precompiles::keccak256::{KECCAK_RATE_BYTES, MEMORY_READS_PER_CYCLE,
   KECCAK_PRECOMPILE_BUFFER_SIZE} (used at src/witness/individual_circuits/keccak256_round_function.rs:214-221).
   The crate is absent from the reference tree; values as in zk_evm_abstractions v1.4.1 (inferred: a round must
   find 136 bytes after at most MEMORY_READS_PER_CYCLE unaligned 32-byte reads, which needs 6 reads; the buffer
   holds MEMORY_READS_PER_CYCLE words). */
#define ZKW_KECCAK_RATE_BYTES 136
#define ZKW_KECCAK_MEMORY_READS_PER_CYCLE 6
#define ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE 192

enum { ZKW_PRECOMPILE_KECCAK256 = 0, ZKW_PRECOMPILE_SHA256 = 1, ZKW_PRECOMPILE_ECRECOVER = 2 };

/* {Keccak256,Sha256}RoundFunctionFSMInputOutputWitness / EcrecoverCircuitFSMInputOutputWitness: the two queue
   states + the internal FSM (keccak256_round_function.rs:420-441, sha256_round_function.rs:302-316; ecrecover
   has no internal part, ecrecover.rs:226-233). Fields a circuit does not have stay zero. */
typedef struct zkw_precompile_fsm {
    zkw_queue_state4 log_queue_state;
    zkw_queue_state12 memory_queue_state;
    uint8_t read_precompile_call;
    uint8_t read_words_for_round; /* keccak: read_unaligned_words_for_round */
    uint8_t padding_round;        /* keccak only */
    uint8_t completed;
    uint32_t timestamp_to_use_for_read;
    uint32_t timestamp_to_use_for_write;
    /* precompile_call_params */
    uint32_t input_page;
    uint32_t input_offset;  /* sha256: word index of the next read; keccak: input_memory_byte_offset */
    uint32_t input_length;  /* keccak: input_memory_byte_length (bytes left) */
    uint32_t output_page;
    uint32_t output_offset;
    uint32_t num_rounds;    /* sha256: rounds left */
    uint32_t needs_full_padding_round; /* keccak only */
    uint32_t buffer_filled;            /* keccak only */
    uint32_t sha256_inner_state[8];
    uint8_t keccak_internal_state[200]; /* [5][5][8]: [x][y] = lane (x, y) little-endian, keccak256_round_function.rs:530-541 */
    uint8_t buffer_bytes[ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE];
    uint32_t _pad;
} zkw_precompile_fsm;

/* One Keccak-f[1600] call of the keccak256 precompile (= one cycle of the Keccak256RoundFunction circuit, type 5), in the
   global round order of the block: the padded 136-byte block absorbed by the call, whether it starts a new request (the
   sponge state is reset to zero first) and the sponge state after the call ([x + 5y] lanes, little-endian bytes).
   ZKW_PRC_KECCAK_ROUNDS of the keccak256 witness; consumed by zkw_keccak_round_synthesize. */
typedef struct zkw_keccak_round_record {
    uint8_t block[136];
    uint8_t reset;
    uint8_t _pad[7];
    uint8_t state_after[200];
} zkw_keccak_round_record;

/* One SHA-256 compression of the sha256 precompile (= one cycle of the Sha256RoundFunction circuit, type 6), in the global
   round order of the block: the 64-byte message block (two memory words, big-endian as hashed), whether it starts a new
   request (the chaining state is reset to the IV first) and the chaining state after the compression.
   ZKW_PRC_SHA256_ROUNDS of the sha256 witness; consumed by zkw_sha256_round_synthesize. */
typedef struct zkw_sha256_round_record {
    uint8_t block[64];
    uint32_t reset;
    uint32_t state_after[8];
    uint32_t _pad;
} zkw_sha256_round_record;

/* {Keccak256RoundFunction,Sha256RoundFunction,Ecrecover}CircuitInstanceWitness */
typedef struct zkw_precompile_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    zkw_queue_state4 initial_log_queue_state;     /* observable_input, first instance only (else placeholder) */
    zkw_queue_state12 initial_memory_queue_state; /* observable_input, first instance only */
    zkw_queue_state12 final_memory_state;         /* observable_output, last instance only */
    zkw_precompile_fsm hidden_fsm_input;
    zkw_precompile_fsm hidden_fsm_output;
    uint64_t first_request, num_requests; /* requests_queue_witness: items of the demuxed request queue */
    uint64_t first_read, num_reads;       /* memory_reads_witness: values of the READ queries, in queue order */
    uint64_t first_round, num_rounds;
} zkw_precompile_instance;

/* ---- StorageApplication (a17) ------------------------------------------------------------------------ */
#define ZKW_STORAGE_TREE_DEPTH 256 /* BinarySparseStorageTree<256, 32, 32, 8, 32, Blake2s256, ZkSyncStorageLeaf>, storage_application.rs:41 */
#define ZKW_STATE_DIFF_RECORD_BYTE_ENCODING_LEN 156 /* state_diff_record.rs:21-53: 20 + 32 + 32 + 8 + 32 + 32 */
/* zkevm_circuits::base_structures::state_diff_record::NUM_KECCAK256_ROUNDS_PER_RECORD_ACCUMULATION (absent crate;
   156 bytes need two 136-byte keccak blocks, storage_application.rs:253-260) */
#define ZKW_NUM_KECCAK256_ROUNDS_PER_RECORD_ACCUMULATION 2

/* StorageApplicationFSMInputOutput, fields as filled at storage_application.rs:293-299 */
typedef struct zkw_storage_application_fsm {
    uint32_t next_enumeration_counter[2]; /* u64_as_u32_le */
    uint8_t current_root_hash[32];
    zkw_queue_state4 current_storage_application_log_state;
    uint8_t current_diffs_keccak_accumulator_state[200]; /* [5][5][8], keccak256_round_function.rs:530-541 */
} zkw_storage_application_fsm;

/* StorageApplicationCircuitInstanceWitness, storage_application.rs:322-336 */
typedef struct zkw_storage_application_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    /* observable_input (StorageApplicationInputData), first instance only */
    uint32_t initial_next_enumeration_counter[2];
    uint8_t initial_root_hash[32];
    uint32_t shard;
    uint32_t _pad0;
    zkw_queue_state4 storage_application_log_state;
    /* observable_output (StorageApplicationOutputData), last instance only */
    uint32_t new_next_enumeration_counter[2];
    uint8_t new_root_hash[32];
    uint8_t state_diffs_keccak256_hash[32];
    zkw_storage_application_fsm hidden_fsm_input;
    zkw_storage_application_fsm hidden_fsm_output;
    /* the instance's slice of storage_queue_witness / merkle_paths / leaf_indexes_for_reads */
    uint64_t first_item;
    uint64_t num_items;
} zkw_storage_application_instance;

/* ---- LinearHasher (type 13): LinearHasherCircuitInstanceWitness, src/witness/individual_circuits/
   data_hasher_and_merklizer.rs:34-60: one instance per block, no FSM (start = completion = 1) ------------------------- */
typedef struct zkw_linear_hasher_instance {
    uint32_t start_flag;
    uint32_t completion_flag;
    zkw_queue_state4 queue_state; /* observable_input: the L1-messages result queue (:34-37) */
    uint8_t keccak256_hash[32];   /* observable_output (:38-49) */
} zkw_linear_hasher_instance;
/* cycles (Keccak-f calls) of the LinearHasher circuit for `capacity` messages of 88 bytes (L2_TO_L1_MESSAGE_BYTE_LENGTH,
   data_hasher_and_merklizer.rs:23): ceil of the padded length over the 136-byte rate; 774 messages -> 501 cycles */
#define ZKW_LINEAR_HASHER_CYCLES(capacity) ((uint32_t)((uint64_t)(capacity) * 88 / 136 + 1))

/* ---- MainVM instance slicing (a19): src/witness/oracle.rs:1229-1469, src/witness/utils.rs:428-496 -------------------- */
/* The eight cycle-stamped FIFOs the reference cuts into per-instance `VmWitnessOracle`s
   (circuit_definitions/src/aux_definitions/witness_oracle.rs:25-36): element k of stream s happened at VM cycle
   stream_cycles[s][k] (ascending). An instance covering cycles [from, to) owns the elements with from <= cycle < to. */
enum {
    ZKW_VMS_MEMORY = 0,                        /* memory_read_witness + memory_write_witness (split by rw_flag) */
    ZKW_VMS_STORAGE_QUERIES = 1,               /* storage_queries */
    ZKW_VMS_REFUNDS = 2,                       /* storage_refund_queries */
    ZKW_VMS_DECOMMIT_REQUESTS = 3,             /* decommittment_requests_witness */
    ZKW_VMS_ROLLBACK_TAILS_FOR_NEW_FRAMES = 4, /* rollback_queue_initial_tails_for_new_frames */
    ZKW_VMS_CALLSTACK_VALUES = 5,              /* callstack_values_witnesses */
    ZKW_VMS_ROLLBACK_HEAD_SEGMENTS = 6,        /* rollback_queue_head_segments */
    ZKW_VMS_NEW_FRAMES = 7,                    /* callstack_new_frames_witnesses (flat_new_frames_history) */
    ZKW_VM_NUM_STREAMS = 8
};

/* StorageLogDetailedState, src/witness/oracle.rs:87-94 (frame_idx is not consumed by the slicing) */
typedef struct zkw_storage_log_detailed_state {
    uint64_t forward_tail[4];
    uint64_t rollback_head[4];
    uint64_t rollback_tail[4];
    uint32_t forward_length;
    uint32_t rollback_length;
} zkw_storage_log_detailed_state;

/* ---- the tracer's raw record of one block (a19, pre-builder half): what WitnessTracer feeds CallstackWithAuxData
   (src/witness/tracer.rs:221-407 -> src/witness/callstack_handler.rs:174-460), in time order --------------------------- */
enum {
    ZKW_VME_LOG = 0,  /* add_log_query(cycle, log_queries[index])                                   callstack_handler.rs:348 */
    ZKW_VME_PUSH = 1, /* push_entry(cycle, entries[2 * index] = the caller as saved, entries[2 * index + 1] = the new frame) :174 */
    ZKW_VME_POP = 2   /* pop_entry(cycle, panicked)                                                                        :224 */
};
typedef struct zkw_vm_event {
    uint32_t kind;
    uint32_t cycle; /* monotonic_cycle_counter */
    uint32_t panicked;
    uint32_t index;
} zkw_vm_event;
typedef struct zkw_vm_trace_summary {
    uint64_t n_flat;                    /* forward ++ reverse(rollback) of the root frame */
    uint64_t original_log_queue_length; /* its applied prefix = original_log_queue_simulator (oracle.rs:322-330): what the demuxer gets */
    uint64_t n_frames;                  /* monotonic_frame_counter */
    uint64_t global_end_of_storage_log[4];
    uint64_t original_log_queue_tail[4];
} zkw_vm_trace_summary;

/* VmInCircuitAuxilaryParameters, src/witness/oracle.rs:98-108 (the CallStackEntry of callstack_state travels with the
   VmLocalState snapshot the instance points at) */
typedef struct zkw_vm_aux_parameters {
    uint64_t callstack_state[12];
    zkw_queue_state12 decommittment_queue_state;
    zkw_queue_state12 memory_queue_state;
    zkw_queue_state4 storage_log_queue_state;
    uint64_t current_frame_rollback_queue_tail[4];
    uint64_t current_frame_rollback_queue_head[4];
    uint32_t current_frame_rollback_queue_segment_length;
    uint32_t _pad;
} zkw_vm_aux_parameters;

/* VmInstanceWitness + the closed-form flags / observable parts vm_instance_witness_to_circuit_formal_input derives
   (src/witness/utils.rs:428-496). The VmWitnessOracle FIFOs are [lo, hi) ranges of the block-wide streams. */
typedef struct zkw_vm_instance {
    uint32_t start_flag;      /* is_first */
    uint32_t completion_flag; /* is_last */
    uint32_t cycle_from;      /* cycles_range */
    uint32_t cycle_to;
    uint32_t snapshot_initial; /* initial_state = vm_snapshots[snapshot_initial].local_state */
    uint32_t snapshot_final;   /* final_state   = vm_snapshots[snapshot_final].local_state */
    uint64_t range[ZKW_VM_NUM_STREAMS][2];
    /* the memory stream's elements of the instance split by rw_flag, order kept: entries [first, first + num) of the
       block-wide index arrays zkw_vm_slice_instances returns (indices into the memory stream) */
    uint64_t first_memory_read, num_memory_reads;
    uint64_t first_memory_write, num_memory_writes;
    zkw_vm_aux_parameters auxilary_initial_parameters;
    zkw_vm_aux_parameters auxilary_final_parameters; /* = the next instance's initial ones; global final states on the last */
    /* observable_input (first instance; zero elsewhere): VmInputData without per_block_context */
    uint64_t rollback_queue_tail_for_block[4];
    uint64_t memory_queue_initial_tail[12];
    uint32_t memory_queue_initial_length;
    uint32_t decommitment_queue_initial_length;
    uint64_t decommitment_queue_initial_tail[12];
    /* observable_output (last instance; zero elsewhere): VmOutputData */
    zkw_queue_state12 memory_queue_final_state;
    zkw_queue_state12 decommitment_queue_final_state;
    zkw_queue_state4 log_queue_final_state;
} zkw_vm_instance;

/* What the tracer leaves for the slicing (all arrays HOST or DEVICE per the context's pointer mode). */
typedef struct zkw_vm_tracer_streams {
    const uint32_t *snapshot_cycles;  /* vm_snapshots[k].at_cycle, ascending, n_snapshots >= 2 */
    size_t n_snapshots;
    const uint32_t *stream_cycles[ZKW_VM_NUM_STREAMS];
    size_t stream_len[ZKW_VM_NUM_STREAMS];
    const zkw_mem_query *vm_memory_queries; /* [stream_len[ZKW_VMS_MEMORY]]: the rw_flag splits reads from writes */
    const uint64_t *memory_queue_tails;     /* [stream_len[ZKW_VMS_MEMORY]][12]: all_memory_queue_states (tail after item k;
                                               its head is the tail after item k-1, num_items = k+1) */
    const uint32_t *decommit_state_cycles;  /* all_decommittment_queue_states: (cycle, state after push k) */
    const uint64_t *decommit_queue_tails;   /* [n_decommit_states][12] */
    size_t n_decommit_states;
    const uint32_t *callstack_sponge_cycles; /* callstack_sponge_encoding_ranges: (cycle, sponge state) */
    const uint64_t *callstack_sponge_states; /* [n_callstack_sponges][12] */
    size_t n_callstack_sponges;
    const uint32_t *storage_log_state_cycles; /* history_of_storage_log_states (BTreeMap: strictly ascending cycles) */
    const zkw_storage_log_detailed_state *storage_log_states;
    size_t n_storage_log_states;
    uint64_t global_end_of_storage_log[4];
} zkw_vm_tracer_streams;

#ifdef __cplusplus
}
#endif
#endif /* ZKW_TYPES_H */
