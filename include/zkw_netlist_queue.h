/* zkw_netlist_queue.h — the QUEUE SECTION of a "zkw trace v4" netlist circuit: the request-queue pops and memory-queue pushes of a
 * hash circuit as Poseidon2 rows inside its trace, tied to the hash netlist by copy constraints. What the reference's circuits do
 * with their `requests_queue.pop_front(..)` / `memory_queue.push(..)` gadgets (era-zkevm_circuits, absent; the out-of-circuit mirror
 * is src/witness/individual_circuits/sha256_round_function.rs:172-246 and decommit_code.rs:136-401), restated on the queue
 * arithmetic the reference does hold: FullWidthQueueSimulator::push (circuit_encodings/src/lib.rs:391-429: absorb the 8-element
 * encoding over the old tail, ONE permutation, the new tail is the whole state), QueueSimulator::push (lib.rs:180-203: hash of the
 * 20-element encoding ++ old tail from a zero state, THREE permutations, the new tail is elements 0..3), the encodings of
 * memory_query.rs:24-118 / log_query.rs:102-396 / decommittment_request.rs:9-74. Placement is this library's own (PARITY UNPINNED
 * at the placement level, like the rest of v4). Shared by csrc/netlist_queue_kernels.cuh, the host side and oracle/netlist_queue.c.
 *
 * Rows. The section starts below the PI row: q0 = NL_USED_ROWS(spec, capacity). Row q0 = QBND: for every queue its state before
 * cycle 0, then for every queue its state after the last cycle (general-purpose columns, back to back). Then the cycles'
 * operations, REGION-major: row r of a cycle's operations (r < q_rows_per_cycle) of cycle c is row q0 + 1 + r * capacity + c, so
 * that consecutive cycles are consecutive rows (coalesced stores of the lane-per-operation fill). Lookup columns of all section
 * rows are zero and not counted in the multiplicity column (like the boundary rows).
 *
 * One operation = an ENC block, then 1 or 3 P2 blocks; a block of n cells folds over ceil(n / G) rows, cell k at (row k / G,
 * column k % G). ENC block: [en | components | enc[8 or 20] | old[w] | new[w]], w = 12 or 4:
 *   en boolean (by rule: free, == the cycle's `reset`, or == 1 - `idle`); components = the item's fields (some LINKED = copies of
 *   netlist cells: the value nibbles of a memory word that is hashed / the digest that is written); enc_j = sum of component * 2^s
 *   (nlq_enc_terms); old = the queue state before (copy of the previous operation's `new` on that queue, across cycles; QBND for
 *   the first); new = old + en * (out - old).
 * P2 block: the 130 variables of the flattened Poseidon2 gate (12 inputs, states after the full rounds, S-box outputs of the
 * partial rounds — oracle/ram_circuit.c orc_poseidon2_flattened); inputs = enc / old by copy; out = its last 12 cells.
 *   PUSH12 / POP12: in = enc[0..8] ++ old[8..12]; out[0..12].     POP4: in_0 = enc[0..8] ++ 0000, in_1 = enc[8..16] ++ out_0[8..12],
 *   in_2 = enc[16..20] ++ old[0..4] ++ out_1[8..12]; out = out_2[0..4].
 * Relations between the operations of a cycle (nlq_rel below): rw flags, consecutive words of one page at one timestamp.
 * NOT constrained here (placed): what spans cycles (the word offset carried between rounds, rounds left, the call's ABI against the
 * first address), ranges of the unlinked components.
 */
#ifndef ZKW_NETLIST_QUEUE_H
#define ZKW_NETLIST_QUEUE_H
#include "zkw_netlist.h"
#include "zkw_types.h"

#define NLQ_P2_CELLS 130
#define NLQ_MAX_OPS 8
#define NLQ_MAX_QUEUES 2
#define NLQ_MAX_TERMS 14

enum { NLQ_PUSH12 = 1, NLQ_POP12 = 2, NLQ_POP4 = 3 }; /* PUSH12 and POP12 are the same arithmetic on a tail / on a head */
enum { NLQ_ITEM_MEM = 1, NLQ_ITEM_LOG = 2, NLQ_ITEM_DECOMMIT = 3, NLQ_ITEM_MEM8 = 4 /* a memory query with its value as 32 bytes (the byte-valued netlists) */,
       NLQ_ITEM_LOGB = 5 /* a log query that also holds written_value and tx_number as BYTES, tied to the limbs by recomposition gates (nlq_aux_*) */ };
enum { NLQ_EN_FREE = 0, NLQ_EN_RESET = 1, NLQ_EN_ACTIVE = 2 };
/* links of a memory query's 64 value nibbles (little end first) to the SHA-256 netlist:
   SHA_BLOCK + arg k: the cycle's message block, memory word k (U256::to_big_endian = block bytes 32k..32k+31; FREE element 2b + hi);
   SHA_DIGEST: the chaining state after the cycle (limb j of the written U256 = H[7 - j]) */
enum { NLQ_LINK_NONE = 0, NLQ_LINK_SHA_BLOCK = 1, NLQ_LINK_SHA_DIGEST = 2,
       NLQ_LINK_KECCAK_DIGEST = 3 /* MEM8: value byte x (little end first) = byte 31 - x of the sponge state after the cycle (the first four lanes, U256::from_big_endian) */,
       NLQ_LINK_LH_MESSAGE = 4 /* LOG: the byte-valued fields of a popped L2 -> L1 message are copies of the bytes the sponge absorbs: byte k of the
          88-byte serialisation (log_query.rs:503-534: shard | is_service | tx_number | address | key | written_value, big end first) of message
          m is byte 88 m + k of the hashed stream = FREE element (88 m + k) % 136 of cycle (88 m + k) / 136 — the cycle the message is popped in
          or the next one. All 88 bytes are linked: shard_id, is_service, the address and key bytes (bytes in the encoding anyway) and the
          tx_number / written_value bytes of NLQ_ITEM_LOGB (the encoding takes their limbs: nlq_aux_* recompose them). */,
       NLQ_LINK_EC_OK = 5 /* MEM8, ECRecover's first write (the success marker, src/witness/individual_circuits/ecrecover.rs:160-178): value byte 0 =
          state byte EK_STATE_OK after the cycle (`ok`, carried there by the netlist's select step), the other bytes = a state byte that is
          the constant zero */ };
#define NLQ_EK_STATE_OK 32   /* = EK_STATE_OK of include/zkw_ecrecover_circuit_spec.h */
#define NLQ_EK_STATE_ZERO 40 /* a state byte of the ECRecover netlist that is the constant 0 */

typedef struct nlq_op { uint8_t kind, item, queue, en_rule, link, link_arg;
                         uint8_t extra, reg_cell[2]; /* REGISTER cells after `new`: register r holds the limb (four byte cells from reg_cell[r]) of the
                            request the cycle is working on (the item operation 0 popped last) — FSM state carried by relations (nlq_rel) */ } nlq_op;
typedef struct nlq_desc { uint32_t n_ops, n_queues; uint32_t width[NLQ_MAX_QUEUES]; nlq_op ops[NLQ_MAX_OPS]; } nlq_desc;
typedef struct nlq_feed { uint32_t en; uint32_t idx; uint32_t aux; } nlq_feed; /* per (cycle, op): enabled?, index of the item (enabled) / of the queue's next item (disabled); aux: the value of the operation's registers beyond the first two (SHA-256: rounds left) */
typedef struct nlq_term { uint16_t cell; uint16_t shift; } nlq_term;

/* Sha256RoundFunction (6): pop the precompile call (first round of a request), read two words, write the digest (last round).
   CodeDecommitter (3): pop the decommit request (first round of a bytecode), write two code words (the second one is missing in the
   last round of a bytecode with an odd word count). */
static const nlq_desc NLQ_DESC_SHA256 = {4, 2, {4, 12}, {
    {NLQ_POP4, NLQ_ITEM_LOG, 0, NLQ_EN_RESET, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM, 1, NLQ_EN_ACTIVE, NLQ_LINK_SHA_BLOCK, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM, 1, NLQ_EN_ACTIVE, NLQ_LINK_SHA_BLOCK, 1, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM, 1, NLQ_EN_FREE, NLQ_LINK_SHA_DIGEST, 0, 3, {24 + 20, 24 + 8}}}}; /* registers: the ABI's page to write (key bytes 20..23), output offset (8..11); rounds left after this one (nlq_feed.aux) */
static const nlq_desc NLQ_DESC_CODE_DECOMMITTER = {3, 2, {12, 12}, {
    {NLQ_POP12, NLQ_ITEM_DECOMMIT, 0, NLQ_EN_RESET, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM, 1, NLQ_EN_ACTIVE, NLQ_LINK_SHA_BLOCK, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM, 1, NLQ_EN_FREE, NLQ_LINK_SHA_BLOCK, 1, 0, {0, 0}}}};

/* Keccak256RoundFunction (5): pop the precompile call (first round of a request), up to MEMORY_READS_PER_CYCLE = 6 reads into the byte
   buffer (keccak256_round_function.rs:232-290: unaligned, so which buffer bytes a word lands on is data — the reads are NOT linked to
   the block), write the digest after a request's last round (linked). */
static const nlq_desc NLQ_DESC_KECCAK256 = {8, 2, {4, 12}, {
    {NLQ_POP4, NLQ_ITEM_LOG, 0, NLQ_EN_RESET, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_FREE, NLQ_LINK_NONE, 0, 0, {0, 0}}, {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_FREE, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_FREE, NLQ_LINK_NONE, 0, 0, {0, 0}}, {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_FREE, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_FREE, NLQ_LINK_NONE, 0, 0, {0, 0}}, {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_FREE, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_FREE, NLQ_LINK_KECCAK_DIGEST, 0, 2, {24 + 20, 24 + 8}}}}; /* registers: the ABI's page to write, output offset */

/* ECRecover (7): every cycle is one request (ecrecover.rs:143-178): pop the call, read hash / v / r / s (four consecutive words), write
   the success marker and the address. The reads' value bytes are the inputs of the EC section (include/zkw_ecrecover.h): that copy
   constraint is stated and checked on the EC side (the input bytes' home cells), so the operations carry no link here; the writes are
   linked to the netlist's state after the cycle (the masked address digest[12..32] as the Keccak256RoundFunction links its digest, `ok`
   in state byte 32). */
static const nlq_desc NLQ_DESC_ECRECOVER = {7, 2, {4, 12}, {
    {NLQ_POP4, NLQ_ITEM_LOG, 0, NLQ_EN_ACTIVE, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_ACTIVE, NLQ_LINK_NONE, 0, 0, {0, 0}}, {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_ACTIVE, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_ACTIVE, NLQ_LINK_NONE, 0, 0, {0, 0}}, {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_ACTIVE, NLQ_LINK_NONE, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_ACTIVE, NLQ_LINK_EC_OK, 0, 0, {0, 0}},
    {NLQ_PUSH12, NLQ_ITEM_MEM8, 1, NLQ_EN_ACTIVE, NLQ_LINK_KECCAK_DIGEST, 0, 0, {0, 0}}}};

/* L1MessagesHasher (13): the circuit pops EVERY message of the queue and hashes its 88-byte serialisation (linear_hasher in the absent
   crate; out of circuit data_hasher_and_merklizer.rs:8-67). A message is popped in the cycle that absorbs its first byte — cycle
   floor(88 m / 136), at most two per cycle. Which block bytes a message lands on depends on the cycle (period 11): the links of its
   byte-valued fields (NLQ_LINK_LH_MESSAGE; link_arg = the slot 0 / 1 of the cycle) are a function of the cycle. One queue. */
static const nlq_desc NLQ_DESC_LINEAR_HASHER = {2, 1, {4, 0}, {
    {NLQ_POP4, NLQ_ITEM_LOGB, 0, NLQ_EN_FREE, NLQ_LINK_LH_MESSAGE, 0, 0, {0, 0}}, {NLQ_POP4, NLQ_ITEM_LOGB, 0, NLQ_EN_FREE, NLQ_LINK_LH_MESSAGE, 1, 0, {0, 0}}}};
/* messages whose first byte is absorbed by cycle c: [nlq_lh_first(c), nlq_lh_first(c + 1)) */
#define NLQ_LH_FIRST(c) (((uint64_t)(c) * 136 + 87) / 88)

#if defined(__HIPCC__)
#define NLQ_HD __host__ __device__ static inline
#else
#define NLQ_HD static inline
#endif

/* RELATIONS between the fields of a cycle's operations — the part of the circuits' FSM arithmetic that is visible inside one cycle:
   en(gate) * (cell_b of op_b - cell_a of op_a - add) = 0; op_a = NLQ_REL_CONST: en(gate) * (cell_b of op_b - add) = 0.
   prev = 1 (3: and NEGATED: b + a - add): cell_a is taken from the PREVIOUS cycle (no relation at cycle 0: an instance's first cycle continues from the FSM input,
   which is placed) and the factor is en(gate) - en(gate2): "this round reads and does not start a request" — the word offset carried
   from round to round. span = n > 1: the operand is the little-endian recomposition of n byte cells cell_a .. cell_a + n - 1 (a limb of
   the popped call's ABI: key bytes 0..3 = input offset, 16..19 = page to read — precompile_abi_in_log; a decommit request's page /
   timestamp bytes) — a request's FIRST address against the call. (Rounds left stay placed.) Memory-query cells: 1 timestamp, 2 page, 3 index, 4 rw, 5 value_is_pointer; log-query cell 17: timestamp. */
#define NLQ_REL_CONST 0xFF
#define NLQ_REL_ACTIVE 0xFE /* as a gate: the factor is 1 - idle of the cycle's netlist header (circuits without an always-on operation) */
typedef struct nlq_rel { uint8_t op_a, cell_a, op_b, cell_b, gate; int8_t add; uint8_t prev, gate2, span; } nlq_rel;
#define NLQ_MAX_RELS 40
/* Sha256RoundFunction: the reads are reads of consecutive words of one page at the call's timestamp, the write is a write one tick
   later (sha256_round_function.rs:204-246: timestamp_to_use_for_read / _write); nothing is a pointer */
#define NLQ_RELS_SHA256 { \
    {NLQ_REL_CONST, 0, 1, 4, 1, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 4, 2, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 3, 4, 3, 1, 0, 0xFF, 1}, \
    {NLQ_REL_CONST, 0, 1, 5, 1, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 5, 2, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 3, 5, 3, 0, 0, 0xFF, 1}, \
    {1, 3, 2, 3, 2, 1, 0, 0xFF, 1}, {1, 2, 2, 2, 2, 0, 0, 0xFF, 1}, {1, 1, 2, 1, 2, 0, 0, 0xFF, 1}, {1, 1, 3, 1, 3, 1, 0, 0xFF, 1}, {0, 17, 1, 1, 0, 0, 0, 0xFF, 1}, \
    {2, 3, 1, 3, 1, 1, 1, 0, 1}, {2, 2, 1, 2, 1, 0, 1, 0, 1}, {2, 1, 1, 1, 1, 0, 1, 0, 1}, \
    {0, 24, 1, 3, 0, 0, 0, 0xFF, 4}, {0, 40, 1, 2, 0, 0, 0, 0xFF, 4}, \
    {3, 0, 0, 0, 1, 0, 1, 0xFF, 1} /* a round that reads pops a call exactly when the round before wrote a digest: ties the write's free `en` to `reset` */, \
    {0, 44, 3, 102, 0, 0, 0, 0xFF, 4}, {0, 32, 3, 103, 0, 0, 0, 0xFF, 4} /* a pop loads the registers (cells 102, 103 of the write: page / offset to write) from the call's ABI */, \
    {3, 102, 3, 102, 1, 0, 1, 0, 1}, {3, 103, 3, 103, 1, 0, 1, 0, 1} /* a round that continues a request keeps them */, \
    {3, 102, 3, 2, 3, 0, 0, 0xFF, 1}, {3, 103, 3, 3, 3, 0, 0, 0xFF, 1} /* the digest is written where the call said */, \
    {0, 48, 3, 104, 0, -1, 0, 0xFF, 4}, {3, 104, 3, 104, 1, -1, 1, 0, 1}, {NLQ_REL_CONST, 0, 3, 104, 3, 0, 0, 0xFF, 1} /* rounds left (cell 104): a pop loads the ABI's round count (key bytes 24..27) - 1, a continuing round counts down, the digest is written at zero */}
/* CodeDecommitter: the code words are written (not pointers) to consecutive words of one page at one timestamp (decommit_code.rs:47-78) */
#define NLQ_RELS_CODE_DECOMMITTER { \
    {NLQ_REL_CONST, 0, 1, 4, 1, 1, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 4, 2, 1, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 1, 5, 1, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 5, 2, 0, 0, 0xFF, 1}, \
    {1, 3, 2, 3, 2, 1, 0, 0xFF, 1}, {1, 2, 2, 2, 2, 0, 0, 0xFF, 1}, {1, 1, 2, 1, 2, 0, 0, 0xFF, 1}, \
    {2, 3, 1, 3, 1, 1, 1, 0, 1}, {2, 2, 1, 2, 1, 0, 1, 0, 1}, {2, 1, 1, 1, 1, 0, 1, 0, 1}, \
    {0, 9, 1, 2, 0, 0, 0, 0xFF, 4}, {0, 13, 1, 1, 0, 0, 0, 0xFF, 4}, {NLQ_REL_CONST, 0, 1, 3, 0, 0, 0, 0xFF, 1}, \
    {2, 0, 0, 0, 1, 1, 3, 0xFF, 1} /* a round pops a request exactly when the round before wrote only one word (a bytecode's last round): en_pop + en_word1(prev) = 1 */}
/* Keccak256RoundFunction: up to six reads of consecutive words of one page at one timestamp (a round may read nothing, so neither the
   call's timestamp nor the write's is tied to a read's inside one cycle) */
#define NLQ_RELS_KECCAK256 { \
    {NLQ_REL_CONST, 0, 1, 4, 1, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 4, 2, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 3, 4, 3, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 4, 4, 4, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 5, 4, 5, 0, 0, 0xFF, 1}, \
    {NLQ_REL_CONST, 0, 6, 4, 6, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 7, 4, 7, 1, 0, 0xFF, 1}, \
    {NLQ_REL_CONST, 0, 1, 5, 1, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 5, 2, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 3, 5, 3, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 4, 5, 4, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 5, 5, 5, 0, 0, 0xFF, 1}, \
    {NLQ_REL_CONST, 0, 6, 5, 6, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 7, 5, 7, 0, 0, 0xFF, 1}, \
    {1, 3, 2, 3, 2, 1, 0, 0xFF, 1}, {2, 3, 3, 3, 3, 1, 0, 0xFF, 1}, {3, 3, 4, 3, 4, 1, 0, 0xFF, 1}, {4, 3, 5, 3, 5, 1, 0, 0xFF, 1}, {5, 3, 6, 3, 6, 1, 0, 0xFF, 1}, \
    {1, 2, 2, 2, 2, 0, 0, 0xFF, 1}, {2, 2, 3, 2, 3, 0, 0, 0xFF, 1}, {3, 2, 4, 2, 4, 0, 0, 0xFF, 1}, {4, 2, 5, 2, 5, 0, 0, 0xFF, 1}, {5, 2, 6, 2, 6, 0, 0, 0xFF, 1}, \
    {1, 1, 2, 1, 2, 0, 0, 0xFF, 1}, {2, 1, 3, 1, 3, 0, 0, 0xFF, 1}, {3, 1, 4, 1, 4, 0, 0, 0xFF, 1}, {4, 1, 5, 1, 5, 0, 0, 0xFF, 1}, {5, 1, 6, 1, 6, 0, 0, 0xFF, 1}, \
    {7, 0, 0, 0, NLQ_REL_ACTIVE, 0, 1, 0xFF, 1} /* an active round pops a call exactly when the round before wrote a digest */, \
    {0, 44, 7, 70, 0, 0, 0, 0xFF, 4}, {0, 32, 7, 71, 0, 0, 0, 0xFF, 4}, {7, 70, 7, 70, NLQ_REL_ACTIVE, 0, 1, 0, 1}, {7, 71, 7, 71, NLQ_REL_ACTIVE, 0, 1, 0, 1}, \
    {7, 70, 7, 2, 7, 0, 0, 0xFF, 1}, {7, 71, 7, 3, 7, 0, 0, 0xFF, 1} /* the registers (cells 70, 71 of the write): loaded by a pop, kept by a round that continues a request, the digest is written there */}
/* ECRecover: four reads of consecutive words of the call's page to read from its input offset at its timestamp, two writes of
   consecutive words of its page to write from its output offset one tick later (ecrecover.rs:143-178; zk_evm's precompile); all in ONE
   cycle, so the whole address arithmetic of a request is relations between the cycle's operations */
#define NLQ_RELS_ECRECOVER { \
    {NLQ_REL_CONST, 0, 1, 4, 1, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 4, 2, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 3, 4, 3, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 4, 4, 4, 0, 0, 0xFF, 1}, \
    {NLQ_REL_CONST, 0, 5, 4, 5, 1, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 6, 4, 6, 1, 0, 0xFF, 1}, \
    {NLQ_REL_CONST, 0, 1, 5, 1, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 2, 5, 2, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 3, 5, 3, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 4, 5, 4, 0, 0, 0xFF, 1}, \
    {NLQ_REL_CONST, 0, 5, 5, 5, 0, 0, 0xFF, 1}, {NLQ_REL_CONST, 0, 6, 5, 6, 0, 0, 0xFF, 1}, \
    {1, 3, 2, 3, 2, 1, 0, 0xFF, 1}, {2, 3, 3, 3, 3, 1, 0, 0xFF, 1}, {3, 3, 4, 3, 4, 1, 0, 0xFF, 1}, {5, 3, 6, 3, 6, 1, 0, 0xFF, 1}, \
    {1, 2, 2, 2, 2, 0, 0, 0xFF, 1}, {2, 2, 3, 2, 3, 0, 0, 0xFF, 1}, {3, 2, 4, 2, 4, 0, 0, 0xFF, 1}, {5, 2, 6, 2, 6, 0, 0, 0xFF, 1}, \
    {1, 1, 2, 1, 2, 0, 0, 0xFF, 1}, {2, 1, 3, 1, 3, 0, 0, 0xFF, 1}, {3, 1, 4, 1, 4, 0, 0, 0xFF, 1}, {5, 1, 6, 1, 6, 0, 0, 0xFF, 1}, \
    {1, 1, 5, 1, 5, 1, 0, 0xFF, 1}, {0, 17, 1, 1, 0, 0, 0, 0xFF, 1}, \
    {0, 24, 1, 3, 0, 0, 0, 0xFF, 4}, {0, 40, 1, 2, 0, 0, 0, 0xFF, 4}, {0, 32, 5, 3, 0, 0, 0, 0xFF, 4}, {0, 44, 5, 2, 0, 0, 0, 0xFF, 4}}
typedef struct nlq_rels { uint32_t n; nlq_rel r[NLQ_MAX_RELS]; } nlq_rels;
static const nlq_rels NLQ_RELS_OF_ECRECOVER = {30, NLQ_RELS_ECRECOVER};
static const nlq_rels NLQ_RELS_OF_SHA256 = {26, NLQ_RELS_SHA256};
static const nlq_rels NLQ_RELS_OF_CODE_DECOMMITTER = {14, NLQ_RELS_CODE_DECOMMITTER};
static const nlq_rels NLQ_RELS_OF_KECCAK256 = {36, NLQ_RELS_KECCAK256};
static const nlq_rels NLQ_RELS_NONE = {0, {{0, 0, 0, 0, 0, 0, 0, 0, 0}}};
static inline const nlq_rels *nlq_rels_of(int circuit_type) {
    return circuit_type == 6 ? &NLQ_RELS_OF_SHA256 : circuit_type == 3 ? &NLQ_RELS_OF_CODE_DECOMMITTER : circuit_type == 5 ? &NLQ_RELS_OF_KECCAK256 :
           circuit_type == 7 ? &NLQ_RELS_OF_ECRECOVER : &NLQ_RELS_NONE;
}

/* host only: kernels take the descriptor by value */
static inline const nlq_desc *nlq_desc_of(int circuit_type) {
    return circuit_type == 6 ? &NLQ_DESC_SHA256 : circuit_type == 3 ? &NLQ_DESC_CODE_DECOMMITTER : circuit_type == 5 ? &NLQ_DESC_KECCAK256 :
           circuit_type == 13 ? &NLQ_DESC_LINEAR_HASHER : circuit_type == 7 ? &NLQ_DESC_ECRECOVER : (const nlq_desc *)0;
}
NLQ_HD uint32_t nlq_kind_width(uint32_t kind) { return kind == NLQ_POP4 ? 4u : 12u; }
NLQ_HD uint32_t nlq_kind_perms(uint32_t kind) { return kind == NLQ_POP4 ? 3u : 1u; }
/* component cells of an item, `en` (cell 0) included; encoding elements */
NLQ_HD uint32_t nlq_item_comps(uint32_t item) { return item == NLQ_ITEM_MEM ? 70u : item == NLQ_ITEM_LOG ? 76u : item == NLQ_ITEM_LOGB ? 110u : item == NLQ_ITEM_MEM8 ? 38u : 18u; }
NLQ_HD uint32_t nlq_item_enc(uint32_t item) { return item == NLQ_ITEM_LOG || item == NLQ_ITEM_LOGB ? 20u : 8u; }
/* cells of the ENC block: [0, comps) | enc | old | new */
NLQ_HD uint32_t nlq_enc0(const nlq_op *op) { return nlq_item_comps(op->item); }
NLQ_HD uint32_t nlq_old0(const nlq_op *op) { return nlq_enc0(op) + nlq_item_enc(op->item); }
NLQ_HD uint32_t nlq_new0(const nlq_op *op) { return nlq_old0(op) + nlq_kind_width(op->kind); }
NLQ_HD uint32_t nlq_reg0(const nlq_op *op) { return nlq_new0(op) + nlq_kind_width(op->kind); }
NLQ_HD uint32_t nlq_enc_cells(const nlq_op *op) { return nlq_reg0(op) + op->extra; }
NLQ_HD uint32_t nlq_rows_for(uint32_t cells, uint32_t g) { return (cells + g - 1) / g; }
NLQ_HD uint32_t nlq_op_rows(const nlq_op *op, uint32_t g) { return nlq_rows_for(nlq_enc_cells(op), g) + nlq_kind_perms(op->kind) * nlq_rows_for(NLQ_P2_CELLS, g); }
/* first row (within a cycle's operations) of operation j / of its P2 block p */
NLQ_HD uint32_t nlq_op_row0(const nlq_desc *d, uint32_t g, uint32_t j) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < j; i++) r += nlq_op_rows(&d->ops[i], g);
    return r;
}
NLQ_HD uint32_t nlq_p2_row0(const nlq_desc *d, uint32_t g, uint32_t j, uint32_t p) {
    return nlq_op_row0(d, g, j) + nlq_rows_for(nlq_enc_cells(&d->ops[j]), g) + p * nlq_rows_for(NLQ_P2_CELLS, g);
}
NLQ_HD uint32_t nlq_rows_per_cycle(const nlq_desc *d, uint32_t g) { return nlq_op_row0(d, g, d->n_ops); }
#define NLQ_BASE(spec, capacity) NL_USED_ROWS(spec, capacity)
/* trace row of row r of the operations of cycle c */
#define NLQ_ROW(spec, capacity, r, c) (NLQ_BASE(spec, capacity) + 1 + (uint64_t)(r) * (capacity) + (c))
NLQ_HD uint64_t nlq_used_rows(const nl_spec *sp, const nlq_desc *d, uint32_t capacity) {
    return NL_USED_ROWS(sp, capacity) + (d ? 1 + (uint64_t)capacity * nlq_rows_per_cycle(d, sp->g) : 0);
}
/* the largest number of cycles whose netlist rows + queue section fit n_rows */
NLQ_HD uint32_t nlq_max_capacity(const nl_spec *sp, const nlq_desc *d, uint64_t n_rows) {
    const uint64_t fixed = 2 * NL_BND_ROWS(sp) + 1 + (d ? 1 : 0), per = sp->rows_per_cycle + (d ? nlq_rows_per_cycle(d, sp->g) : 0);
    return n_rows > fixed ? (uint32_t)((n_rows - fixed) / per) : 0;
}
/* QBND row: column of element k of queue q's state before cycle 0 (out == 0) / after the last cycle (out == 1) */
NLQ_HD uint32_t nlq_bnd_col(const nlq_desc *d, uint32_t q, uint32_t out, uint32_t k) {
    uint32_t col = 0;
    for (uint32_t o = 0; o < 2; o++)
        for (uint32_t i = 0; i < d->n_queues; i++) {
            if (o == out && i == q) return col + k;
            col += d->width[i];
        }
    return col;
}
NLQ_HD uint32_t nlq_bnd_cells(const nlq_desc *d) { return nlq_bnd_col(d, d->n_queues, 1, 0); }

/* ---- components. MEM: 1 timestamp, 2 page, 3 index, 4 rw, 5 value_is_pointer, 6 + t: value nibble t (little end first).
   LOG: 1..8 read_value limbs, 9..16 written_value limbs, 17 timestamp, 18 tx_number, 19 aux_byte, 20 shard_id, 21 rw, 22 is_service,
   23 rollback, 24 + b: key byte b (little end first), 56 + b: address byte b.   DECOMMIT: 1..8 hash limbs, 9..12 memory_page bytes,
   13..16 timestamp bytes, 17 is_fresh. */
#define NLQ_MEM_NIBBLE0 6
/* byte of the 88-byte serialisation that a field cell of a LOG item holds, or -1 */
NLQ_HD int nlq_lh_byte_of_cell(uint32_t cell) {
    if (cell == 20) return 0;                                     /* shard_id */
    if (cell == 22) return 1;                                     /* is_service as a byte */
    if (cell >= 56 && cell < 76) return 4 + (19 - (int)(cell - 56)); /* address, big end first */
    if (cell >= 24 && cell < 56) return 24 + (31 - (int)(cell - 24)); /* key, big end first */
    if (cell >= 76 && cell < 108) return 56 + (31 - (int)(cell - 76)); /* written_value (LOGB), big end first */
    if (cell == 108) return 3;                                     /* tx_number (LOGB): big end first */
    if (cell == 109) return 2;
    return -1;
}
/* can this cell be a copy of a netlist cell (whether it is may depend on the cycle: nlq_link_target) */
NLQ_HD int nlq_comp_linked(const nlq_op *op, uint32_t cell) {
    if (op->link == NLQ_LINK_NONE) return 0;
    if (op->link == NLQ_LINK_LH_MESSAGE) return nlq_lh_byte_of_cell(cell) >= 0;
    return cell >= NLQ_MEM_NIBBLE0 && cell < NLQ_MEM_NIBBLE0 + (op->item == NLQ_ITEM_MEM8 ? 32u : 64u);
}
/* the netlist reference a linked cell copies, and the cycle it is seen from (*next: 1 = the state AFTER the cycle = CYC of cycle + 1);
   the links that do not depend on the cycle */
NLQ_HD uint32_t nlq_link_ref(const nlq_op *op, uint32_t cell, uint32_t *next) {
    const uint32_t t = cell - NLQ_MEM_NIBBLE0;
    if (op->link == NLQ_LINK_KECCAK_DIGEST) { *next = 1; return NL_REF_CYC + (31 - t); }
    if (op->link == NLQ_LINK_EC_OK) { *next = 1; return NL_REF_CYC + (t == 0 ? NLQ_EK_STATE_OK : NLQ_EK_STATE_ZERO); }
    if (op->link == NLQ_LINK_SHA_DIGEST) { *next = 1; return NL_REF_CYC + 8 * (7 - t / 8) + t % 8; }
    *next = 0;
    return NL_REF_FREE + 2 * (32 * op->link_arg + 31 - t / 2) + (t & 1);
}
/* The link of `cell` of an operation of cycle c, if it has one there: the netlist reference and the cycle it is resolved in (a CYC
   reference of cycle c + 1 = the state after c; a FREE reference = that cycle's element). Returns 0 when the cell is free in this cycle
   (L1MessagesHasher: the slot holds no message by the 88 / 136 pattern, or the byte lies beyond the last cycle). */
NLQ_HD int nlq_link_target(const nlq_op *op, uint32_t c, uint32_t capacity, uint32_t cell, uint32_t *cycle, uint32_t *ref) {
    if (!nlq_comp_linked(op, cell)) return 0;
    if (op->link == NLQ_LINK_LH_MESSAGE) {
        const uint64_t m = NLQ_LH_FIRST(c) + op->link_arg;
        if (m >= NLQ_LH_FIRST(c + 1)) return 0;
        const uint64_t pos = 88 * m + (uint64_t)nlq_lh_byte_of_cell(cell);
        if (pos / 136 >= capacity) return 0;
        *cycle = (uint32_t)(pos / 136);
        *ref = NL_REF_FREE + (uint32_t)(pos % 136);
        return 1;
    }
    uint32_t next = 0;
    *ref = nlq_link_ref(op, cell, &next);
    *cycle = c + next;
    return 1;
}

/* enc element j of an item = sum over its terms of cell * 2^shift (in the field): number of terms, term i (no arrays: the kernels
   keep this in registers) */
NLQ_HD uint32_t nlq_enc_n_terms(uint32_t item, uint32_t j) {
    if (item == NLQ_ITEM_MEM) return j < 2 ? 1u : j == 2 ? 3u : j < 7 ? 14u : 8u;
    if (item == NLQ_ITEM_MEM8) return j < 2 ? 1u : j == 2 ? 3u : j < 7 ? 7u : 4u;
    if (item == NLQ_ITEM_LOG || item == NLQ_ITEM_LOGB) return j <= 17 ? 4u : j == 18 ? 2u : 1u;
    return j < 3 ? 4u : 1u;
}
NLQ_HD nlq_term nlq_enc_term(uint32_t item, uint32_t j, uint32_t i) {
    nlq_term t;
    uint32_t cell, shift = 0;
    if (item == NLQ_ITEM_MEM) { /* memory_query.rs:60-110: timestamp | page | index + rw << 32 + is_ptr << 33 | limb w + three rider bytes of limbs 5..7 | limb 4 */
        if (j < 2) cell = 1 + j;
        else if (j == 2) { cell = 3 + i; shift = i ? 31 + i : 0; }
        else if (i < 8) { cell = NLQ_MEM_NIBBLE0 + 8 * (j == 7 ? 4 : j - 3) + i; shift = 4 * i; }
        else { const uint32_t ii = i - 8, byte = 20 + 3 * (j - 3) + ii / 2; cell = NLQ_MEM_NIBBLE0 + 2 * byte + (ii & 1); shift = 32 + 8 * (ii / 2) + 4 * (ii & 1); }
    } else if (item == NLQ_ITEM_MEM8) { /* the same encoding over value BYTES: 6 + x = value byte x, little end first */
        if (j < 2) cell = 1 + j;
        else if (j == 2) { cell = 3 + i; shift = i ? 31 + i : 0; }
        else if (i < 4) { cell = NLQ_MEM_NIBBLE0 + 4 * (j == 7 ? 4 : j - 3) + i; shift = 8 * i; }
        else { cell = NLQ_MEM_NIBBLE0 + 20 + 3 * (j - 3) + (i - 4); shift = 32 + 8 * (i - 4); }
    } else if (item == NLQ_ITEM_LOG || item == NLQ_ITEM_LOGB) { /* log_query.rs:150-360: a limb + three bytes of key ++ address | tx_number, address[19], aux_byte, shard_id | rw + 2 * is_service | rollback */
        if (j < 17) {
            if (i == 0) cell = j < 16 ? 1 + j : 17;
            else { const uint32_t x = 3 * j + (i - 1); cell = x < 32 ? 24 + x : 56 + (x - 32); shift = 32 + 8 * (i - 1); }
        } else if (j == 17) { cell = i == 0 ? 18 : i == 1 ? 56 + 19 : i == 2 ? 19 : 20; shift = i ? 24 + 8 * i : 0; }
        else if (j == 18) { cell = 21 + i; shift = i; }
        else cell = 23;
    } else { /* decommittment_request.rs:20-70: hash limb + three bytes of page / timestamp / is_fresh */
        if (j < 3 && i) { cell = 9 + 3 * j + (i - 1); shift = 32 + 8 * (i - 1); }
        else cell = 1 + j;
    }
    t.cell = (uint16_t)cell; t.shift = (uint16_t)shift;
    return t;
}
/* recomposition gates of an item (NLQ_ITEM_LOGB): relation r says cell nlq_aux_result = sum over its terms of cell * 2^shift.
   r < 8: written_value limb r = its four bytes (cells 76 + 4 r ..); r = 8: tx_number = its two bytes (cells 108, 109) */
NLQ_HD uint32_t nlq_aux_n(uint32_t item) { return item == NLQ_ITEM_LOGB ? 9u : 0u; }
NLQ_HD uint32_t nlq_aux_result(uint32_t item, uint32_t r) { (void)item; return r < 8 ? 9u + r : 18u; }
/* the gate whose result `cell` is, or -1 (the fill computes such a cell from the gate's terms: in a disabled operation the byte cells
   may be copies of padding bytes, and the limb must still be their recomposition) */
NLQ_HD int nlq_aux_of_cell(uint32_t item, uint32_t cell) {
    if (item != NLQ_ITEM_LOGB) return -1;
    return cell >= 9 && cell <= 16 ? (int)(cell - 9) : cell == 18 ? 8 : -1;
}
NLQ_HD uint32_t nlq_aux_n_terms(uint32_t item, uint32_t r) { (void)item; return r < 8 ? 4u : 2u; }
NLQ_HD nlq_term nlq_aux_term(uint32_t item, uint32_t r, uint32_t i) {
    nlq_term t;
    (void)item;
    t.cell = (uint16_t)(r < 8 ? 76 + 4 * r + i : 108 + i);
    t.shift = (uint16_t)(8 * i);
    return t;
}

/* component `cell` (>= 1) of an item record (zkw_mem_query / zkw_log_query / zkw_decommit_query); a null record is all zeros */
NLQ_HD uint64_t nlq_item_component(uint32_t item, const void *rec, uint32_t cell) {
    if (!rec) return 0;
    if (item == NLQ_ITEM_MEM) {
        const zkw_mem_query *q = (const zkw_mem_query *)rec;
        switch (cell) {
            case 1: return q->timestamp;
            case 2: return q->page;
            case 3: return q->index;
            case 4: return q->rw_flag ? 1 : 0;
            case 5: return q->value_is_pointer ? 1 : 0;
            default: { const uint32_t t = cell - NLQ_MEM_NIBBLE0; return (q->value[t / 8] >> (4 * (t % 8))) & 15u; }
        }
    }
    if (item == NLQ_ITEM_MEM8) {
        const zkw_mem_query *q = (const zkw_mem_query *)rec;
        switch (cell) {
            case 1: return q->timestamp;
            case 2: return q->page;
            case 3: return q->index;
            case 4: return q->rw_flag ? 1 : 0;
            case 5: return q->value_is_pointer ? 1 : 0;
            default: { const uint32_t t = cell - NLQ_MEM_NIBBLE0; return (q->value[t / 4] >> (8 * (t % 4))) & 255u; }
        }
    }
    if (item == NLQ_ITEM_LOG || item == NLQ_ITEM_LOGB) {
        const zkw_log_query *q = (const zkw_log_query *)rec;
        if (cell >= 76) { /* LOGB: written_value bytes (little end first), tx_number bytes */
            if (cell < 108) { const uint32_t b = cell - 76; return (q->written_value[b / 4] >> (8 * (b % 4))) & 255u; }
            return ((uint32_t)q->tx_number_in_block >> (8 * (cell - 108))) & 255u;
        }
        if (cell <= 8) return q->read_value[cell - 1];
        if (cell <= 16) return q->written_value[cell - 9];
        switch (cell) {
            case 17: return q->timestamp;
            case 18: return q->tx_number_in_block;
            case 19: return q->aux_byte;
            case 20: return q->shard_id;
            case 21: return q->rw_flag ? 1 : 0;
            case 22: return q->is_service ? 1 : 0;
            case 23: return q->rollback ? 1 : 0;
            default: break;
        }
        if (cell < 56) { const uint32_t b = cell - 24; return (q->key[b / 4] >> (8 * (b % 4))) & 255u; }
        { const uint32_t b = cell - 56; return (q->address[b / 4] >> (8 * (b % 4))) & 255u; }
    }
    {
        const zkw_decommit_query *q = (const zkw_decommit_query *)rec;
        if (cell <= 8) return q->hash[cell - 1];
        if (cell <= 12) return (q->memory_page >> (8 * (cell - 9))) & 255u;
        if (cell <= 16) return (q->timestamp >> (8 * (cell - 13))) & 255u;
        return q->is_fresh ? 1 : 0;
    }
}
NLQ_HD uint32_t nlq_item_bytes(uint32_t item) {
    return item == NLQ_ITEM_MEM || item == NLQ_ITEM_MEM8 ? (uint32_t)sizeof(zkw_mem_query) : item == NLQ_ITEM_LOG || item == NLQ_ITEM_LOGB ? (uint32_t)sizeof(zkw_log_query) : (uint32_t)sizeof(zkw_decommit_query);
}
#endif /* ZKW_NETLIST_QUEUE_H */
