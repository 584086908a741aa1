/* zkw_netlist_closed_form.h — the CLOSED-FORM SECTION of a "zkw trace v4" netlist circuit (types 3, 5, 6, 7, 10, 13): what the
 * reference's circuits derive in-circuit around their round function — the commitments of the observable input / output and of the
 * hidden FSM input / output, the compact form and the public input (ClosedFormInputCompactForm::from_full_form +
 * commit_variable_length_encodable_item, src/witness/utils.rs:269-306), and the start-flag selection of the state the first cycle
 * continues from (`state = start_flag ? fresh : hidden_fsm_input`, mirrored out of circuit by
 * src/witness/individual_circuits/sha256_round_function.rs:172-201, keccak256_round_function.rs:214-231, decommit_code.rs:172-199,
 * ecrecover.rs:120-141) — as rows of the trace, the way docs/KERNELS.md 3.20 does it for the six queue circuits. Until round 5 the
 * PI row of these traces was PLACED: any public input passed the checkers.
 *
 * Layout only (record types, cell positions, the tie tables); the arithmetic is csrc/netlist_closed_form_kernels.cuh on the device and
 * oracle/netlist_closed_form.c in the test oracle. Placement is this library's own (the gadget bodies are in the absent
 * era-zkevm_circuits crate): PARITY UNPINNED at the placement level, like the rest of v4.
 *
 * Rows. The section starts at row c0 = nlcf_first_row(): below the queue section (below the EC section for ECRecover), general-purpose
 * columns only (cell k of a block at row k / G, column k % G; lookup cells zero, not counted in the multiplicity column).
 *   HEADER block: [start | completion | OI words | OO words | FI words | FO words | tie cells]. A word is one element of the flat
 *     encodings the commitments absorb (oracle/public_input.c, csrc/public_input_kernels.cuh: field order of the reference's struct
 *     literals), FREE unless a tie binds it. start / completion boolean.
 *   TIE cells [a | b | r_0 .. r_{n-1}] bind a word to a REGISTER of the trace (a QBND cell of the queue section: the state of a queue
 *     before cycle 0 / after the last cycle; or state elements of the netlist's BND_IN / BND_OUT rows, n digits of `bits` bits):
 *     r_t = copy of the register cell, R = sum r_t << (bits t);
 *       IN:         a = copy of OI word (or the constant 0), b = copy of FI word;  R = b + start (a - b)
 *       IN_ALWAYS:  a = copy of OI word;                                           R = a         (circuits without a hidden FSM)
 *       OUT:        a = copy of FO word;                                           R = a
 *       OUT_LIVE:   a = copy of FO word;                                           (1 - completion) (R - a) = 0   (the hash state: an instance
 *                   that ends its block early leaves the state of an EMPTY round in the FSM output — keccak256_round_function.rs:376-394,
 *                   sha256_round_function.rs:262-279 —, where this library's idle cycles carry the last digest; nothing consumes that output)
 *       OO:         a = copy of OO word, b = copy of the FO word of the register;  R = b, a = completion b   (b absent: R = a = completion R)
 *       DONE:       a = copy of FO word;                                           completion (R - a) = 0   (the popped queue's head after the
 *                   last pop of the LAST instance is its tail: every request of the block was served)
 *     The register of an IN tie may also be a word of the FSM OUTPUT (NLCF_REG_FO_WORD): the far end of a queue — the tail of the popped
 *     queue, the head of the memory queue — never moves, FO word = start ? OI word : FI word.
 *     The GATED kinds tie the FSM words that the relations of the queue section (zkw_netlist_queue.h, nlq_rel with prev = 1) carry from
 *     cycle to cycle — the word offset / page / timestamp of the next read, the page / offset to write, the rounds left, "the round before
 *     wrote a digest" — at the two ends of an instance, where those relations have no neighbour cycle. Cells [a | b | r | g1 | g2]:
 *     r, g1, g2 = copies of a cell of an operation and of the enables of two operations (g2 may be absent) in cycle 0 / the last cycle;
 *       IN_GATED:   a = copy of OI word or the constant a_const, b = copy of FI word;  (g1 - g2) (r - (b + start (a - b)) - add) = 0
 *                   — the relation of cycle c >= 1 with the FSM input in the place of cycle c - 1
 *       OUT_GATED:  a = copy of FO word;                                  (1 - completion) (g1 - g2) (a -/+ r - add) = 0   (no gate cells: 1)
 *                   — what the next instance continues from is what this one's last cycle holds (a non-final instance has no idle cycle)
 *     A gate NLCF_GATE_ACTIVE stands for 1 - idle of the cycle's netlist header (its digit copies the idle cell).
 *   P2 blocks (the 130 variables of the flattened Poseidon2 gate, ceil(130 / G) rows each): the four sponges in overwrite mode from
 *     the state (0, .., 0, n) — permutation p absorbs words 8p .. 8p + 7 (copies; constants 0 beyond n) over the capacity the
 *     permutation before left (copies) —, then the three permutations over the 18 words of the compact form [start, completion,
 *     c(OI), c(OO), c(FI), c(FO)] (copies of the sponges' last outputs 0..3; the constant 0 for an empty encoding). The PI row's four
 *     cells are copies of the last permutation's outputs 0..3.
 * What stays only committed (FREE words): everything the ties below do not name — the `completed` / `padding_round` flags and the
 * write timestamp of the internal FSMs, Keccak's byte offset / length / buffer and the decommitter's round count and length (no registers
 * for them in the queue section), the queue LENGTHS, StorageApplication's words other than the OUTGOING root (NLCF_DESC_STORAGE_APPLICATION
 * below ties the FSM output's current_root_hash / the observable output's new_root_hash to the state after the instance's last cycle; the root an
 * instance STARTS from — the FSM input's / the observable input's initial_root_hash — is not tied: the trace has no register the first walk is
 * compared with (the queue side type 10 lacks), so the chain of roots between instances is carried by the commitments only — a known gap).
 */
#ifndef ZKW_NETLIST_CLOSED_FORM_H
#define ZKW_NETLIST_CLOSED_FORM_H
#include "zkw_netlist_queue.h"
#include "zkw_ecrecover_ec_spec.h"

enum { NLCF_OI = 0, NLCF_OO = 1, NLCF_FI = 2, NLCF_FO = 3 };
enum { NLCF_IN = 1, NLCF_IN_ALWAYS = 2, NLCF_OUT = 3, NLCF_OUT_OO = 4, NLCF_OUT_LIVE = 5, NLCF_IN_GATED = 6, NLCF_OUT_GATED = 7, NLCF_DONE = 8 };
enum { NLCF_REG_QUEUE_BEFORE = 0, NLCF_REG_QUEUE_AFTER = 1, NLCF_REG_STATE_IN = 2, NLCF_REG_STATE_OUT = 3,
       NLCF_REG_OP_FIRST = 4 /* a cell of an operation of the queue section in cycle 0 */, NLCF_REG_OP_LAST = 5 /* ... in the last cycle */,
       NLCF_REG_FO_WORD = 6 /* a word of the FSM output (the header block's own cell): what an instance hands on unchanged */ };
#define NLCF_MAX_GROUPS 32
#define NLCF_NO_GATE 0xFF
#define NLCF_GATE_ACTIVE 0xFE /* as a gate: 1 - idle of the cycle's netlist header (the digit is a copy of the idle cell) */
#define NLCF_CP_WORDS 18
#define NLCF_CP_PERMS 3

/* a run of `count` ties: tie j binds register reg0 + j (QUEUE: element of the queue state; STATE: elements (reg0 + j) * n_cells ..)
   to words a_word0 + j / b_word0 + j (lanes_xy: the FSM holds Keccak's bytes as [x][y][8], the netlist as lane x + 5 y) */
typedef struct nlcf_group { uint8_t kind, reg_kind, queue, n_cells, bits, lanes_xy; uint16_t count, reg0; int16_t a_word0, b_word0;
                            /* the GATED kinds (count = 1; queue = the operation, reg0 = its cell): */ int8_t add; uint8_t a_const, gate, gate2, negate; } nlcf_group;
typedef struct nlcf_desc { uint16_t n[4]; uint16_t n_groups; nlcf_group g[NLCF_MAX_GROUPS]; } nlcf_desc;

#if defined(__GNUC__)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wmissing-field-initializers" /* the ungated kinds leave {add, a_const, gate, gate2, negate} zero */
#endif
/* Sha256RoundFunction (6). OI = PrecompileFunctionInputData {log queue: head 0..3, tail 4..7, length 8; memory queue: head 9..20, tail
   21..32, length 33}; OO = the final memory queue state (head 0..11, tail 12..23, length 24); FSM = 3 flags, sha256_inner_state 3..10,
   2 timestamps, 5 call parameters, log queue 18..26, memory queue 27..51. The netlist's state = the chaining value as 8 x 8 nibbles. */
/* ... and the internal FSM's address arithmetic at the two ends of an instance (operations of NLQ_DESC_SHA256: 0 pop, 1 / 2 reads, 3 digest
   write with its registers 102 page to write, 103 offset to write, 104 rounds left; memory-query cells 1 timestamp, 2 page, 3 index). FSM
   words: 0 read_precompile_call, 1 read_words_for_round, 11 timestamp_to_use_for_read, 13 input_page, 14 input_offset = the next word to
   read, 15 output_page, 16 output_offset, 17 num_rounds = rounds left (sha256_round_function.rs:204-246,302-316). */
static const nlcf_desc NLCF_DESC_SHA256 = {{34, 25, 52, 52}, 25, {
    {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 0, 1, 0, 0, 4, 0, 0, 18}, {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 1, 1, 0, 0, 12, 0, 21, 39},
    {NLCF_IN, NLCF_REG_STATE_IN, 0, 8, 4, 0, 8, 0, -1, 3},
    {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 4, 0, 18, -1}, {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 39, -1},
    {NLCF_OUT_LIVE, NLCF_REG_STATE_OUT, 0, 8, 4, 0, 8, 0, 3, -1},
    {NLCF_OUT_OO, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 12, 39},
    /* cycle 0 continues a request (it reads and does not pop): the first read is the next word of the FSM's page at its timestamp, the registers are the FSM's */
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 1, 3, 0, 0, 1, 3, -1, 14, 0, 0, 1, 0, 0}, {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 1, 3, 0, 0, 1, 2, -1, 13, 0, 0, 1, 0, 0},
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 1, 3, 0, 0, 1, 1, -1, 11, 0, 0, 1, 0, 0},
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 3, 3, 0, 0, 1, 102, -1, 15, 0, 0, 1, 0, 0}, {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 3, 3, 0, 0, 1, 103, -1, 16, 0, 0, 1, 0, 0},
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 3, 3, 0, 0, 1, 104, -1, 17, -1, 0, 1, 0, 0},
    /* an active cycle 0 pops a call exactly when the FSM says so (a first instance: always) */
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 0, 2, 0, 0, 1, 0, -1, 0, 0, 1, 1, NLCF_NO_GATE, 0},
    /* the FSM output of a non-final instance is what its last cycle holds */
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 2, 1, 0, 0, 1, 3, 14, -1, 1, 0, 0, 0, 0}, {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 2, 1, 0, 0, 1, 2, 13, -1, 0, 0, 0, 0, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 2, 1, 0, 0, 1, 1, 11, -1, 0, 0, 0, 0, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 3, 1, 0, 0, 1, 102, 15, -1, 0, 0, 0, 0, 0}, {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 3, 1, 0, 0, 1, 103, 16, -1, 0, 0, 0, 0, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 3, 1, 0, 0, 1, 104, 17, -1, 0, 0, 0, 0, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 3, 1, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0} /* read_precompile_call = the last cycle wrote a digest */,
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 3, 1, 0, 0, 1, 0, 1, -1, 1, 0, 0, 0, 1} /* read_words_for_round = 1 - that */,
    /* the far ends never move; the last instance leaves the popped queue empty */
    {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 4, 22, 4, 22}, {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 12, 27, 9, 27}, {NLCF_DONE, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 4, 0, 22, -1}}};
/* CodeDecommitter (3). OI = CodeDecommitterInputData {memory queue 0..24, sorted requests queue 25..49}; OO = the final memory queue
   state; FSM = sha256_inner_state 0..7, hash_to_compare_against 8..15, 5 counters, 3 flags, requests queue 24..48, memory queue 49..73.
   Queue 0 of the section = the requests (popped: head), queue 1 = the memory queue (pushed: tail). */
/* ... FSM words 16 current_index = the next word to write, 17 current_page, 18 timestamp, 21 state_get_from_queue, 22 state_decommit
   (decommit_code.rs:228-350); operations of NLQ_DESC_CODE_DECOMMITTER: 0 pop, 1 first word (every active cycle), 2 second word (absent in a
   bytecode's last round). */
static const nlcf_desc NLCF_DESC_CODE_DECOMMITTER = {{50, 25, 74, 74}, 21, {
    {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 0, 1, 0, 0, 12, 0, 25, 24}, {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 1, 1, 0, 0, 12, 0, 12, 61},
    {NLCF_IN, NLCF_REG_STATE_IN, 0, 8, 4, 0, 8, 0, -1, 0},
    {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 12, 0, 24, -1}, {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 61, -1},
    {NLCF_OUT_LIVE, NLCF_REG_STATE_OUT, 0, 8, 4, 0, 8, 0, 0, -1},
    {NLCF_OUT_OO, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 12, 61},
    /* cycle 0 continues a bytecode: its first word goes where the FSM says; it pops a request exactly when the FSM says so */
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 1, 3, 0, 0, 1, 3, -1, 16, 0, 0, 1, 0, 0}, {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 1, 3, 0, 0, 1, 2, -1, 17, 0, 0, 1, 0, 0},
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 1, 3, 0, 0, 1, 1, -1, 18, 0, 0, 1, 0, 0},
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 0, 2, 0, 0, 1, 0, -1, 21, 0, 1, 1, NLCF_NO_GATE, 0},
    /* a non-final instance leaves: the index after the last word written (the second word when the last cycle has one), page, timestamp, the state */
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 2, 2, 0, 0, 1, 3, 16, -1, 1, 0, 2, NLCF_NO_GATE, 0}, {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 1, 3, 0, 0, 1, 3, 16, -1, 1, 0, 1, 2, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 1, 1, 0, 0, 1, 2, 17, -1, 0, 0, 0, 0, 0}, {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 1, 1, 0, 0, 1, 1, 18, -1, 0, 0, 0, 0, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 2, 1, 0, 0, 1, 0, 21, -1, 1, 0, 0, 0, 1} /* state_get_from_queue = 1 - "the last cycle wrote a second word" */,
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 2, 1, 0, 0, 1, 0, 22, -1, 0, 0, 0, 0, 0} /* state_decommit = that */,
    /* the far ends never move; the last instance leaves the popped queue empty */
    {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 12, 36, 37, 36}, {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 12, 49, 0, 49}, {NLCF_DONE, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 12, 0, 36, -1}}};
/* Keccak256RoundFunction (5). OI / OO as type 6; FSM = 4 flags, keccak_internal_state 4..203 ([x][y][8] bytes), 2 timestamps, 6 call
   parameters, the byte buffer 212..403 and its fill 404, log queue 405..413, memory queue 414..438. The netlist's state = the sponge
   state as 200 bytes, lane x + 5 y. */
/* ... FSM words 0 read_precompile_call, 209 output_page, 210 output_offset (keccak256_round_function.rs:420-441); operations of
   NLQ_DESC_KECCAK256: 0 pop, 1..6 unaligned reads, 7 digest write with its registers 70 page to write, 71 offset to write. (The byte
   offset / length of the input and the byte buffer have no register in the queue section: committed only.) */
static const nlcf_desc NLCF_DESC_KECCAK256 = {{34, 25, 439, 439}, 16, {
    {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 0, 1, 0, 0, 4, 0, 0, 405}, {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 1, 1, 0, 0, 12, 0, 21, 426},
    {NLCF_IN, NLCF_REG_STATE_IN, 0, 1, 8, 1, 200, 0, -1, 4},
    {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 4, 0, 405, -1}, {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 426, -1},
    {NLCF_OUT_LIVE, NLCF_REG_STATE_OUT, 0, 1, 8, 1, 200, 0, 4, -1},
    {NLCF_OUT_OO, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 12, 426},
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 7, 3, 0, 0, 1, 70, -1, 209, 0, 0, NLCF_GATE_ACTIVE, 0, 0}, {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 7, 3, 0, 0, 1, 71, -1, 210, 0, 0, NLCF_GATE_ACTIVE, 0, 0},
    {NLCF_IN_GATED, NLCF_REG_OP_FIRST, 0, 2, 0, 0, 1, 0, -1, 0, 0, 1, NLCF_GATE_ACTIVE, NLCF_NO_GATE, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 7, 1, 0, 0, 1, 70, 209, -1, 0, 0, 0, 0, 0}, {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 7, 1, 0, 0, 1, 71, 210, -1, 0, 0, 0, 0, 0},
    {NLCF_OUT_GATED, NLCF_REG_OP_LAST, 7, 1, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0} /* read_precompile_call = the last cycle wrote a digest */,
    /* the far ends never move; the last instance leaves the popped queue empty */
    {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 4, 409, 4, 409}, {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 12, 414, 9, 414}, {NLCF_DONE, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 4, 0, 409, -1}}};
/* ECRecover (7). OI / OO as type 6; FSM = EcrecoverCircuitFSMInputOutput {log queue 0..8, memory queue 9..33} (ecrecover.rs:226-233): a
   cycle is a whole request, the netlist carries nothing between cycles. */
static const nlcf_desc NLCF_DESC_ECRECOVER = {{34, 25, 34, 34}, 8, {
    {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 0, 1, 0, 0, 4, 0, 0, 0}, {NLCF_IN, NLCF_REG_QUEUE_BEFORE, 1, 1, 0, 0, 12, 0, 21, 21},
    {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 4, 0, 0, -1}, {NLCF_OUT, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 21, -1},
    {NLCF_OUT_OO, NLCF_REG_QUEUE_AFTER, 1, 1, 0, 0, 12, 0, 12, 21},
    /* the far ends never move; the last instance leaves the popped queue empty */
    {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 4, 4, 4, 4}, {NLCF_IN, NLCF_REG_FO_WORD, 0, 1, 0, 0, 12, 9, 9, 9}, {NLCF_DONE, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 4, 0, 4, -1}}};
/* L1MessagesHasher (13). OI = LinearHasherInputData {queue_state: head 0..3, tail 4..7, length 8}, OO = LinearHasherOutputData
   {keccak256_hash: 32 bytes}, no hidden FSM (one instance per block: data_hasher_and_merklizer.rs:34-60). The pops start at the
   queue's head and END AT ITS TAIL — every message of the queue is hashed, none left out (the reference's circuit pops until the queue
   is empty and the queue's consistency check compares head and tail then); the digest is the first 32 bytes of the sponge state after
   the last cycle. A caller's queue_state must be the queue's: tail = the state after its last push (head, when it is empty). */
static const nlcf_desc NLCF_DESC_LINEAR_HASHER = {{9, 32, 0, 0}, 3, {
    {NLCF_IN_ALWAYS, NLCF_REG_QUEUE_BEFORE, 0, 1, 0, 0, 4, 0, 0, -1}, {NLCF_IN_ALWAYS, NLCF_REG_QUEUE_AFTER, 0, 1, 0, 0, 4, 0, 4, -1},
    {NLCF_OUT_OO, NLCF_REG_STATE_OUT, 0, 1, 8, 0, 32, 0, 0, -1}}};
/* StorageApplication (10). OI = {shard, initial_root_hash 32 bytes, enumeration counter 2, log queue 9} = 44, OO = {new_root_hash 32,
   counter 2, state_diffs_keccak256_hash 32} = 66, FSM = {root hash 32, counter 2, log queue 9, Keccak accumulator 200} = 243
   (storage_application.rs:286-336). The trace holds the Blake2s walks only (no queue side, docs/KERNELS.md 3.17): the ROOT an instance
   hands on is the running hash after its last cycle (the root its last walk computed; an instance without walks carries its root through
   its idle cycles), and the block's new_root_hash is that of the last instance; the other words are committed, not tied. */
static const nlcf_desc NLCF_DESC_STORAGE_APPLICATION = {{44, 66, 243, 243}, 2, {
    {NLCF_OUT, NLCF_REG_STATE_OUT, 0, 1, 8, 0, 32, 0, 0, -1}, {NLCF_OUT_OO, NLCF_REG_STATE_OUT, 0, 1, 8, 0, 32, 0, 0, 0}}};

#if defined(__GNUC__)
#pragma GCC diagnostic pop
#endif

static inline const nlcf_desc *nlcf_desc_of(int circuit_type) {
    return circuit_type == 6 ? &NLCF_DESC_SHA256 : circuit_type == 3 ? &NLCF_DESC_CODE_DECOMMITTER : circuit_type == 5 ? &NLCF_DESC_KECCAK256 :
           circuit_type == 7 ? &NLCF_DESC_ECRECOVER : circuit_type == 13 ? &NLCF_DESC_LINEAR_HASHER :
           circuit_type == 10 ? &NLCF_DESC_STORAGE_APPLICATION : (const nlcf_desc *)0;
}

/* ---- cells of the HEADER block */
#define NLCF_CELL_START 0u
#define NLCF_CELL_COMPLETION 1u
NLQ_HD uint32_t nlcf_word_cell(const nlcf_desc *d, uint32_t part, uint32_t k) {
    uint32_t c = 2;
    for (uint32_t p = 0; p < part; p++) c += d->n[p];
    return c + k;
}
NLQ_HD uint32_t nlcf_group_cell0(const nlcf_desc *d, uint32_t gi) {
    uint32_t c = nlcf_word_cell(d, 4, 0);
    for (uint32_t i = 0; i < gi; i++) c += (uint32_t)d->g[i].count * (2u + d->g[i].n_cells);
    return c;
}
NLQ_HD uint32_t nlcf_header_cells(const nlcf_desc *d) { return nlcf_group_cell0(d, d->n_groups); }
/* tie j of group gi: its first cell; the word index of its `a` / `b` side (-1: none) */
NLQ_HD uint32_t nlcf_tie_cell0(const nlcf_desc *d, uint32_t gi, uint32_t j) { return nlcf_group_cell0(d, gi) + j * (2u + d->g[gi].n_cells); }
NLQ_HD int32_t nlcf_tie_word(const nlcf_group *g, int32_t word0, uint32_t j) {
    if (word0 < 0) return -1;
    if (!g->lanes_xy) return word0 + (int32_t)j;
    const uint32_t lane = j / 8, x = lane % 5, y = lane / 5; /* netlist byte j of lane x + 5 y = FSM byte [x][y][j % 8] */
    return word0 + (int32_t)(8 * (5 * x + y) + j % 8);
}
/* which part the words of a tie's sides belong to */
NLQ_HD uint32_t nlcf_a_part(const nlcf_group *g) { return g->kind == NLCF_OUT || g->kind == NLCF_OUT_LIVE || g->kind == NLCF_OUT_GATED || g->kind == NLCF_DONE ? NLCF_FO : g->kind == NLCF_OUT_OO ? NLCF_OO : NLCF_OI; }
NLQ_HD uint32_t nlcf_b_part(const nlcf_group *g) { return g->kind == NLCF_OUT_OO ? NLCF_FO : NLCF_FI; }

/* ---- P2 blocks: sponge of part p = perms [nlcf_perm0(p), nlcf_perm0(p + 1)), then the compact form's three */
NLQ_HD uint32_t nlcf_part_perms(const nlcf_desc *d, uint32_t part) { return (d->n[part] + 7u) / 8u; }
NLQ_HD uint32_t nlcf_perm0(const nlcf_desc *d, uint32_t part) {
    uint32_t c = 0;
    for (uint32_t p = 0; p < part && p < 4; p++) c += nlcf_part_perms(d, p);
    return c;
}
NLQ_HD uint32_t nlcf_n_perms(const nlcf_desc *d) { return nlcf_perm0(d, 4) + NLCF_CP_PERMS; }
NLQ_HD uint32_t nlcf_header_rows(const nlcf_desc *d, uint32_t g) { return nlq_rows_for(nlcf_header_cells(d), g); }
NLQ_HD uint32_t nlcf_rows(const nlcf_desc *d, uint32_t g) { return nlcf_header_rows(d, g) + nlcf_n_perms(d) * nlq_rows_for(NLQ_P2_CELLS, g); }
/* row (relative to the section's first row) of P2 block `perm` */
NLQ_HD uint32_t nlcf_perm_row0(const nlcf_desc *d, uint32_t g, uint32_t perm) { return nlcf_header_rows(d, g) + perm * nlq_rows_for(NLQ_P2_CELLS, g); }

/* ---- where the section is (host only: kernels take the first row and the descriptor by value) */
static inline uint64_t nlcf_first_row(int circuit_type, const nl_spec *sp, uint32_t cycles) {
    return nlq_used_rows(sp, nlq_desc_of(circuit_type), cycles) + (circuit_type == 7 ? (uint64_t)cycles * EC_ROWS_PER_CYCLE : 0);
}
/* rows a trace of `cycles` cycles uses: netlist + queue section (+ EC section) + closed-form section */
static inline uint64_t nlcf_used_rows(int circuit_type, const nl_spec *sp, uint32_t cycles) {
    const nlcf_desc *d = nlcf_desc_of(circuit_type);
    return nlcf_first_row(circuit_type, sp, cycles) + (d ? nlcf_rows(d, sp->g) : 0);
}
#endif /* ZKW_NETLIST_CLOSED_FORM_H */
