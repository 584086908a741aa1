/* precompiles.c — TEST INFRASTRUCTURE: CPU restatement, sequential like the reference, of
 *   keccak256_decompose_into_per_circuit_witness  src/witness/individual_circuits/keccak256_round_function.rs:23-528
 *   sha256_decompose_into_per_circuit_witness     src/witness/individual_circuits/sha256_round_function.rs:23-406
 *   ecrecover_decompose_into_per_circuit_witness  src/witness/individual_circuits/ecrecover.rs:12-262
 *
 * From absent crates (restated, see include/zkw_types.h and DESIGN.md "inferred constants"): PrecompileCallABI
 * (zkevm_opcode_defs v1.4.1: the request's key as four u64: [in_offset | in_length << 32, out_offset | out_length
 * << 32, page_to_read | page_to_write << 32, precompile_interpreted_data]); the keccak ByteBuffer
 * (zk_evm_abstractions v1.4.1: fill appends, consume::<N> returns the first N bytes and shifts the rest down).
 * The hash cores are the RustCrypto eager block functions: `update` with exactly one block applies one
 * compression / permutation, `transmute_state` exposes the raw state.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

extern const uint32_t ORC_SHA256_IV[8];
void orc_sha256_compress(uint32_t state[8], const uint8_t block[64]);
void orc_keccak_f1600(uint64_t a[25]);

typedef struct {
    uint32_t input_memory_offset, input_memory_length, output_memory_offset, output_memory_length;
    uint32_t memory_page_to_read, memory_page_to_write;
    uint64_t precompile_interpreted_data;
} abi_t;

static abi_t precompile_abi_in_log(const zkw_log_query *q) {
    abi_t a;
    a.input_memory_offset = q->key[0]; a.input_memory_length = q->key[1];
    a.output_memory_offset = q->key[2]; a.output_memory_length = q->key[3];
    a.memory_page_to_read = q->key[4]; a.memory_page_to_write = q->key[5];
    a.precompile_interpreted_data = (uint64_t)q->key[6] | ((uint64_t)q->key[7] << 32);
    return a;
}

static void word_be(const uint32_t *w, uint8_t out[32]) { /* U256::to_big_endian */
    for (int k = 0; k < 8; k++) {
        uint32_t limb = w[7 - k];
        out[4 * k] = (uint8_t)(limb >> 24); out[4 * k + 1] = (uint8_t)(limb >> 16);
        out[4 * k + 2] = (uint8_t)(limb >> 8); out[4 * k + 3] = (uint8_t)limb;
    }
}

static void encode_keccak_state(const uint64_t st[25], uint8_t out[200]) { /* :530-541 */
    for (int idx = 0; idx < 25; idx++) {
        const int i = idx % 5, j = idx / 5;
        for (int b = 0; b < 8; b++) out[(i * 5 + j) * 8 + b] = (uint8_t)(st[idx] >> (8 * b));
    }
}

static void keccak_absorb_block(uint64_t st[25], const uint8_t block[136]) {
    for (int k = 0; k < 17; k++) {
        uint64_t lane = 0;
        for (int b = 0; b < 8; b++) lane |= (uint64_t)block[8 * k + b] << (8 * b);
        st[k] ^= lane;
    }
    orc_keccak_f1600(st);
}

/* the running state of the walk shared by the three circuits */
typedef struct {
    int kind;
    const uint64_t *req_tails; /* [n_req][4] new tails of the demuxed request queue (pushed from the empty queue) */
    size_t n_req, popped;
    const zkw_queue_state12 *mem_in;
    const uint64_t *mem_tails;
    size_t pushed, reads;
} walk_t;

static void log_state(const walk_t *w, zkw_queue_state4 *s) { /* take_queue_state_from_simulator after `popped` pops */
    memset(s, 0, sizeof *s);
    if (w->popped) memcpy(s->head, w->req_tails + 4 * (w->popped - 1), 32);
    if (w->n_req) memcpy(s->tail, w->req_tails + 4 * (w->n_req - 1), 32);
    s->length = (uint32_t)(w->n_req - w->popped);
}

static void mem_state(const walk_t *w, zkw_queue_state12 *s) { /* take_sponge_like_queue_state_from_simulator */
    *s = *w->mem_in;
    if (w->pushed) memcpy(s->tail, w->mem_tails + 12 * (w->pushed - 1), 96);
    s->length = w->mem_in->length + (uint32_t)w->pushed;
}

/* Outputs: mem_enc [n_q*8], mem_tails [n_q*12] (the given queries appended to the memory queue), instances.
   Returns the number of instances, or <0 when one of the reference's asserts fails. */
int64_t orc_precompile_build_ex(int kind, const zkw_log_query *requests, const uint64_t *req_tails, size_t n_req,
                                const zkw_mem_query *mem_q, size_t n_q, uint32_t capacity, const zkw_queue_state12 *mem_in,
                                uint64_t *mem_enc, uint64_t *mem_tails, zkw_precompile_instance *instances,
                                zkw_keccak_round_record *keccak_rounds);
/* sha256 only, may be NULL (set before orc_precompile_build_ex; test infrastructure, single-threaded use) */
static zkw_sha256_round_record *g_sha_rounds = NULL;
void orc_precompile_set_sha256_rounds(zkw_sha256_round_record *r) { g_sha_rounds = r; }
int64_t orc_precompile_build(int kind, const zkw_log_query *requests, const uint64_t *req_tails, size_t n_req,
                             const zkw_mem_query *mem_q, size_t n_q, uint32_t capacity, const zkw_queue_state12 *mem_in,
                             uint64_t *mem_enc, uint64_t *mem_tails, zkw_precompile_instance *instances) {
    return orc_precompile_build_ex(kind, requests, req_tails, n_req, mem_q, n_q, capacity, mem_in, mem_enc, mem_tails, instances, NULL);
}
/* keccak_rounds (keccak256 only, may be NULL): one record per Keccak-f call in the global round order, at most
   n_q + n_req of them — the witness of the Keccak256RoundFunction circuit's cycles (keccak_circuit.c) */
int64_t orc_precompile_build_ex(int kind, const zkw_log_query *requests, const uint64_t *req_tails, size_t n_req,
                                const zkw_mem_query *mem_q, size_t n_q, uint32_t capacity, const zkw_queue_state12 *mem_in,
                                uint64_t *mem_enc, uint64_t *mem_tails, zkw_precompile_instance *instances,
                                zkw_keccak_round_record *keccak_rounds) {
    if (capacity == 0) return -2;
    orc_encode_memory_queries(mem_q, n_q, mem_enc);
    orc_queue_push_chain_full(mem_enc, n_q, mem_in->tail, mem_tails);
    walk_t w = {kind, req_tails, n_req, 0, mem_in, mem_tails, 0, 0};

    uint32_t sha_empty[8];
    uint8_t keccak_empty[200], zero_block[136] = {0};
    memcpy(sha_empty, ORC_SHA256_IV, 32);
    orc_sha256_compress(sha_empty, zero_block);
    uint64_t ke[25] = {0};
    keccak_absorb_block(ke, zero_block);
    encode_keccak_state(ke, keccak_empty);

    if (n_req == 0) { /* the dummy instance, keccak :88-157, sha256 :82-150, ecrecover :60-103 */
        if (n_q) return -3;
        zkw_precompile_instance *o = instances;
        memset(o, 0, sizeof *o);
        o->start_flag = o->completion_flag = 1;
        log_state(&w, &o->initial_log_queue_state);
        o->initial_memory_queue_state = *mem_in;
        o->final_memory_state = *mem_in;
        o->hidden_fsm_input.log_queue_state = o->initial_log_queue_state;
        o->hidden_fsm_input.memory_queue_state = *mem_in;
        o->hidden_fsm_output.log_queue_state = o->initial_log_queue_state;
        o->hidden_fsm_output.memory_queue_state = *mem_in;
        if (kind != ZKW_PRECOMPILE_ECRECOVER) {
            o->hidden_fsm_input.read_precompile_call = 1;
            o->hidden_fsm_output.completed = 1;
            if (kind == ZKW_PRECOMPILE_SHA256) memcpy(o->hidden_fsm_output.sha256_inner_state, sha_empty, 32);
            else memcpy(o->hidden_fsm_output.keccak_internal_state, keccak_empty, 200);
        }
        return 1;
    }

    size_t n_inst = 0, round_counter = 0, total_rounds = 0, first_round = 0, first_req = 0, first_read = 0;
    zkw_precompile_fsm fsm_in;
    memset(&fsm_in, 0, sizeof fsm_in);
    if (kind != ZKW_PRECOMPILE_ECRECOVER) fsm_in.read_precompile_call = 1;
    zkw_queue_state4 log_in;
    zkw_queue_state12 memq_in = *mem_in;
    log_state(&w, &log_in);
    const zkw_queue_state4 log_initial = log_in;

    for (size_t r = 0; r < n_req; r++) {
        const zkw_log_query *request = requests + r;
        w.popped++; /* pop_and_output_intermediate_data */
        abi_t abi = precompile_abi_in_log(request);
        const int is_last_request = r == n_req - 1;
        uint32_t sha[8];
        uint64_t kst[25] = {0};
        uint8_t buf[ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE] = {0};
        size_t filled = 0;
        memcpy(sha, ORC_SHA256_IV, 32);
        size_t num_rounds;
        int needs_extra_padding_round = 0;
        size_t padding_space = 0;
        if (kind == ZKW_PRECOMPILE_SHA256) num_rounds = (size_t)abi.precompile_interpreted_data;
        else if (kind == ZKW_PRECOMPILE_ECRECOVER) num_rounds = 1;
        else {
            num_rounds = ((size_t)abi.input_memory_length + 135) / 136;
            padding_space = abi.input_memory_length % 136;
            needs_extra_padding_round = padding_space == 0;
            if (needs_extra_padding_round) num_rounds++;
        }
        if (num_rounds == 0) return -4; /* a request always has a first round carrying `new_request` */
        size_t rounds_left = num_rounds;
        /* 0 GetRequestFromQueue, 1 RunRoundFunction, 2 RunPaddingRound, 3 Finished */
        int state = 1;
        if (kind == ZKW_PRECOMPILE_KECCAK256 && abi.input_memory_length == 0 && num_rounds == 1) state = 2;

        for (size_t round = 0; round < num_rounds; round++) {
            const int is_last_round = round == num_rounds - 1;
            if (kind == ZKW_PRECOMPILE_SHA256) {
                uint8_t block[64];
                for (int k = 0; k < 2; k++) {
                    if (w.pushed >= n_q || mem_q[w.pushed].rw_flag) return -5;
                    word_be(mem_q[w.pushed].value, block + 32 * k);
                    w.pushed++; w.reads++;
                    abi.input_memory_offset += 1;
                }
                orc_sha256_compress(sha, block);
                if (g_sha_rounds) {
                    zkw_sha256_round_record *rec = g_sha_rounds + total_rounds;
                    memset(rec, 0, sizeof *rec);
                    memcpy(rec->block, block, 64);
                    rec->reset = round == 0;
                    memcpy(rec->state_after, sha, 32);
                }
                rounds_left--;
            } else if (kind == ZKW_PRECOMPILE_ECRECOVER) {
                for (int k = 0; k < 4; k++) {
                    if (w.pushed >= n_q || mem_q[w.pushed].rw_flag) return -5;
                    w.pushed++; w.reads++;
                }
                for (int k = 0; k < 2; k++) {
                    if (w.pushed >= n_q || !mem_q[w.pushed].rw_flag) return -6;
                    w.pushed++;
                }
            } else {
                const int paddings_round = needs_extra_padding_round && is_last_round;
                for (int slot = 0; slot < ZKW_KECCAK_MEMORY_READS_PER_CYCLE; slot++) {
                    const uint32_t memory_index = abi.input_memory_offset / 32, unalignment = abi.input_memory_offset % 32;
                    const uint32_t at_most = 32 - unalignment;
                    const uint32_t meaningful = abi.input_memory_length >= at_most ? at_most : abi.input_memory_length;
                    const int enough = filled + meaningful <= ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE;
                    const int should_read = meaningful != 0 && enough;
                    if (paddings_round && should_read) return -7;
                    if (!should_read) continue;
                    if (w.pushed >= n_q || mem_q[w.pushed].rw_flag) return -5;
                    if (mem_q[w.pushed].index != memory_index) return -8; /* :294 */
                    abi.input_memory_offset += meaningful;
                    abi.input_memory_length -= meaningful;
                    uint8_t be[32];
                    word_be(mem_q[w.pushed].value, be);
                    w.pushed++; w.reads++;
                    memcpy(buf + filled, be + unalignment, meaningful); /* fill_with_bytes */
                    filled += meaningful;
                }
                uint8_t block[136];
                memcpy(block, buf, 136); /* consume::<136> */
                filled = filled < 136 ? 0 : filled - 136;
                memmove(buf, buf + 136, ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE - 136);
                memset(buf + ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE - 136, 0, 136);
                if (is_last_round) {
                    if (needs_extra_padding_round) { block[0] = 0x01; block[135] = 0x80; }
                    else if (padding_space == 135) block[135] = 0x81;
                    else { block[padding_space] = 0x01; block[135] = 0x80; }
                }
                keccak_absorb_block(kst, block);
                if (keccak_rounds) {
                    zkw_keccak_round_record *rec = keccak_rounds + total_rounds;
                    memset(rec, 0, sizeof *rec);
                    memcpy(rec->block, block, 136);
                    rec->reset = round == 0;
                    for (int l = 0; l < 25; l++)
                        for (int by = 0; by < 8; by++) rec->state_after[8 * l + by] = (uint8_t)(kst[l] >> (8 * by));
                }
                const int next_round_is_padding = needs_extra_padding_round && round + 2 == num_rounds;
                if (state == 1 && next_round_is_padding) state = 2;
            }
            if (is_last_round) {
                if (kind != ZKW_PRECOMPILE_ECRECOVER) { /* the single write with the digest */
                    if (w.pushed >= n_q || !mem_q[w.pushed].rw_flag) return -6;
                    w.pushed++;
                }
                state = is_last_request ? 3 : 0;
            }
            round_counter++;
            total_rounds++;
            if (round_counter == capacity || (is_last_request && is_last_round)) {
                const int early_termination = round_counter != capacity;
                round_counter = 0;
                const int finished = is_last_request && is_last_round;
                if (finished && w.pushed != n_q) return -9; /* memory_queries_it.next().is_none() */
                zkw_precompile_fsm out;
                memset(&out, 0, sizeof out);
                if (kind != ZKW_PRECOMPILE_ECRECOVER) {
                    out.completed = state == 3;
                    out.read_words_for_round = state == 1;
                    out.read_precompile_call = state == 0;
                    out.padding_round = state == 2;
                    out.timestamp_to_use_for_read = request->timestamp;
                    out.timestamp_to_use_for_write = request->timestamp + 1;
                    out.input_page = abi.memory_page_to_read;
                    out.input_offset = abi.input_memory_offset;
                    out.output_page = abi.memory_page_to_write;
                    out.output_offset = abi.output_memory_offset;
                    if (kind == ZKW_PRECOMPILE_SHA256) {
                        out.num_rounds = (uint32_t)rounds_left;
                        memcpy(out.sha256_inner_state, early_termination ? sha_empty : sha, 32);
                    } else {
                        out.input_length = abi.input_memory_length;
                        out.needs_full_padding_round = needs_extra_padding_round;
                        out.buffer_filled = (uint32_t)filled;
                        if (early_termination) { /* :376-394: what the circuit leaves behind once all requests are done */
                            memset(buf, 0, sizeof buf);
                            memcpy(out.keccak_internal_state, keccak_empty, 200);
                        } else {
                            encode_keccak_state(kst, out.keccak_internal_state);
                        }
                        memcpy(out.buffer_bytes, buf, sizeof buf);
                    }
                }
                log_state(&w, &out.log_queue_state);
                mem_state(&w, &out.memory_queue_state);
                zkw_precompile_instance *o = instances + n_inst;
                memset(o, 0, sizeof *o);
                o->start_flag = n_inst == 0;
                o->completion_flag = finished;
                if (n_inst == 0) { o->initial_log_queue_state = log_initial; o->initial_memory_queue_state = *mem_in; }
                if (finished) o->final_memory_state = out.memory_queue_state;
                o->hidden_fsm_input = fsm_in;
                o->hidden_fsm_input.log_queue_state = log_in;
                o->hidden_fsm_input.memory_queue_state = memq_in;
                o->hidden_fsm_output = out;
                o->first_request = first_req; o->num_requests = r + 1 - first_req;
                o->first_read = first_read; o->num_reads = w.reads - first_read;
                o->first_round = first_round; o->num_rounds = total_rounds - first_round;
                first_req = r + 1; first_read = w.reads; first_round = total_rounds;
                n_inst++;
                fsm_in = out;
                log_in = out.log_queue_state;
                memq_in = out.memory_queue_state;
            }
        }
    }
    return (int64_t)n_inst;
}
