/* decommitter.c — TEST INFRASTRUCTURE: CPU restatement of compute_decommitter_circuit_snapshots
 * (src/witness/individual_circuits/decommit_code.rs:20-439), sequential like the reference. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

extern const uint32_t ORC_SHA256_IV[8];
void orc_sha256_compress(uint32_t state[8], const uint8_t block[64]);

static void word_be(const uint32_t *w, uint8_t out[32]) { /* U256::to_big_endian */
    for (int k = 0; k < 8; k++) {
        uint32_t limb = w[7 - k];
        out[4 * k] = (uint8_t)(limb >> 24); out[4 * k + 1] = (uint8_t)(limb >> 16);
        out[4 * k + 2] = (uint8_t)(limb >> 8); out[4 * k + 3] = (uint8_t)limb;
    }
}

void orc_bytecode_hash(const uint32_t *words, size_t n_words, uint32_t top_limb, uint32_t hash_out[8]) {
    uint32_t st[8];
    memcpy(st, ORC_SHA256_IV, 32);
    const size_t rounds = (n_words + 1) / 2;
    for (size_t r = 0; r < rounds; r++) {
        uint8_t block[64] = {0};
        word_be(words + 8 * (2 * r), block);
        if (2 * r + 1 < n_words) word_be(words + 8 * (2 * r + 1), block + 32);
        else {
            block[32] = 0x80;
            uint32_t bits = (uint32_t)(n_words * 32 * 8);
            block[60] = (uint8_t)(bits >> 24); block[61] = (uint8_t)(bits >> 16); block[62] = (uint8_t)(bits >> 8); block[63] = (uint8_t)bits;
        }
        orc_sha256_compress(st, block);
    }
    for (int j = 1; j < 8; j++) hash_out[7 - j] = st[j]; /* BE word j of the digest = LE limb 7-j */
    hash_out[7] = top_limb;
}

static void qs12(zkw_queue_state12 *s, const uint64_t *head, const uint64_t *tail, uint32_t len) {
    memset(s, 0, sizeof *s);
    if (head) memcpy(s->head, head, 96);
    if (tail) memcpy(s->tail, tail, 96);
    s->length = len;
}

/* may be NULL (set before orc_decommitter_build; test infrastructure, single-threaded use): one record per SHA-256 round */
static zkw_sha256_round_record *g_dcm_sha_rounds = NULL;
void orc_decommitter_set_sha256_rounds(zkw_sha256_round_record *r) { g_dcm_sha_rounds = r; }
int64_t orc_decommitter_build(const zkw_decommit_query *requests, const uint64_t *dedup_tails, size_t n_requests,
                              const uint32_t *words, const uint64_t *word_offsets, uint32_t capacity,
                              const zkw_queue_state12 *mem_in, zkw_mem_query *mem_q, uint64_t *mem_enc,
                              uint64_t *mem_tails, uint32_t *round_states, zkw_decommitter_instance *instances) {
    if (n_requests == 0 || capacity == 0) return -2;
    const size_t total_words = word_offsets[n_requests] - word_offsets[0];
    /* :47-78: every code word becomes a write into the request's code page, appended to the memory queue */
    size_t wpos = 0;
    for (size_t k = 0; k < n_requests; k++) {
        if (!requests[k].is_fresh) return -3;
        const size_t nw = word_offsets[k + 1] - word_offsets[k];
        if (nw == 0) return -4;
        for (size_t idx = 0; idx < nw; idx++, wpos++) {
            zkw_mem_query *m = mem_q + wpos;
            memset(m, 0, sizeof *m);
            m->timestamp = requests[k].timestamp;
            m->page = requests[k].memory_page;
            m->index = (uint32_t)idx;
            m->rw_flag = 1;
            memcpy(m->value, words + 8 * (word_offsets[k] - word_offsets[0] + idx), 32);
        }
    }
    orc_encode_memory_queries(mem_q, total_words, mem_enc);
    orc_queue_push_chain_full(mem_enc, total_words, mem_in->tail, mem_tails);

    const uint64_t *dedup_final = dedup_tails + 12 * (n_requests - 1);
    zkw_decommitter_fsm fsm; /* fsm_internals + the two queue states, carried across instances */
    memset(&fsm, 0, sizeof fsm);
    uint32_t sha[8] = {0};
    size_t req = 0, words_done = 0, rounds_left = 0, word_in_req = 0, round_g = 0, popped = 0, inst = 0;
    int state = 0; /* 0 BeginNew, 1 DecommitMore, 2 Done */
    int start = 1;
    zkw_decommitter_fsm prev_out;
    memset(&prev_out, 0, sizeof prev_out);
    for (;;) {
        zkw_decommitter_instance *w = instances + inst;
        memset(w, 0, sizeof *w);
        w->start_flag = start;
        w->first_round = round_g; w->first_request = popped; w->first_word = words_done;
        /* memory queue state BEFORE this instance's words: all_memory_queue_states[start_idx + offset - 1], :161-170 */
        zkw_queue_state12 mq;
        qs12(&mq, mem_in->head, words_done ? mem_tails + 12 * (words_done - 1) : mem_in->tail, mem_in->length + (uint32_t)words_done);
        w->hidden_fsm_input = prev_out; /* placeholder for the first instance */
        w->hidden_fsm_input.memory_queue_state = mq;
        if (start) {
            start = 0;
            w->memory_queue_initial_state = *mem_in;
            qs12(&w->sorted_requests_queue_initial_state, NULL, dedup_final, (uint32_t)n_requests);
        }
        for (uint32_t cyc = 0; cyc < capacity; cyc++) {
            if (state == 0) { /* BeginNew, :228-283 */
                const zkw_decommit_query *q = requests + req;
                memcpy(sha, ORC_SHA256_IV, 32);
                popped++;
                const uint32_t num_words = q->hash[7] & 0xFFFF; /* (hash.0[3] >> 32) as u16 : low half of the top limb */
                if (!(num_words & 1)) return -6;
                if (num_words != word_offsets[req + 1] - word_offsets[req]) return -7;
                rounds_left = ((size_t)num_words + 1) / 2;
                fsm.state_get_from_queue = 0; fsm.state_decommit = 1;
                fsm.num_rounds_left = (uint32_t)rounds_left;
                memcpy(fsm.sha256_inner_state, ORC_SHA256_IV, 32);
                fsm.current_index = 0; fsm.current_page = q->memory_page; fsm.timestamp = q->timestamp;
                fsm.length_in_bits = num_words * 32 * 8;
                memcpy(fsm.hash_to_compare_against, q->hash, 32);
                fsm.hash_to_compare_against[7] &= 0; /* the 4 most significant bytes zeroed, :272-277 */
                word_in_req = 0;
                state = 1;
            }
            /* DecommitMore, :285-350 */
            uint8_t block[64] = {0};
            fsm.num_rounds_left--; rounds_left--;
            const uint32_t *base = words + 8 * (word_offsets[req] - word_offsets[0]);
            word_be(base + 8 * word_in_req, block);
            word_in_req++; words_done++; fsm.current_index++;
            if (rounds_left != 0) {
                word_be(base + 8 * word_in_req, block + 32);
                word_in_req++; words_done++; fsm.current_index++;
            } else {
                block[32] = 0x80;
                const uint32_t bits = fsm.length_in_bits;
                block[60] = (uint8_t)(bits >> 24); block[61] = (uint8_t)(bits >> 16); block[62] = (uint8_t)(bits >> 8); block[63] = (uint8_t)bits;
            }
            orc_sha256_compress(sha, block);
            memcpy(round_states + 8 * round_g, sha, 32);
            if (g_dcm_sha_rounds) { /* the cycle of the CodeDecommitter circuit */
                zkw_sha256_round_record *rec = g_dcm_sha_rounds + round_g;
                memset(rec, 0, sizeof *rec);
                memcpy(rec->block, block, 64);
                rec->reset = (size_t)fsm.num_rounds_left + 1 == ((size_t)(requests[req].hash[7] & 0xFFFF) + 1) / 2;
                memcpy(rec->state_after, sha, 32);
            }
            round_g++;
            if (rounds_left == 0) {
                for (int j = 1; j < 8; j++)
                    if (sha[j] != fsm.hash_to_compare_against[7 - j]) return -5;
                if (fsm.hash_to_compare_against[7] != 0) return -5;
                if (req + 1 == n_requests) {
                    state = 2;
                    fsm.state_get_from_queue = 0; fsm.state_decommit = 0; fsm.finished = 1;
                } else {
                    state = 0; req++;
                    fsm.state_get_from_queue = 1; fsm.state_decommit = 0;
                }
            }
            if (state == 2) break;
        }
        memcpy(fsm.sha256_inner_state, sha, 32); /* :352-359 */
        qs12(&fsm.decommittment_requests_queue_state, dedup_tails + 12 * (popped - 1), dedup_final, (uint32_t)(n_requests - popped));
        qs12(&fsm.memory_queue_state, mem_in->head, mem_tails + 12 * (words_done - 1), mem_in->length + (uint32_t)words_done);
        w->hidden_fsm_output = fsm;
        w->num_rounds = round_g - w->first_round;
        w->num_requests = popped - w->first_request;
        w->num_words = words_done - w->first_word;
        prev_out = fsm;
        inst++;
        if (state == 2) {
            w->completion_flag = 1;
            w->memory_queue_final_state = fsm.memory_queue_state;
            break;
        }
    }
    if (words_done != total_words) return -8;
    return (int64_t)inst;
}
