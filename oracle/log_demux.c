/* log_demux.c — TEST INFRASTRUCTURE: CPU restatement of compute_logs_demux
 * (src/witness/individual_circuits/log_demux.rs:20-388), sequential like the reference. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* log_demux.rs:174-249: -1 = not routed (other precompile addresses), -2 = unreachable!() in the reference */
static int route(const zkw_log_query *q, const zkw_demux_params *p) {
    if (q->aux_byte == p->storage_aux_byte) return q->shard_id == 0 ? ZKW_DEMUX_STORAGE : -2;
    if (q->aux_byte == p->l1_message_aux_byte) return ZKW_DEMUX_L1_MESSAGES;
    if (q->aux_byte == p->event_aux_byte) return ZKW_DEMUX_EVENTS;
    if (q->aux_byte == p->precompile_aux_byte) {
        if (q->rollback) return -2; /* assert!(!query.rollback) */
        int high_zero = !(q->address[1] | q->address[2] | q->address[3] | q->address[4]);
        if (high_zero && q->address[0] == p->keccak256_address) return ZKW_DEMUX_KECCAK256;
        if (high_zero && q->address[0] == p->sha256_address) return ZKW_DEMUX_SHA256;
        if (high_zero && q->address[0] == p->ecrecover_address) return ZKW_DEMUX_ECRECOVER;
        return -1;
    }
    return -2;
}

static void qs4(zkw_queue_state4 *s, const uint64_t *head, const uint64_t *tail, uint32_t len) {
    memset(s, 0, sizeof *s);
    if (head) memcpy(s->head, head, 32);
    if (tail) memcpy(s->tail, tail, 32);
    s->length = len;
}

int64_t orc_log_demux_build(const zkw_log_query *q, size_t n, uint32_t capacity, const zkw_demux_params *params,
                            uint64_t *in_enc, uint64_t *in_old_tails, uint64_t *in_new_tails, zkw_log_query *out_q,
                            uint64_t *out_enc, uint64_t *out_old_tails, uint64_t *out_new_tails, uint64_t *out_offsets,
                            zkw_log_demux_instance *instances) {
    if (capacity == 0) return -2;
    memset(out_offsets, 0, 7 * sizeof(uint64_t));
    if (n == 0) { /* :51-107: one all-placeholder instance */
        memset(instances, 0, sizeof *instances);
        instances->start_flag = instances->completion_flag = 1;
        return 1;
    }
    const uint64_t zero4[4] = {0};
    /* the original log queue as built in src/witness/oracle.rs:308-350 */
    orc_encode_log_queries(q, n, NULL, in_enc);
    orc_queue_push_chain_log(in_enc, n, zero4, in_old_tails, in_new_tails);

    /* queue sizes and offsets */
    size_t cnt[6] = {0};
    for (size_t i = 0; i < n; i++) {
        int r = route(q + i, params);
        if (r == -2) return -3;
        if (r >= 0) cnt[r]++;
    }
    for (int k = 0; k < 6; k++) out_offsets[k + 1] = out_offsets[k] + cnt[k];

    const size_t num_chunks = (n + capacity - 1) / capacity;
    size_t filled[6] = {0};
    uint64_t tails[6][4];
    memset(tails, 0, sizeof tails);
    for (size_t idx = 0; idx < num_chunks; idx++) {
        const size_t lo = idx * capacity, hi = lo + capacity < n ? lo + capacity : n, last = hi - 1;
        for (size_t t = lo; t < hi; t++) { /* :171-251 */
            int r = route(q + t, params);
            if (r < 0) continue;
            const size_t dst = out_offsets[r] + filled[r];
            out_q[dst] = q[t];
            orc_encode_log_queries(q + t, 1, NULL, out_enc + 20 * dst);
            orc_queue_push_chain_log(out_enc + 20 * dst, 1, tails[r], out_old_tails + 4 * dst, out_new_tails + 4 * dst);
            memcpy(tails[r], out_new_tails + 4 * dst, 32);
            filled[r]++;
        }
        zkw_log_demux_instance *w = instances + idx;
        memset(w, 0, sizeof *w);
        w->start_flag = idx == 0;
        w->completion_flag = idx == num_chunks - 1;
        w->first_item = lo;
        w->num_items = hi - lo;
        qs4(&w->initial_log_queue_state, NULL, in_new_tails + 4 * (n - 1), (uint32_t)n);
        zkw_log_demux_fsm *fo = &w->hidden_fsm_output;
        /* :275-281: head := tail after the chunk, tail := full tail, length := remaining */
        qs4(&fo->initial_log_queue_state, in_new_tails + 4 * last, in_new_tails + 4 * (n - 1), (uint32_t)(n - hi));
        for (int k = 0; k < 6; k++) qs4(&fo->queue_state[k], NULL, tails[k], (uint32_t)filled[k]);
        if (idx > 0) w->hidden_fsm_input = instances[idx - 1].hidden_fsm_output;
        if (idx == num_chunks - 1)
            for (int k = 0; k < 6; k++) w->output_queue_state[k] = fo->queue_state[k];
    }
    for (int k = 0; k < 6; k++)
        if (filled[k] != cnt[k]) return -4;
    return (int64_t)num_chunks;
}

/* ---- linear hasher (data_hasher_and_merklizer.rs:8-67) */
void orc_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]);

void orc_serialize_l1_message(const zkw_log_query *q, uint8_t out[88]) {
    int o = 0;
    out[o++] = q->shard_id;
    out[o++] = q->is_service ? 1 : 0;
    out[o++] = (uint8_t)(q->tx_number_in_block >> 8);
    out[o++] = (uint8_t)q->tx_number_in_block;
    for (int k = 4; k >= 0; k--) { uint32_t l = q->address[k]; out[o++] = (uint8_t)(l >> 24); out[o++] = (uint8_t)(l >> 16); out[o++] = (uint8_t)(l >> 8); out[o++] = (uint8_t)l; }
    for (int k = 7; k >= 0; k--) { uint32_t l = q->key[k]; out[o++] = (uint8_t)(l >> 24); out[o++] = (uint8_t)(l >> 16); out[o++] = (uint8_t)(l >> 8); out[o++] = (uint8_t)l; }
    for (int k = 7; k >= 0; k--) { uint32_t l = q->written_value[k]; out[o++] = (uint8_t)(l >> 24); out[o++] = (uint8_t)(l >> 16); out[o++] = (uint8_t)(l >> 8); out[o++] = (uint8_t)l; }
}

void orc_linear_keccak256(const zkw_log_query *q, size_t n, uint8_t hash_out[32]) {
    uint8_t *buf = (uint8_t *)malloc(n * 88 + 1);
    for (size_t i = 0; i < n; i++) orc_serialize_l1_message(q + i, buf + 88 * i);
    orc_keccak256(buf, n * 88, hash_out);
    free(buf);
}
