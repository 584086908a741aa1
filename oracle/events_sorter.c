/* events_sorter.c — TEST INFRASTRUCTURE: CPU restatement of compute_events_dedup_and_sort and
 * sort_and_dedup_events_log (src/witness/individual_circuits/events_sort_dedup.rs:16-580), sequential
 * like the reference (the per-chunk loop replays the circuit's pops/pushes item by item). */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { zkw_log_query q; size_t orig; } ekey;

/* events_sort_dedup.rs:81-90: by timestamp; on equal timestamps the rollback goes after its forward
   twin. The reference comparator is not a strict weak order for other ties; well-formed queues
   have at most one forward and one rollback per timestamp, for which it equals this stable order. */
static int ekey_cmp(const void *a, const void *b) {
    const ekey *x = (const ekey *)a, *y = (const ekey *)b;
    if (x->q.timestamp != y->q.timestamp) return x->q.timestamp < y->q.timestamp ? -1 : 1;
    if (x->q.rollback != y->q.rollback) return x->q.rollback < y->q.rollback ? -1 : 1;
    return x->orig < y->orig ? -1 : (x->orig > y->orig ? 1 : 0);
}

static zkw_log_query normalized(const zkw_log_query *p) { /* :541-553 */
    zkw_log_query r;
    memset(&r, 0, sizeof r);
    r.tx_number_in_block = p->tx_number_in_block;
    r.shard_id = p->shard_id;
    memcpy(r.address, p->address, sizeof r.address);
    memcpy(r.key, p->key, sizeof r.key);
    memcpy(r.written_value, p->written_value, sizeof r.written_value);
    r.is_service = p->is_service;
    return r;
}

static void qs4(zkw_queue_state4 *s, const uint64_t *head, const uint64_t *tail, uint32_t len) {
    memset(s, 0, sizeof *s);
    if (head) memcpy(s->head, head, 32);
    if (tail) memcpy(s->tail, tail, 32);
    s->length = len;
}

int64_t orc_events_sorter_build(const zkw_log_query *q, size_t n, uint32_t capacity, const zkw_queue_state4 *result_in,
                                zkw_log_query *sorted_q, uint64_t *unsorted_enc, uint64_t *sorted_enc,
                                uint64_t *unsorted_old_tails, uint64_t *unsorted_new_tails,
                                uint64_t *sorted_old_tails, uint64_t *sorted_new_tails, zkw_log_query *result_q,
                                uint64_t *result_enc, uint64_t *result_new_tails, uint64_t *n_result,
                                uint64_t *challenges, uint64_t *lhs_z, uint64_t *rhs_z,
                                zkw_events_sorter_instance *instances) {
    if (capacity == 0) return -2;
    zkw_queue_state4 result_state;
    if (result_in) result_state = *result_in; else qs4(&result_state, NULL, NULL, 0);
    *n_result = 0;
    if (n == 0) { /* :27-76: single dummy witness, accumulators forced to ONE */
        zkw_events_sorter_instance *w = instances;
        memset(w, 0, sizeof *w);
        w->start_flag = w->completion_flag = 1;
        for (int r = 0; r < 2; r++) {
            w->hidden_fsm_input.lhs_accumulator[r] = w->hidden_fsm_input.rhs_accumulator[r] = 1;
            w->hidden_fsm_output.lhs_accumulator[r] = w->hidden_fsm_output.rhs_accumulator[r] = 1;
        }
        { /* the circuit derives its challenges whatever the queue holds (the trace's closed-form section does): those of two empty queues */
            const uint64_t z4[4] = {0};
            orc_fs_challenges(z4, 0, z4, 0, 4, 21, challenges);
        }
        return 1;
    }
    const uint64_t zero4[4] = {0};
    /* the unsorted queue as the demuxer left it (LogQueue.simulator / .states) */
    orc_encode_log_queries(q, n, NULL, unsorted_enc);
    orc_queue_push_chain_log(unsorted_enc, n, zero4, unsorted_old_tails, unsorted_new_tails);

    ekey *keys = (ekey *)malloc(n * sizeof *keys);
    for (size_t i = 0; i < n; i++) { keys[i].q = q[i]; keys[i].orig = i; }
    qsort(keys, n, sizeof *keys, ekey_cmp);
    for (size_t i = 0; i < n; i++) sorted_q[i] = keys[i].q;
    free(keys);

    /* intermediate sorted queue, :92-99 */
    orc_encode_log_queries(sorted_q, n, NULL, sorted_enc);
    orc_queue_push_chain_log(sorted_enc, n, zero4, sorted_old_tails, sorted_new_tails);

    /* sort_and_dedup_events_log, :508-580 */
    size_t nd = 0;
    {
        int have = 0;
        zkw_log_query top;
        for (size_t i = 0; i < n; i++) {
            const zkw_log_query *el = sorted_q + i;
            if (el->shard_id != 0) return -3; /* "only rollup shard is supported" */
            if (!have) {
                if (el->rollback) return -4;
                top = *el; have = 1;
            } else {
                zkw_log_query previous = top;
                have = 0;
                if (previous.timestamp == el->timestamp) {
                    if (previous.rollback || !el->rollback || !previous.rw_flag || !el->rw_flag ||
                        previous.tx_number_in_block != el->tx_number_in_block ||
                        memcmp(previous.address, el->address, 20) || memcmp(previous.key, el->key, 32) ||
                        memcmp(previous.written_value, el->written_value, 32) || previous.is_service != el->is_service)
                        return -5;
                    continue; /* rolled back: both dropped */
                }
                if (el->rollback) return -6;
                top = *el; have = 1;
                result_q[nd++] = normalized(&previous);
            }
        }
        if (have) result_q[nd++] = normalized(&top);
    }
    orc_encode_log_queries(result_q, nd, NULL, result_enc);

    const uint64_t *u_final = unsorted_new_tails + 4 * (n - 1), *s_final = sorted_new_tails + 4 * (n - 1);
    /* challenges N = 4, 21 per repetition, :104-116 */
    orc_fs_challenges(u_final, (uint32_t)n, s_final, (uint32_t)n, 4, 21, challenges);
    for (int rep = 0; rep < 2; rep++)
        if (orc_grand_product_chains(unsorted_enc, sorted_enc, n, 20, challenges + 21 * rep, lhs_z + rep * n,
                                     rhs_z + rep * n) != 0)
            return -7;

    /* per-chunk simulation, :232-490 */
    const size_t num_circuits = (n + capacity - 1) / capacity;
    uint32_t previous_key = 0;
    zkw_log_query previous_item;
    memset(&previous_item, 0, sizeof previous_item);
    zkw_queue_state4 cur_unsorted, cur_sorted, cur_result = result_state;
    qs4(&cur_unsorted, NULL, NULL, 0);
    qs4(&cur_sorted, NULL, NULL, 0);
    uint64_t cur_lhs[2] = {1, 1}, cur_rhs[2] = {1, 1};
    size_t it = 0; /* deduplicated_queries_it */
    for (size_t idx = 0; idx < num_circuits; idx++) {
        const size_t lo = idx * capacity, hi = lo + capacity < n ? lo + capacity : n, last = hi - 1;
        const int is_first = idx == 0, is_last = idx == num_circuits - 1;
        uint32_t new_last_key = previous_key;
        zkw_log_query new_last_item = previous_item;
        uint32_t current_timestamp = previous_item.timestamp;
        int exhausted = 0;
        for (size_t t = lo; t < hi; t++) {
            const zkw_log_query *item = sorted_q + t;
            const int first_ever = (t == lo) && is_first, is_last_ever = (t == hi - 1) && is_last;
            if (!first_ever) {
                if (!item->rw_flag) return -8;
                if (current_timestamp == item->timestamp) {
                    if (!item->rollback) return -9;
                } else {
                    if (item->rollback) return -10;
                    if (!new_last_item.rollback) {
                        if (it < nd) {
                            const zkw_log_query *nq = result_q + it;
                            if (memcmp(nq->address, new_last_item.address, 20) || memcmp(nq->key, new_last_item.key, 32) ||
                                memcmp(nq->written_value, new_last_item.written_value, 32)) return -11;
                            orc_queue_push_chain_log(result_enc + 20 * it, 1, result_state.tail, NULL, result_new_tails + 4 * it);
                            memcpy(result_state.tail, result_new_tails + 4 * it, 32);
                            result_state.length++;
                            it++;
                        } else {
                            if (!is_last || exhausted) return -12;
                            exhausted = 1;
                        }
                    }
                }
            }
            new_last_key = item->timestamp;
            new_last_item = *item;
            current_timestamp = item->timestamp;
            if (is_last_ever && !exhausted && !new_last_item.rollback) {
                if (it >= nd) return -13;
                orc_queue_push_chain_log(result_enc + 20 * it, 1, result_state.tail, NULL, result_new_tails + 4 * it);
                memcpy(result_state.tail, result_new_tails + 4 * it, 32);
                result_state.length++;
                it++;
            }
        }
        zkw_events_sorter_instance *w = instances + idx;
        memset(w, 0, sizeof *w);
        w->start_flag = is_first; w->completion_flag = is_last;
        w->first_item = lo; w->num_items = hi - lo;
        qs4(&w->initial_log_queue_state, NULL, u_final, (uint32_t)n);
        qs4(&w->intermediate_sorted_queue_state, NULL, s_final, (uint32_t)n);
        zkw_events_sorter_fsm *fi = &w->hidden_fsm_input, *fo = &w->hidden_fsm_output;
        memcpy(fi->lhs_accumulator, cur_lhs, 16); memcpy(fi->rhs_accumulator, cur_rhs, 16);
        fi->initial_unsorted_queue_state = cur_unsorted;
        fi->intermediate_sorted_queue_state = cur_sorted;
        fi->final_result_queue_state = cur_result;
        fi->previous_key = previous_key;
        fi->previous_item = previous_item;
        for (int r = 0; r < 2; r++) { fo->lhs_accumulator[r] = lhs_z[r * n + last]; fo->rhs_accumulator[r] = rhs_z[r * n + last]; }
        qs4(&fo->initial_unsorted_queue_state, unsorted_new_tails + 4 * last, u_final, (uint32_t)(n - hi));
        qs4(&fo->intermediate_sorted_queue_state, sorted_new_tails + 4 * last, s_final, (uint32_t)(n - hi));
        fo->final_result_queue_state = result_state;
        fo->previous_key = new_last_key;
        fo->previous_item = new_last_item;
        if ((hi - lo) % capacity != 0) { /* :469-479 */
            fo->previous_key = 0;
            memset(&fo->previous_item, 0, sizeof fo->previous_item);
        }
        memcpy(cur_lhs, fo->lhs_accumulator, 16); memcpy(cur_rhs, fo->rhs_accumulator, 16);
        previous_key = new_last_key; previous_item = new_last_item;
        cur_result = result_state;
        cur_unsorted = fo->initial_unsorted_queue_state;
        cur_sorted = fo->intermediate_sorted_queue_state;
    }
    if (it != nd) return -14; /* :494 */
    instances[num_circuits - 1].final_queue_state = result_state; /* :496-503 */
    *n_result = nd;
    return (int64_t)num_circuits;
}
