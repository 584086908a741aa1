/* decommit_sorter.c — TEST INFRASTRUCTURE: CPU restatement of compute_decommitts_sorter_circuit_snapshots
 * (src/witness/individual_circuits/sort_decommit_requests.rs:20-420), sequential like the reference. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { zkw_decommit_query q; size_t orig; } dkey;

/* sort_decommit_requests.rs:77-82: by hash (U256 numeric order), then timestamp; par_sort_by is stable */
static int dkey_cmp(const void *a, const void *b) {
    const dkey *x = (const dkey *)a, *y = (const dkey *)b;
    for (int k = 7; k >= 0; k--)
        if (x->q.hash[k] != y->q.hash[k]) return x->q.hash[k] < y->q.hash[k] ? -1 : 1;
    if (x->q.timestamp != y->q.timestamp) return x->q.timestamp < y->q.timestamp ? -1 : 1;
    return x->orig < y->orig ? -1 : (x->orig > y->orig ? 1 : 0);
}

static void qs_from(zkw_queue_state12 *s, const uint64_t *head, const uint64_t *tail, uint32_t len) {
    memset(s, 0, sizeof *s);
    if (head) memcpy(s->head, head, 96);
    if (tail) memcpy(s->tail, tail, 96);
    s->length = len;
}

int64_t orc_decommit_sorter_build(const zkw_decommit_query *q, size_t n, uint32_t capacity,
                                  const zkw_queue_state12 *dedup_in, zkw_decommit_query *sorted_q,
                                  uint64_t *unsorted_enc, uint64_t *sorted_enc, uint64_t *unsorted_tails,
                                  uint64_t *sorted_tails, zkw_decommit_query *dedup_q, uint64_t *dedup_enc,
                                  uint64_t *dedup_tails, uint64_t *n_dedup, uint64_t *challenges,
                                  uint64_t *lhs_z, uint64_t *rhs_z, zkw_decommit_sorter_instance *instances) {
    if (n == 0 || capacity == 0) return -2; /* "VM should have made some code decommits", :38-41 */
    const uint64_t zero12[12] = {0};
    const size_t num_circuits = (n + capacity - 1) / capacity; /* :54-56 */

    /* unsorted queue, :58-64 */
    orc_encode_decommit_queries(q, n, unsorted_enc);
    orc_queue_push_chain_full(unsorted_enc, n, zero12, unsorted_tails);

    /* sort, :66-82 */
    dkey *keys = (dkey *)malloc(n * sizeof *keys);
    for (size_t i = 0; i < n; i++) { keys[i].q = q[i]; keys[i].orig = i; }
    qsort(keys, n, sizeof *keys, dkey_cmp);
    for (size_t i = 0; i < n; i++) sorted_q[i] = keys[i].q;
    free(keys);

    /* self-check, :99-114 */
    for (size_t i = 1; i < n; i++) {
        if (!memcmp(sorted_q[i].hash, sorted_q[i - 1].hash, 32)) {
            if (sorted_q[i].memory_page != sorted_q[i - 1].memory_page) return -3;
            if (!(sorted_q[i].timestamp > sorted_q[i - 1].timestamp)) return -3;
        }
    }

    /* sorted queue + deduplicated queue + per-chunk snapshots, :116-172 */
    orc_encode_decommit_queries(sorted_q, n, sorted_enc);
    orc_queue_push_chain_full(sorted_enc, n, zero12, sorted_tails);

    zkw_queue_state12 dedup_state; /* the deduplicated simulator's (head, tail, num_items) */
    if (dedup_in) dedup_state = *dedup_in; else qs_from(&dedup_state, NULL, NULL, 0);
    zkw_queue_state12 previous_dedup_state = dedup_state;
    uint32_t first_encountered_timestamp = 0;
    size_t nd = 0, counter = 0, chunk = 0;
    zkw_queue_state12 *dedup_snap = (zkw_queue_state12 *)calloc(num_circuits, sizeof(zkw_queue_state12));
    uint32_t (*prev_keys)[9] = calloc(num_circuits, sizeof *prev_keys);
    zkw_decommit_query *prev_recs = (zkw_decommit_query *)calloc(num_circuits, sizeof *prev_recs);
    uint32_t *first_ts = (uint32_t *)calloc(num_circuits, 4);
    for (size_t idx = 0; idx < n; idx++) {
        const zkw_decommit_query *qq = sorted_q + idx;
        const int last = idx == n - 1;
        if (qq->is_fresh) {
            first_encountered_timestamp = qq->timestamp;
            previous_dedup_state = dedup_state;
            dedup_q[nd] = *qq;
            orc_encode_decommit_queries(qq, 1, dedup_enc + 8 * nd);
            orc_queue_push_chain_full(dedup_enc + 8 * nd, 1, dedup_state.tail, dedup_tails + 12 * nd);
            memcpy(dedup_state.tail, dedup_tails + 12 * nd, 96);
            dedup_state.length += 1;
            nd++;
        }
        counter++;
        if (counter == capacity) {
            counter = 0;
            dedup_snap[chunk] = last ? dedup_state : previous_dedup_state;
            prev_keys[chunk][0] = qq->timestamp; /* concatenate_key, :422-435 */
            memcpy(&prev_keys[chunk][1], qq->hash, 32);
            prev_recs[chunk] = *qq;
            prev_recs[chunk].decommitted_length = 0; /* DecommitQueryWitness has no length field */
            first_ts[chunk] = first_encountered_timestamp;
            chunk++;
        }
    }
    if (counter > 0) { /* :174-181: partial last chunk -> placeholders */
        dedup_snap[chunk] = dedup_state;
        chunk++;
    }
    *n_dedup = nd;

    /* challenges and chains, :205-243 */
    const uint64_t *u_final = unsorted_tails + 12 * (n - 1), *s_final = sorted_tails + 12 * (n - 1);
    orc_fs_challenges(u_final, (uint32_t)n, s_final, (uint32_t)n, 12, 9, challenges);
    for (int rep = 0; rep < 2; rep++)
        if (orc_grand_product_chains(unsorted_enc, sorted_enc, n, 8, challenges + 9 * rep, lhs_z + rep * n,
                                     rhs_z + rep * n) != 0) {
            free(dedup_snap); free(prev_keys); free(prev_recs); free(first_ts);
            return -4;
        }

    /* instances, :245-420 */
    for (size_t i = 0; i < num_circuits; i++) {
        zkw_decommit_sorter_instance *w = instances + i;
        memset(w, 0, sizeof *w);
        const size_t lo = i * capacity, hi = lo + capacity < n ? lo + capacity : n, last = hi - 1;
        w->start_flag = i == 0;
        w->completion_flag = i == num_circuits - 1;
        w->first_item = lo;
        w->num_items = hi - lo;
        qs_from(&w->initial_queue_state, NULL, u_final, (uint32_t)n);
        qs_from(&w->sorted_queue_initial_state, NULL, s_final, (uint32_t)n);
        if (i == num_circuits - 1) w->final_queue_state = dedup_state; /* output_passthrough_data */
        if (i > 0) w->hidden_fsm_input = instances[i - 1].hidden_fsm_output; /* placeholder (zeros) for i = 0 */
        zkw_decommit_sorter_fsm *fo = &w->hidden_fsm_output;
        /* queue states after POPPING the chunk from the full simulators, :262-290, 309-337 */
        qs_from(&fo->initial_queue_state, unsorted_tails + 12 * last, u_final, (uint32_t)(n - hi));
        qs_from(&fo->sorted_queue_state, sorted_tails + 12 * last, s_final, (uint32_t)(n - hi));
        fo->final_queue_state = dedup_snap[i];
        for (int rep = 0; rep < 2; rep++) {
            fo->lhs_accumulator[rep] = lhs_z[rep * n + last];
            fo->rhs_accumulator[rep] = rhs_z[rep * n + last];
        }
        memcpy(fo->previous_packed_key, prev_keys[i], 36);
        fo->previous_record = prev_recs[i];
        fo->first_encountered_timestamp = first_ts[i];
    }
    free(dedup_snap); free(prev_keys); free(prev_recs); free(first_ts);
    return (int64_t)num_circuits;
}
