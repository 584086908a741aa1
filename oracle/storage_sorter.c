/* storage_sorter.c — TEST INFRASTRUCTURE: CPU restatement of sort_storage_access_queries
 * (circuit_sequencer_api/src/sort_storage_access.rs:19-260) and compute_storage_dedup_and_sort
 * (src/witness/individual_circuits/storage_sort_dedup.rs:12-703), sequential like the reference. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct { zkw_log_query q; uint32_t ext; } skey;

/* sort_storage_access.rs:31-42: (shard compared with ITSELF: always Equal), address, key,
   extended timestamp. H160 / U256 order = numeric order = most significant limb first. */
static int skey_cmp(const void *a, const void *b) {
    const skey *x = (const skey *)a, *y = (const skey *)b;
    for (int k = 4; k >= 0; k--)
        if (x->q.address[k] != y->q.address[k]) return x->q.address[k] < y->q.address[k] ? -1 : 1;
    for (int k = 7; k >= 0; k--)
        if (x->q.key[k] != y->q.key[k]) return x->q.key[k] < y->q.key[k] ? -1 : 1;
    return x->ext < y->ext ? -1 : (x->ext > y->ext ? 1 : 0);
}

static int same_cell(const zkw_log_query *a, const zkw_log_query *b) {
    return a->shard_id == b->shard_id && !memcmp(a->address, b->address, 20) && !memcmp(a->key, b->key, 32);
}

/* L::create_partially_filled_from_fields, log_query.rs:49-72 */
static zkw_log_query partial(const zkw_log_query *c, const uint32_t *read, const uint32_t *written, int rw) {
    zkw_log_query r;
    memset(&r, 0, sizeof r);
    r.shard_id = c->shard_id;
    memcpy(r.address, c->address, 20);
    memcpy(r.key, c->key, 32);
    memcpy(r.read_value, read, 32);
    memcpy(r.written_value, written, 32);
    r.rw_flag = rw ? 1 : 0;
    return r;
}

static void qs4(zkw_queue_state4 *s, const uint64_t *head, const uint64_t *tail, uint32_t len) {
    memset(s, 0, sizeof *s);
    if (head) memcpy(s->head, head, 32);
    if (tail) memcpy(s->tail, tail, 32);
    s->length = len;
}

int64_t orc_storage_sorter_build(const zkw_log_query *q, size_t n, uint32_t capacity, zkw_log_query *sorted_q,
                                 uint32_t *sorted_ext_ts, uint64_t *unsorted_enc, uint64_t *lhs_enc, uint64_t *sorted_enc,
                                 uint64_t *unsorted_old_tails, uint64_t *unsorted_new_tails, uint64_t *sorted_old_tails,
                                 uint64_t *sorted_new_tails, zkw_log_query *result_q, uint64_t *result_enc,
                                 uint64_t *result_new_tails, uint64_t *n_result, uint64_t *challenges, uint64_t *lhs_z,
                                 uint64_t *rhs_z, zkw_storage_sorter_instance *instances) {
    if (capacity == 0) return -2;
    *n_result = 0;
    if (n == 0) { /* storage_sort_dedup.rs:23-70 */
        zkw_storage_sorter_instance *w = instances;
        memset(w, 0, sizeof *w);
        w->start_flag = w->completion_flag = 1;
        for (int r = 0; r < 2; r++) w->hidden_fsm_output.lhs_accumulator[r] = w->hidden_fsm_output.rhs_accumulator[r] = 1;
        w->hidden_fsm_output.cycle_idx = 4; /* the reference's hack, :45 */
        { /* the circuit derives its challenges whatever the queue holds (the trace's closed-form section does): those of two empty queues */
            const uint64_t z4[4] = {0};
            orc_fs_challenges(z4, 0, z4, 0, 4, 21, challenges);
        }
        return 1;
    }
    const uint64_t zero4[4] = {0};
    /* the demuxed storage queue (plain encodings) */
    orc_encode_log_queries(q, n, NULL, unsorted_enc);
    orc_queue_push_chain_log(unsorted_enc, n, zero4, unsorted_old_tails, unsorted_new_tails);

    /* ---- sort_storage_access_queries */
    skey *keys = (skey *)malloc(n * sizeof *keys);
    for (size_t i = 0; i < n; i++) { keys[i].q = q[i]; keys[i].ext = (uint32_t)i; }
    qsort(keys, n, sizeof *keys, skey_cmp);
    for (size_t i = 0; i < n; i++) { sorted_q[i] = keys[i].q; sorted_ext_ts[i] = keys[i].ext; }
    free(keys);
    size_t nd = 0;
    for (size_t s = 0; s < n;) {
        size_t e = s;
        while (e < n && same_cell(sorted_q + e, sorted_q + s)) e++;
        uint32_t initial[8], current[8];
        int have = 0, did_read_d0 = 0;
        size_t stack_cap = e - s, sp = 0;
        size_t *stack = (size_t *)malloc(stack_cap * sizeof(size_t));
        for (size_t t = s; t < e; t++) {
            const zkw_log_query *el = sorted_q + t;
            if (!have) { if (!el->rw_flag) did_read_d0 = 1; }
            else if (!el->rw_flag && sp == 0) did_read_d0 = 1;
            if (!have) {
                if (el->rw_flag && el->rollback) { free(stack); return -3; }
                memcpy(initial, el->read_value, 32);
                memcpy(current, el->read_value, 32);
                have = 1;
            }
            if (!el->rw_flag) {
                if (memcmp(el->read_value, current, 32)) { free(stack); return -4; }
            } else if (!el->rollback) {
                if (memcmp(el->read_value, current, 32)) { free(stack); return -5; }
                memcpy(current, el->written_value, 32);
                stack[sp++] = t;
            } else {
                if (sp == 0) { free(stack); return -6; }
                const zkw_log_query *pc = sorted_q + stack[--sp];
                if (memcmp(el->read_value, pc->read_value, 32) || memcmp(el->written_value, pc->written_value, 32) ||
                    memcmp(el->written_value, current, 32)) { free(stack); return -7; }
                memcpy(current, el->read_value, 32);
            }
        }
        const int stack_empty = sp == 0;
        free(stack);
        if (!did_read_d0 && stack_empty) {
            if (memcmp(initial, current, 32)) return -8;
        } else if (!memcmp(initial, current, 32)) {
            if (did_read_d0 || !stack_empty) result_q[nd++] = partial(sorted_q + s, initial, current, 0);
        } else {
            result_q[nd++] = partial(sorted_q + s, initial, current, 1);
        }
        s = e;
    }
    orc_encode_log_queries(result_q, nd, NULL, result_enc);

    /* ---- compute_storage_dedup_and_sort */
    orc_encode_log_queries(sorted_q, n, sorted_ext_ts, sorted_enc); /* LogWithExtendedEnumerationQueueSimulator, :82-90 */
    orc_queue_push_chain_log(sorted_enc, n, zero4, sorted_old_tails, sorted_new_tails);
    {
        uint32_t *idx = (uint32_t *)malloc(n * 4); /* lhs: extended_timestamp = position in the unsorted queue, :128-143 */
        for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
        orc_encode_log_queries(q, n, idx, lhs_enc);
        free(idx);
    }
    const uint64_t *u_final = unsorted_new_tails + 4 * (n - 1), *s_final = sorted_new_tails + 4 * (n - 1);
    orc_fs_challenges(u_final, (uint32_t)n, s_final, (uint32_t)n, 4, 21, challenges);
    for (int rep = 0; rep < 2; rep++)
        if (orc_grand_product_chains(lhs_enc, sorted_enc, n, 20, challenges + 21 * rep, lhs_z + rep * n, rhs_z + rep * n) != 0)
            return -9;

    const size_t num_circuits = (n + capacity - 1) / capacity;
    uint64_t cur_lhs[2] = {1, 1}, cur_rhs[2] = {1, 1};
    uint32_t prev_packed[13] = {0}, prev_key[8] = {0}, prev_addr[5] = {0}, prev_ts = 0, cycle_idx = 0;
    uint32_t c_has = 0, c_base[8] = {0}, c_cur[8] = {0}, c_depth = 0;
    size_t it = 0;
    zkw_queue_state4 result_state, cur_unsorted, cur_sorted, cur_final;
    qs4(&result_state, NULL, NULL, 0);
    qs4(&cur_unsorted, NULL, NULL, 0);
    qs4(&cur_sorted, NULL, NULL, 0);
    cur_final = result_state;
#define PUSH_NEXT()                                                                                         \
    do {                                                                                                    \
        orc_queue_push_chain_log(result_enc + 20 * it, 1, result_state.tail, NULL, result_new_tails + 4 * it); \
        memcpy(result_state.tail, result_new_tails + 4 * it, 32);                                           \
        result_state.length++;                                                                              \
        it++;                                                                                               \
    } while (0)
    for (size_t idx = 0; idx < num_circuits; idx++) {
        const size_t lo = idx * capacity, hi = lo + capacity < n ? lo + capacity : n, last = hi - 1;
        const int is_first = idx == 0, is_last = idx == num_circuits - 1;
        uint32_t n_has = c_has, n_base[8], n_cur[8], n_depth = c_depth, cur_addr[5], cur_key[8];
        memcpy(n_base, c_base, 32); memcpy(n_cur, c_cur, 32);
        memcpy(cur_addr, prev_addr, 20); memcpy(cur_key, prev_key, 32);
        int exhausted = 0;
        for (size_t t = lo; t < hi; t++) {
            const zkw_log_query *item = sorted_q + t;
            const int first_ever = (t == lo) && is_first, is_last_ever = (t == hi - 1) && is_last;
            int start_new = first_ever;
            if (!first_ever) {
                const int same = !memcmp(cur_addr, item->address, 20) && !memcmp(cur_key, item->key, 32);
                if (same) {
                    if (item->rw_flag) {
                        if (!item->rollback) { n_depth++; memcpy(n_cur, item->written_value, 32); }
                        else { n_depth--; memcpy(n_cur, item->read_value, 32); }
                    } else {
                        if (n_depth == 0) n_has = 1;
                        memcpy(n_cur, item->read_value, 32);
                    }
                } else {
                    if (n_depth > 0 || n_has) { /* :394-457 */
                        if (it < nd) {
                            const zkw_log_query *nq = result_q + it;
                            const int eq = !memcmp(n_cur, n_base, 32);
                            const int want_rw = n_depth > 0 ? !eq : 0;
                            if (nq->rw_flag != want_rw || memcmp(nq->address, cur_addr, 20) || memcmp(nq->key, cur_key, 32) ||
                                memcmp(nq->read_value, (n_depth > 0 && eq) ? n_cur : n_base, 32) ||
                                memcmp(nq->written_value, n_depth > 0 ? n_cur : n_base, 32))
                                return -10;
                            PUSH_NEXT();
                        } else {
                            if (!is_last || exhausted) return -11;
                            exhausted = 1;
                        }
                    }
                    start_new = 1;
                }
            }
            if (start_new) {
                if (item->rw_flag) {
                    if (item->rollback) return -12;
                    n_depth = 1; n_has = 0;
                    memcpy(n_cur, item->written_value, 32);
                } else {
                    n_depth = 0; n_has = 1;
                    memcpy(n_cur, item->read_value, 32);
                }
                memcpy(n_base, item->read_value, 32);
            }
            memcpy(cur_addr, item->address, 20);
            memcpy(cur_key, item->key, 32);
            if (is_last_ever && !exhausted && (n_depth > 0 || n_has)) {
                if (it >= nd) return -13;
                PUSH_NEXT();
            }
        }
        zkw_storage_sorter_instance *w = instances + idx;
        memset(w, 0, sizeof *w);
        w->start_flag = is_first; w->completion_flag = is_last;
        w->shard_id_to_process = 0;
        w->first_item = lo; w->num_items = hi - lo;
        qs4(&w->unsorted_log_queue_state, NULL, u_final, (uint32_t)n);
        qs4(&w->intermediate_sorted_queue_state, NULL, s_final, (uint32_t)n);
        zkw_storage_sorter_fsm *fi = &w->hidden_fsm_input, *fo = &w->hidden_fsm_output;
        memcpy(fi->lhs_accumulator, cur_lhs, 16); memcpy(fi->rhs_accumulator, cur_rhs, 16);
        fi->current_unsorted_queue_state = cur_unsorted;
        fi->current_intermediate_sorted_queue_state = cur_sorted;
        fi->current_final_sorted_queue_state = cur_final;
        fi->cycle_idx = cycle_idx;
        memcpy(fi->previous_packed_key, prev_packed, 52); memcpy(fi->previous_key, prev_key, 32);
        memcpy(fi->previous_address, prev_addr, 20); fi->previous_timestamp = prev_ts;
        fi->this_cell_has_explicit_read_and_rollback_depth_zero = c_has;
        memcpy(fi->this_cell_base_value, c_base, 32); memcpy(fi->this_cell_current_value, c_cur, 32);
        fi->this_cell_current_depth = c_depth;

        const zkw_log_query *lq = sorted_q + last;
        uint32_t last_packed[13];
        memcpy(last_packed, lq->key, 32); memcpy(last_packed + 8, lq->address, 20); /* comparison_key, log_query.rs:82-92 */
        for (int r = 0; r < 2; r++) { fo->lhs_accumulator[r] = lhs_z[r * n + last]; fo->rhs_accumulator[r] = rhs_z[r * n + last]; }
        qs4(&fo->current_unsorted_queue_state, unsorted_new_tails + 4 * last, u_final, (uint32_t)(n - hi));
        qs4(&fo->current_intermediate_sorted_queue_state, sorted_new_tails + 4 * last, s_final, (uint32_t)(n - hi));
        fo->current_final_sorted_queue_state = result_state;
        fo->cycle_idx = cycle_idx + capacity;
        memcpy(fo->previous_packed_key, last_packed, 52); memcpy(fo->previous_key, lq->key, 32);
        memcpy(fo->previous_address, lq->address, 20); fo->previous_timestamp = sorted_ext_ts[last];
        fo->this_cell_has_explicit_read_and_rollback_depth_zero = n_has;
        memcpy(fo->this_cell_base_value, n_base, 32); memcpy(fo->this_cell_current_value, n_cur, 32);
        fo->this_cell_current_depth = n_depth;
        if ((hi - lo) % capacity != 0) { /* :613-636 */
            memset(fo->previous_packed_key, 0, 52); memset(fo->previous_key, 0, 32);
            memset(fo->previous_address, 0, 20); fo->previous_timestamp = 0;
            fo->this_cell_has_explicit_read_and_rollback_depth_zero = 0;
        } else if (is_last) {
            fo->this_cell_has_explicit_read_and_rollback_depth_zero = 0;
        }
        memcpy(cur_lhs, fo->lhs_accumulator, 16); memcpy(cur_rhs, fo->rhs_accumulator, 16);
        memcpy(prev_packed, last_packed, 52); memcpy(prev_key, lq->key, 32); memcpy(prev_addr, lq->address, 20);
        prev_ts = sorted_ext_ts[last];
        c_has = n_has; memcpy(c_base, n_base, 32); memcpy(c_cur, n_cur, 32); c_depth = n_depth;
        cur_final = result_state;
        cur_unsorted = fo->current_unsorted_queue_state;
        cur_sorted = fo->current_intermediate_sorted_queue_state;
        cycle_idx += capacity;
    }
#undef PUSH_NEXT
    if (it != nd) return -14;
    instances[num_circuits - 1].final_sorted_queue_state = result_state;
    *n_result = nd;
    return (int64_t)num_circuits;
}
