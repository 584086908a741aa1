/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 *   ExtendedCallstackEntry::encoding_witness     circuit_encodings/src/callstack_entry.rs:36-179
 *   FullWidthStackSimulator::{push, pop}         circuit_encodings/src/lib.rs:558-644
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static int is_kernel_address(const uint32_t a[5]) {
    /* zk_evm CallStackEntry::is_kernel_mode: bytes 0..18 of the big-endian address are zero */
    return a[4] == 0 && a[3] == 0 && a[2] == 0 && a[1] == 0 && a[0] < (1u << 16);
}

void orc_encode_callstack_entry(const zkw_callstack_entry *e, uint64_t o[32]) {
    for (int k = 0; k < 4; k++) { o[k] = e->rollback_queue_head[k]; o[4 + k] = e->rollback_queue_tail[k]; }
    for (int k = 0; k < 5; k++) { o[8 + k] = e->code_address[k]; o[13 + k] = e->this_address[k]; o[18 + k] = e->msg_sender[k]; }
    for (int k = 0; k < 4; k++) o[23 + k] = e->context_u128_value[k];
    o[27] = (uint64_t)e->code_page | ((uint64_t)e->pc << 32) | ((uint64_t)e->this_shard_id << 48) |
            ((uint64_t)(e->is_static ? 1 : 0) << 56);
    o[28] = (uint64_t)e->base_memory_page | ((uint64_t)e->sp << 32) | ((uint64_t)e->caller_shard_id << 48) |
            ((uint64_t)is_kernel_address(e->this_address) << 56);
    o[29] = (uint64_t)e->ergs_remaining | ((uint64_t)e->exception_handler_location << 32) |
            ((uint64_t)e->code_shard_id << 48) | ((uint64_t)(e->is_local_frame ? 1 : 0) << 56);
    const uint32_t len = e->rollback_queue_segment_length;
    o[30] = (uint64_t)e->heap_bound | ((uint64_t)(len & 0xFF) << 32) | ((uint64_t)((len >> 8) & 0xFF) << 40);
    o[31] = (uint64_t)e->aux_heap_bound | ((uint64_t)((len >> 16) & 0xFF) << 32) | ((uint64_t)(len >> 24) << 40);
}

/* Replays `n_ops` operations on an empty stack. is_push[i] != 0 pushes the next entry of `pushed`, otherwise
   the top is popped. Per operation (FullWidthStackIntermediateStates, lib.rs:511-521): previous_state,
   new_state, depth (= num_items after the operation), the 4 round outputs of the entry's absorption
   (round_function_execution_pairs[r].1; the inputs are the previous output with the rate overwritten), and
   the index in `pushed` of the element pushed / returned. Returns 0, -1 on a pop from the empty stack
   (`self.witness.pop().unwrap()`, lib.rs:619), -2 when pushes outnumber `n_pushed`. */
int orc_callstack_simulate(const uint8_t *is_push, size_t n_ops, const zkw_callstack_entry *pushed, size_t n_pushed,
                           uint64_t *previous_state, uint64_t *new_state, uint32_t *depth, uint64_t *round_states,
                           uint32_t *entry_index) {
    size_t cap = n_ops + 1, sp = 0, next = 0;
    uint32_t *stack_entry = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint64_t *stack_prev = (uint64_t *)malloc(cap * 12 * sizeof(uint64_t)); /* witness[k].1: state before the push */
    uint64_t state[12];
    memset(state, 0, sizeof state);
    int rc = 0;
    for (size_t i = 0; i < n_ops; i++) {
        uint64_t enc[32], cur[12];
        memcpy(cur, state, 96);
        if (is_push[i]) {
            if (next >= n_pushed) { rc = -2; break; }
            orc_encode_callstack_entry(pushed + next, enc);
            memcpy(stack_prev + 12 * sp, state, 96);
            stack_entry[sp++] = (uint32_t)next;
            orc_absorb_multiple_rounds(state, enc, 4, round_states + 48 * i);
            memcpy(previous_state + 12 * i, cur, 96);
            memcpy(new_state + 12 * i, state, 96);
            entry_index[i] = (uint32_t)next++;
        } else {
            if (sp == 0) { rc = -1; break; }
            sp--;
            const uint32_t k = stack_entry[sp];
            orc_encode_callstack_entry(pushed + k, enc);
            uint64_t re[12];
            memcpy(re, stack_prev + 12 * sp, 96);
            orc_absorb_multiple_rounds(re, enc, 4, round_states + 48 * i);
            if (memcmp(re, state, 96) != 0) { rc = -3; break; } /* assert_eq!(new_state, self.state), lib.rs:631 */
            memcpy(state, stack_prev + 12 * sp, 96);
            memcpy(previous_state + 12 * i, cur, 96);
            memcpy(new_state + 12 * i, state, 96);
            entry_index[i] = k;
        }
        depth[i] = (uint32_t)sp;
    }
    free(stack_entry);
    free(stack_prev);
    return rc;
}

void orc_encode_callstack_entries(const zkw_callstack_entry *e, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; i++) orc_encode_callstack_entry(e + i, out + 32 * i);
}
