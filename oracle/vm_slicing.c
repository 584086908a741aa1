/* vm_slicing.c — TEST INFRASTRUCTURE: CPU restatement, sequential and literal (take_while / skip_while as loops), of the
 * MainVM instance slicing of src/witness/oracle.rs:1229-1469 and of the flags / observable parts that
 * vm_instance_witness_to_circuit_formal_input (src/witness/utils.rs:428-496) derives. Checker of zkw_vm_slice_instances. */
#include "oracle.h"
#include <string.h>

static void full_queue_state(const uint64_t *tails, size_t k_plus_one, zkw_queue_state12 *st) {
    /* transform_sponge_like_queue_state(all_*_queue_states[k]) (src/witness/utils.rs:73-85) copies the intermediate state's
       `head`, which push_and_output_intermediate_data fills with the simulator's own head (circuit_encodings/src/lib.rs:419-421):
       the memory and decommitment simulators only ever push, so it stays the initial all-zero head */
    memset(st, 0, sizeof *st);
    if (k_plus_one >= 1) memcpy(st->tail, tails + 12 * (k_plus_one - 1), 96);
    st->length = (uint32_t)k_plus_one;
}

static void aux_for_entry(const zkw_vm_tracer_streams *s, uint32_t at_cycle, zkw_vm_aux_parameters *a) {
    memset(a, 0, sizeof *a);
    /* :1245-1253 */
    size_t index_plus_one = 0;
    while (index_plus_one < s->stream_len[ZKW_VMS_MEMORY] && s->stream_cycles[ZKW_VMS_MEMORY][index_plus_one] < at_cycle) index_plus_one++;
    full_queue_state(s->memory_queue_tails, index_plus_one, &a->memory_queue_state);
    /* :1255-1261 */
    size_t d = 0;
    while (d < s->n_decommit_states && s->decommit_state_cycles[d] < at_cycle) d++;
    full_queue_state(s->decommit_queue_tails, d, &a->decommittment_queue_state);
    /* :1265-1273 */
    size_t c = 0;
    while (c < s->n_callstack_sponges && s->callstack_sponge_cycles[c] < at_cycle) c++;
    if (c) memcpy(a->callstack_state, s->callstack_sponge_states + 12 * (c - 1), 96);
    /* :1359-1375 */
    size_t l = 0;
    while (l < s->n_storage_log_states && s->storage_log_state_cycles[l] < at_cycle) l++;
    if (l) {
        const zkw_storage_log_detailed_state *st = &s->storage_log_states[l - 1];
        memcpy(a->storage_log_queue_state.tail, st->forward_tail, 32);
        a->storage_log_queue_state.length = st->forward_length;
        memcpy(a->current_frame_rollback_queue_tail, st->rollback_tail, 32);
        memcpy(a->current_frame_rollback_queue_head, st->rollback_head, 32);
        a->current_frame_rollback_queue_segment_length = st->rollback_length;
    } else {
        memcpy(a->current_frame_rollback_queue_tail, s->global_end_of_storage_log, 32);
        memcpy(a->current_frame_rollback_queue_head, s->global_end_of_storage_log, 32);
    }
}

int orc_vm_slice_instances(const zkw_vm_tracer_streams *s, zkw_vm_instance *out, uint32_t *read_index, uint32_t *write_index,
                           uint64_t *n_reads, uint64_t *n_writes) {
    if (s->n_snapshots < 2) return -1;
    const size_t n_inst = s->n_snapshots - 1;
    uint64_t reads = 0, writes = 0;
    for (size_t i = 0; i < n_inst; i++) {
        zkw_vm_instance *v = &out[i];
        memset(v, 0, sizeof *v);
        const uint32_t from = s->snapshot_cycles[i], to = s->snapshot_cycles[i + 1];
        v->start_flag = i == 0;
        v->completion_flag = i + 1 == n_inst;
        v->cycle_from = from;
        v->cycle_to = to;
        v->snapshot_initial = (uint32_t)i;
        v->snapshot_final = (uint32_t)i + 1;
        for (int k = 0; k < ZKW_VM_NUM_STREAMS; k++) { /* skip_while(< from).take_while(< to), :1278-1343 */
            size_t lo = 0;
            while (lo < s->stream_len[k] && s->stream_cycles[k][lo] < from) lo++;
            size_t hi = lo;
            while (hi < s->stream_len[k] && s->stream_cycles[k][hi] < to) hi++;
            v->range[k][0] = lo;
            v->range[k][1] = hi;
        }
        v->first_memory_read = reads;
        v->first_memory_write = writes;
        for (size_t k = v->range[ZKW_VMS_MEMORY][0]; k < v->range[ZKW_VMS_MEMORY][1]; k++) { /* :1281-1292 */
            if (s->vm_memory_queries[k].rw_flag) { if (write_index) write_index[writes] = (uint32_t)k; writes++; }
            else { if (read_index) read_index[reads] = (uint32_t)k; reads++; }
        }
        v->num_memory_reads = reads - v->first_memory_read;
        v->num_memory_writes = writes - v->first_memory_write;
        aux_for_entry(s, from, &v->auxilary_initial_parameters);
        if (i) out[i - 1].auxilary_final_parameters = v->auxilary_initial_parameters; /* :1406-1409 */
    }
    { /* special pass for the last one, :1414-1468 */
        zkw_vm_aux_parameters *a = &out[n_inst - 1].auxilary_final_parameters;
        memset(a, 0, sizeof *a);
        full_queue_state(s->memory_queue_tails, s->stream_len[ZKW_VMS_MEMORY], &a->memory_queue_state);
        full_queue_state(s->decommit_queue_tails, s->n_decommit_states, &a->decommittment_queue_state);
        if (s->n_storage_log_states) {
            const zkw_storage_log_detailed_state *st = &s->storage_log_states[s->n_storage_log_states - 1];
            memcpy(a->storage_log_queue_state.tail, st->forward_tail, 32);
            a->storage_log_queue_state.length = st->forward_length;
            memcpy(a->current_frame_rollback_queue_tail, st->rollback_tail, 32);
            memcpy(a->current_frame_rollback_queue_head, st->rollback_head, 32);
            a->current_frame_rollback_queue_segment_length = st->rollback_length;
        }
    }
    /* vm_instance_witness_to_circuit_formal_input, src/witness/utils.rs:456-483 */
    {
        zkw_vm_instance *f = &out[0];
        const zkw_vm_aux_parameters *a = &f->auxilary_initial_parameters;
        memcpy(f->rollback_queue_tail_for_block, a->current_frame_rollback_queue_tail, 32);
        memcpy(f->memory_queue_initial_tail, a->memory_queue_state.tail, 96);
        f->memory_queue_initial_length = a->memory_queue_state.length;
        memcpy(f->decommitment_queue_initial_tail, a->decommittment_queue_state.tail, 96);
        f->decommitment_queue_initial_length = a->decommittment_queue_state.length;
        zkw_vm_instance *l = &out[n_inst - 1];
        l->memory_queue_final_state = l->auxilary_final_parameters.memory_queue_state;
        l->decommitment_queue_final_state = l->auxilary_final_parameters.decommittment_queue_state;
        l->log_queue_final_state = l->auxilary_final_parameters.storage_log_queue_state;
    }
    /* the writes / reads outside every window (after the last snapshot) stay out, as in the reference */
    if (n_reads) *n_reads = reads;
    if (n_writes) *n_writes = writes;
    return 0;
}
