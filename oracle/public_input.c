/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * Public-input commitment of a closed-form input:
 *   simulate_public_input_value_from_witness   src/witness/utils.rs:269-306
 *   CircuitMaker::process                      src/witness/postprocessing/mod.rs:353-405
 *   RecursionRequest::encoding_witness         circuit_encodings/src/recursion_request.rs:13-28
 *
 * `commit_variable_length_encodable_item` / `ClosedFormInputCompactForm::from_full_form` live in the absent
 * zkevm_circuits crate (v1.4.1, fsm_input_output/mod.rs). Restated from its published algorithm: the flat
 * encoding is hashed by a Poseidon2 sponge in overwrite mode from the zero state with the length written into
 * the last capacity element (the same length specialisation produce_fs_challenges uses, utils.rs:511-517), the
 * last chunk zero padded, commitment = the first 4 state words; an empty encoding commits to zero. The field
 * order of the encodings follows the struct declarations as mirrored by the reference's own struct literals
 * (W/ram_permutation.rs:373-409). PARITY UNPINNED: no reference fixture carries a closed-form witness next to
 * its public input (tests/golden/reference_public_inputs.json holds only the outputs).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "../include/zkw_ram_circuit_spec.h"

void orc_commit_var_length(const uint64_t *enc, size_t n, uint64_t out[4]) {
    uint64_t s[12];
    memset(s, 0, sizeof s);
    s[11] = (uint64_t)n; /* apply_length_specialization */
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        memcpy(s, enc + i, 64);
        orc_poseidon2_permutation(s);
    }
    if (i < n) {
        for (int k = 0; k < 8; k++) s[k] = i + k < n ? enc[i + k] : 0;
        orc_poseidon2_permutation(s);
    }
    memcpy(out, s, 32);
}

size_t orc_put_queue12(const zkw_queue_state12 *q, uint64_t *o) {
    memcpy(o, q->head, 96);
    memcpy(o + 12, q->tail, 96);
    o[24] = q->length;
    return 25;
}

/* RamPermutationInputData: unsorted_queue_initial_state, sorted_queue_initial_state, snapshot length */
size_t orc_ram_encode_observable_input(const zkw_ram_instance *in, uint64_t out[ORC_RAM_INPUT_ENC_LEN]) {
    size_t m = 0;
    m += orc_put_queue12(&in->unsorted_queue_initial_state, out + m);
    m += orc_put_queue12(&in->sorted_queue_initial_state, out + m);
    out[m++] = in->non_deterministic_bootloader_memory_snapshot_length;
    return m;
}

/* RamPermutationFSMInputOutput, fields in the order of W/ram_permutation.rs:385-406 */
size_t orc_ram_encode_fsm(const zkw_ram_fsm *f, uint64_t out[ORC_RAM_FSM_ENC_LEN]) {
    size_t m = 0;
    for (int r = 0; r < 2; r++) out[m++] = f->lhs_accumulator[r];
    for (int r = 0; r < 2; r++) out[m++] = f->rhs_accumulator[r];
    m += orc_put_queue12(&f->current_unsorted_queue_state, out + m);
    m += orc_put_queue12(&f->current_sorted_queue_state, out + m);
    for (int k = 0; k < 3; k++) out[m++] = f->previous_sorting_key[k];
    for (int k = 0; k < 2; k++) out[m++] = f->previous_full_key[k];
    for (int k = 0; k < 8; k++) out[m++] = f->previous_value[k];
    out[m++] = f->previous_is_ptr ? 1 : 0;
    out[m++] = f->num_nondeterministic_writes;
    return m;
}

/* compact form = [start, completion, C(observable_input), C(observable_output), C(fsm_in), C(fsm_out)] (18);
   public input = C(compact form). `first` = the first instance of the same block: CircuitMaker::process
   overwrites every instance's observable input with the first one's (postprocessing/mod.rs:358-364). */
void orc_ram_public_input(const zkw_ram_instance *first, const zkw_ram_instance *in, uint64_t compact[18],
                          uint64_t pi[4]) {
    uint64_t buf[ORC_RAM_FSM_ENC_LEN];
    compact[0] = in->start_flag ? 1 : 0;
    compact[1] = in->completion_flag ? 1 : 0;
    size_t m = orc_ram_encode_observable_input(first, buf);
    orc_commit_var_length(buf, m, compact + 2);
    orc_commit_var_length(buf, 0, compact + 6); /* observable output = () */
    m = orc_ram_encode_fsm(&in->hidden_fsm_input, buf);
    orc_commit_var_length(buf, m, compact + 10);
    m = orc_ram_encode_fsm(&in->hidden_fsm_output, buf);
    orc_commit_var_length(buf, m, compact + 14);
    orc_commit_var_length(compact, 18, pi);
}

void orc_ram_public_inputs(const zkw_ram_instance *inst, size_t n, uint64_t *compact, uint64_t *pi) {
    const zkw_ram_instance *first = inst;
    for (size_t i = 0; i < n; i++) {
        if (inst[i].start_flag) first = inst + i;
        orc_ram_public_input(first, inst + i, compact + 18 * i, pi + 4 * i);
    }
}

/* the recursion queue CircuitMaker::process feeds (postprocessing/mod.rs:393-400): tails[i] = full-width queue
   state after pushing RecursionRequest{circuit_type, pi[i]} */
void orc_recursion_queue(uint64_t circuit_type, const uint64_t *pi, size_t n, const uint64_t tail_in[12],
                         uint64_t *enc /* n*8 */, uint64_t *tails /* n*12 */) {
    for (size_t i = 0; i < n; i++) orc_encode_recursion_request(circuit_type, pi + 4 * i, enc + 8 * i);
    orc_queue_push_chain_full(enc, n, tail_in, tails);
}

/* ---- CodeDecommittmentsSorter (type 2): encodings in the order of the reference's struct literals
   (sort_decommit_requests.rs:402-414; DecommitQuery = {code_hash, page, is_first, timestamp}) */
size_t orc_ds_encode_fsm(const zkw_decommit_sorter_fsm *f, uint64_t out[ORC_DS_FSM_ENC_LEN]) {
    size_t m = 0;
    m += orc_put_queue12(&f->initial_queue_state, out + m);
    m += orc_put_queue12(&f->sorted_queue_state, out + m);
    m += orc_put_queue12(&f->final_queue_state, out + m);
    for (int r = 0; r < 2; r++) out[m++] = f->lhs_accumulator[r];
    for (int r = 0; r < 2; r++) out[m++] = f->rhs_accumulator[r];
    for (int k = 0; k < 9; k++) out[m++] = f->previous_packed_key[k];
    for (int k = 0; k < 8; k++) out[m++] = f->previous_record.hash[k];
    out[m++] = f->previous_record.memory_page;
    out[m++] = f->previous_record.is_fresh ? 1 : 0;
    out[m++] = f->previous_record.timestamp;
    out[m++] = f->first_encountered_timestamp;
    return m;
}

void orc_ds_public_inputs(const zkw_decommit_sorter_instance *inst, size_t n, uint64_t *compact, uint64_t *pi) {
    const zkw_decommit_sorter_instance *first = inst;
    uint64_t buf[ORC_DS_FSM_ENC_LEN];
    for (size_t i = 0; i < n; i++) {
        if (inst[i].start_flag) first = inst + i;
        uint64_t *cf = compact + 18 * i;
        cf[0] = inst[i].start_flag ? 1 : 0;
        cf[1] = inst[i].completion_flag ? 1 : 0;
        size_t m = orc_put_queue12(&first->initial_queue_state, buf);
        m += orc_put_queue12(&first->sorted_queue_initial_state, buf + m);
        orc_commit_var_length(buf, m, cf + 2);
        m = orc_put_queue12(&inst[i].final_queue_state, buf);
        orc_commit_var_length(buf, m, cf + 6);
        m = orc_ds_encode_fsm(&inst[i].hidden_fsm_input, buf);
        orc_commit_var_length(buf, m, cf + 10);
        m = orc_ds_encode_fsm(&inst[i].hidden_fsm_output, buf);
        orc_commit_var_length(buf, m, cf + 14);
        orc_commit_var_length(cf, 18, pi + 4 * i);
    }
}

/* ---- the 4-wide log-queue circuits (LogDemuxer 4, StorageSorter 9, EventsSorter / L1MessagesSorter 11 / 12). Field
   order = the struct declarations as mirrored by include/zkw_types.h (the reference's struct literals: log_demux.rs:
   283-301, storage_sort_dedup.rs:577-612, events_sort_dedup.rs:426-455); LogQuery in the declaration order of the
   in-circuit struct (absent crate). PARITY UNPINNED like the types above. */
size_t orc_put_queue4(const zkw_queue_state4 *q, uint64_t *o) {
    memcpy(o, q->head, 32);
    memcpy(o + 4, q->tail, 32);
    o[8] = q->length;
    return 9;
}
static size_t put_log_query(const zkw_log_query *q, uint64_t *o) {
    size_t m = 0;
    for (int k = 0; k < 5; k++) o[m++] = q->address[k];
    for (int k = 0; k < 8; k++) o[m++] = q->key[k];
    for (int k = 0; k < 8; k++) o[m++] = q->read_value[k];
    for (int k = 0; k < 8; k++) o[m++] = q->written_value[k];
    o[m++] = q->rw_flag ? 1 : 0;
    o[m++] = q->aux_byte;
    o[m++] = q->rollback ? 1 : 0;
    o[m++] = q->is_service ? 1 : 0;
    o[m++] = q->shard_id;
    o[m++] = q->tx_number_in_block;
    o[m++] = q->timestamp;
    return m;
}
static void compact_and_pi(int start, int completion, const uint64_t *in, size_t n_in, const uint64_t *out, size_t n_out,
                           const uint64_t *fi, size_t n_fi, const uint64_t *fo, size_t n_fo, uint64_t cf[18], uint64_t pi[4]) {
    cf[0] = start ? 1 : 0;
    cf[1] = completion ? 1 : 0;
    orc_commit_var_length(in, n_in, cf + 2);
    orc_commit_var_length(out, n_out, cf + 6);
    orc_commit_var_length(fi, n_fi, cf + 10);
    orc_commit_var_length(fo, n_fo, cf + 14);
    orc_commit_var_length(cf, 18, pi);
}

size_t orc_ld_fsm(const zkw_log_demux_fsm *f, uint64_t *o) {
    size_t m = orc_put_queue4(&f->initial_log_queue_state, o);
    for (int c = 0; c < ZKW_DEMUX_NUM_QUEUES; c++) m += orc_put_queue4(&f->queue_state[c], o + m);
    return m;
}
void orc_log_demux_public_inputs(const zkw_log_demux_instance *inst, size_t n, uint64_t *compact, uint64_t *pi) {
    const zkw_log_demux_instance *first = inst;
    uint64_t in[16], out[64], fi[64], fo[64];
    for (size_t i = 0; i < n; i++) {
        if (inst[i].start_flag) first = inst + i;
        const size_t n_in = orc_put_queue4(&first->initial_log_queue_state, in);
        size_t n_out = 0;
        for (int c = 0; c < ZKW_DEMUX_NUM_QUEUES; c++) n_out += orc_put_queue4(&inst[i].output_queue_state[c], out + n_out);
        const size_t n_fi = orc_ld_fsm(&inst[i].hidden_fsm_input, fi), n_fo = orc_ld_fsm(&inst[i].hidden_fsm_output, fo);
        compact_and_pi(inst[i].start_flag, inst[i].completion_flag, in, n_in, out, n_out, fi, n_fi, fo, n_fo, compact + 18 * i, pi + 4 * i);
    }
}

size_t orc_es_fsm(const zkw_events_sorter_fsm *f, uint64_t *o) {
    size_t m = 0;
    for (int r = 0; r < 2; r++) o[m++] = f->lhs_accumulator[r];
    for (int r = 0; r < 2; r++) o[m++] = f->rhs_accumulator[r];
    m += orc_put_queue4(&f->initial_unsorted_queue_state, o + m);
    m += orc_put_queue4(&f->intermediate_sorted_queue_state, o + m);
    m += orc_put_queue4(&f->final_result_queue_state, o + m);
    o[m++] = f->previous_key;
    return m + put_log_query(&f->previous_item, o + m);
}
void orc_events_sorter_public_inputs(const zkw_events_sorter_instance *inst, size_t n, uint64_t *compact, uint64_t *pi) {
    const zkw_events_sorter_instance *first = inst;
    uint64_t in[32], out[16], fi[80], fo[80];
    for (size_t i = 0; i < n; i++) {
        if (inst[i].start_flag) first = inst + i;
        size_t n_in = orc_put_queue4(&first->initial_log_queue_state, in);
        n_in += orc_put_queue4(&first->intermediate_sorted_queue_state, in + n_in);
        const size_t n_out = orc_put_queue4(&inst[i].final_queue_state, out);
        const size_t n_fi = orc_es_fsm(&inst[i].hidden_fsm_input, fi), n_fo = orc_es_fsm(&inst[i].hidden_fsm_output, fo);
        compact_and_pi(inst[i].start_flag, inst[i].completion_flag, in, n_in, out, n_out, fi, n_fi, fo, n_fo, compact + 18 * i, pi + 4 * i);
    }
}

size_t orc_ss_fsm(const zkw_storage_sorter_fsm *f, uint64_t *o) {
    size_t m = 0;
    for (int r = 0; r < 2; r++) o[m++] = f->lhs_accumulator[r];
    for (int r = 0; r < 2; r++) o[m++] = f->rhs_accumulator[r];
    m += orc_put_queue4(&f->current_unsorted_queue_state, o + m);
    m += orc_put_queue4(&f->current_intermediate_sorted_queue_state, o + m);
    m += orc_put_queue4(&f->current_final_sorted_queue_state, o + m);
    o[m++] = f->cycle_idx;
    for (int k = 0; k < ZKW_STORAGE_PACKED_KEY_LENGTH; k++) o[m++] = f->previous_packed_key[k];
    for (int k = 0; k < 8; k++) o[m++] = f->previous_key[k];
    for (int k = 0; k < 5; k++) o[m++] = f->previous_address[k];
    o[m++] = f->previous_timestamp;
    o[m++] = f->this_cell_has_explicit_read_and_rollback_depth_zero ? 1 : 0;
    for (int k = 0; k < 8; k++) o[m++] = f->this_cell_base_value[k];
    for (int k = 0; k < 8; k++) o[m++] = f->this_cell_current_value[k];
    o[m++] = f->this_cell_current_depth;
    return m;
}
void orc_storage_sorter_public_inputs(const zkw_storage_sorter_instance *inst, size_t n, uint64_t *compact, uint64_t *pi) {
    const zkw_storage_sorter_instance *first = inst;
    uint64_t in[32], out[16], fi[80], fo[80];
    for (size_t i = 0; i < n; i++) {
        if (inst[i].start_flag) first = inst + i;
        in[0] = first->shard_id_to_process;
        size_t n_in = 1 + orc_put_queue4(&first->unsorted_log_queue_state, in + 1);
        n_in += orc_put_queue4(&first->intermediate_sorted_queue_state, in + n_in);
        const size_t n_out = orc_put_queue4(&inst[i].final_sorted_queue_state, out);
        const size_t n_fi = orc_ss_fsm(&inst[i].hidden_fsm_input, fi), n_fo = orc_ss_fsm(&inst[i].hidden_fsm_output, fo);
        compact_and_pi(inst[i].start_flag, inst[i].completion_flag, in, n_in, out, n_out, fi, n_fi, fo, n_fo, compact + 18 * i, pi + 4 * i);
    }
}

/* ---- the remaining non-VM circuits: CodeDecommitter 3, Keccak256 / Sha256 / ECRecover round functions 5 / 6 / 7,
   StorageApplication 10, LinearHasher 13. The closed-form structs live in the absent zkevm_circuits crate (v1.4.1:
   code_unpacker_sha256/input.rs, keccak256_round_function/input.rs, sha256_round_function/input.rs, ecrecover/input.rs,
   storage_application/input.rs, linear_hasher/input.rs); field order restated from their published declarations, the
   field SETS are the ones the reference's builders fill (decommit_code.rs:172-199,363-401; keccak256_round_function.rs:
   420-441; sha256_round_function.rs:302-316; ecrecover.rs:215-233; storage_application.rs:286-336;
   data_hasher_and_merklizer.rs:34-60). One field element per Boolean / UInt8 / UInt16 / UInt32, eight per UInt256
   (u32 limbs, least significant first). PARITY UNPINNED like the types above. */
static size_t put_bytes(const uint8_t *b, size_t n, uint64_t *o) {
    for (size_t k = 0; k < n; k++) o[k] = b[k];
    return n;
}
static size_t put_u32s(const uint32_t *w, size_t n, uint64_t *o) {
    for (size_t k = 0; k < n; k++) o[k] = w[k];
    return n;
}
/* CodeDecommitterFSMInputOutput { internal_fsm: CodeDecommittmentFSM, decommittment_requests_queue_state, memory_queue_state } */
static size_t dcm_fsm(const zkw_decommitter_fsm *f, uint64_t *o) {
    size_t m = put_u32s(f->sha256_inner_state, 8, o);
    m += put_u32s(f->hash_to_compare_against, 8, o + m);
    o[m++] = f->current_index;
    o[m++] = f->current_page;
    o[m++] = f->timestamp;
    o[m++] = f->num_rounds_left;
    o[m++] = f->length_in_bits;
    o[m++] = f->state_get_from_queue ? 1 : 0;
    o[m++] = f->state_decommit ? 1 : 0;
    o[m++] = f->finished ? 1 : 0;
    m += orc_put_queue12(&f->decommittment_requests_queue_state, o + m);
    m += orc_put_queue12(&f->memory_queue_state, o + m);
    return m;
}
/* {Keccak256,Sha256}RoundFunctionFSMInputOutput { internal_fsm, log_queue_state, memory_queue_state };
   EcrecoverCircuitFSMInputOutput { log_queue_state, memory_queue_state } */
static size_t pre_fsm(int kind, const zkw_precompile_fsm *f, uint64_t *o) {
    size_t m = 0;
    if (kind == ZKW_PRECOMPILE_KECCAK256) {
        o[m++] = f->read_precompile_call ? 1 : 0;
        o[m++] = f->read_words_for_round ? 1 : 0;
        o[m++] = f->completed ? 1 : 0;
        o[m++] = f->padding_round ? 1 : 0;
        m += put_bytes(f->keccak_internal_state, 200, o + m);
        o[m++] = f->timestamp_to_use_for_read;
        o[m++] = f->timestamp_to_use_for_write;
        o[m++] = f->input_page;   /* Keccak256PrecompileCallParams */
        o[m++] = f->input_offset;
        o[m++] = f->input_length;
        o[m++] = f->output_page;
        o[m++] = f->output_offset;
        o[m++] = f->needs_full_padding_round ? 1 : 0;
        m += put_bytes(f->buffer_bytes, ZKW_KECCAK_PRECOMPILE_BUFFER_SIZE, o + m); /* ByteBuffer { bytes, filled } */
        o[m++] = f->buffer_filled;
    } else if (kind == ZKW_PRECOMPILE_SHA256) {
        o[m++] = f->read_precompile_call ? 1 : 0;
        o[m++] = f->read_words_for_round ? 1 : 0;
        o[m++] = f->completed ? 1 : 0;
        m += put_u32s(f->sha256_inner_state, 8, o + m);
        o[m++] = f->timestamp_to_use_for_read;
        o[m++] = f->timestamp_to_use_for_write;
        o[m++] = f->input_page;   /* Sha256PrecompileCallParams */
        o[m++] = f->input_offset;
        o[m++] = f->output_page;
        o[m++] = f->output_offset;
        o[m++] = f->num_rounds;
    }
    m += orc_put_queue4(&f->log_queue_state, o + m);
    m += orc_put_queue12(&f->memory_queue_state, o + m);
    return m;
}
/* StorageApplicationFSMInputOutput { current_root_hash, next_enumeration_counter, current_storage_application_log_state,
   current_diffs_keccak_accumulator_state } */
static size_t sap_fsm(const zkw_storage_application_fsm *f, uint64_t *o) {
    size_t m = put_bytes(f->current_root_hash, 32, o);
    m += put_u32s(f->next_enumeration_counter, 2, o + m);
    m += orc_put_queue4(&f->current_storage_application_log_state, o + m);
    m += put_bytes(f->current_diffs_keccak_accumulator_state, 200, o + m);
    return m;
}

/* the four flat encodings of instance i (w: 4 slices of ORC_CF_MAX_FSM_LEN words: observable input — of the block's FIRST instance,
   postprocessing/mod.rs:358-364 —, observable output, hidden FSM input, hidden FSM output), their lengths, {start, completion} */
int orc_cf_encode(int circuit_type, const void *instances, size_t i, uint64_t *w, size_t n[4], int flags[2]) {
    uint64_t *in = w, *out = w + ORC_CF_MAX_FSM_LEN, *fi = w + 2 * ORC_CF_MAX_FSM_LEN, *fo = w + 3 * ORC_CF_MAX_FSM_LEN;
    size_t n_in = 0, n_out = 0, n_fi = 0, n_fo = 0, first = i;
    int start = 0, completion = 0;
    switch (circuit_type) {
        case 3: {
            const zkw_decommitter_instance *w_ = (const zkw_decommitter_instance *)instances;
            while (first > 0 && !w_[first].start_flag) first--;
            start = w_[i].start_flag; completion = w_[i].completion_flag;
            /* CodeDecommitterInputData { memory_queue_initial_state, sorted_requests_queue_initial_state } */
            n_in = orc_put_queue12(&w_[first].memory_queue_initial_state, in);
            n_in += orc_put_queue12(&w_[first].sorted_requests_queue_initial_state, in + n_in);
            n_out = orc_put_queue12(&w_[i].memory_queue_final_state, out);
            n_fi = dcm_fsm(&w_[i].hidden_fsm_input, fi);
            n_fo = dcm_fsm(&w_[i].hidden_fsm_output, fo);
            break;
        }
        case 5: case 6: case 7: {
            const zkw_precompile_instance *w_ = (const zkw_precompile_instance *)instances;
            const int kind = circuit_type - 5; /* ZKW_PRECOMPILE_KECCAK256 .. ZKW_PRECOMPILE_ECRECOVER */
            while (first > 0 && !w_[first].start_flag) first--;
            start = w_[i].start_flag; completion = w_[i].completion_flag;
            /* PrecompileFunctionInputData { initial_log_queue_state, initial_memory_queue_state } */
            n_in = orc_put_queue4(&w_[first].initial_log_queue_state, in);
            n_in += orc_put_queue12(&w_[first].initial_memory_queue_state, in + n_in);
            n_out = orc_put_queue12(&w_[i].final_memory_state, out);
            n_fi = pre_fsm(kind, &w_[i].hidden_fsm_input, fi);
            n_fo = pre_fsm(kind, &w_[i].hidden_fsm_output, fo);
            break;
        }
        case 10: {
            const zkw_storage_application_instance *w_ = (const zkw_storage_application_instance *)instances;
            while (first > 0 && !w_[first].start_flag) first--;
            start = w_[i].start_flag; completion = w_[i].completion_flag;
            /* StorageApplicationInputData { shard, initial_root_hash, initial_next_enumeration_counter,
               storage_application_log_state } */
            in[0] = w_[first].shard;
            n_in = 1 + put_bytes(w_[first].initial_root_hash, 32, in + 1);
            n_in += put_u32s(w_[first].initial_next_enumeration_counter, 2, in + n_in);
            n_in += orc_put_queue4(&w_[first].storage_application_log_state, in + n_in);
            /* StorageApplicationOutputData { new_root_hash, new_next_enumeration_counter, state_diffs_keccak256_hash } */
            n_out = put_bytes(w_[i].new_root_hash, 32, out);
            n_out += put_u32s(w_[i].new_next_enumeration_counter, 2, out + n_out);
            n_out += put_bytes(w_[i].state_diffs_keccak256_hash, 32, out + n_out);
            n_fi = sap_fsm(&w_[i].hidden_fsm_input, fi);
            n_fo = sap_fsm(&w_[i].hidden_fsm_output, fo);
            break;
        }
        case 13: {
            const zkw_linear_hasher_instance *w_ = (const zkw_linear_hasher_instance *)instances;
            start = w_[i].start_flag; completion = w_[i].completion_flag;
            n_in = orc_put_queue4(&w_[i].queue_state, in);      /* LinearHasherInputData { queue_state } */
            n_out = put_bytes(w_[i].keccak256_hash, 32, out); /* LinearHasherOutputData { keccak256_hash } */
            break;                                          /* hidden FSM = (): nothing absorbed */
        }
        default: return -1;
    }
    n[0] = n_in; n[1] = n_out; n[2] = n_fi; n[3] = n_fo;
    flags[0] = start ? 1 : 0; flags[1] = completion ? 1 : 0;
    return 0;
}

int orc_closed_form_public_inputs(int circuit_type, const void *instances, size_t n, uint64_t *compact, uint64_t *pi) {
    uint64_t *w = malloc(4 * ORC_CF_MAX_FSM_LEN * sizeof(uint64_t));
    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; i++) {
        size_t m[4];
        int flags[2];
        rc = orc_cf_encode(circuit_type, instances, i, w, m, flags);
        if (rc == 0)
            compact_and_pi(flags[0], flags[1], w, m[0], w + ORC_CF_MAX_FSM_LEN, m[1], w + 2 * ORC_CF_MAX_FSM_LEN, m[2], w + 3 * ORC_CF_MAX_FSM_LEN, m[3],
                           compact + 18 * i, pi + 4 * i);
    }
    free(w);
    return rc;
}
