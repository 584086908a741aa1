/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * Recursion-layer witnesses, src/witness/recursive_aggregation.rs:
 *   compute_encodable_item_from_witness::<AllocatedVerificationKey>  :45-68   orc_vk_commitment
 *   compute_leaf_params                                              :163-216 orc_leaf_params
 *   compute_leaf_vks_and_params_commitment                           :218-240 orc_leaf_vks_and_params_commitment
 *   create_leaf_witnesses (the leaf circuit's public input)          :71-161  orc_leaf_public_input
 *   create_node_witnesses (merge, split points, public input)        :270-421 orc_node_witness
 * `CircuitVarLengthEncodable` of the structs involved lives in absent crates (boojum: AllocatedVerificationKey, QueueState;
 * zkevm_circuits: RecursionLeafParameters, RecursionLeafInput, RecursionNodeInput). PINNED by the reference's committed
 * proofs: tests/test_oracle_recursion.py reproduces the public inputs of leaf_layer_proof_{6,10,15}_0.json from
 * setup/base_layer/vk_{4,8,13}.json, setup/recursion_layer/vk_{6,10,15}.json and the base proofs' public inputs. The node
 * input follows the same rules (field order of the struct literal at :307-312) but no committed node proof is reachable
 * (each commits to all 13 leaf parameter sets, 10 of which predate the committed VKs).
 */
#include <string.h>
#include "oracle.h"

void orc_vk_commitment(const uint64_t *cap, size_t cap_size, uint64_t out[4]) { orc_commit_var_length(cap, 4 * cap_size, out); }

void orc_leaf_params(uint8_t circuit_type, const uint64_t *base_cap, const uint64_t *leaf_cap, size_t cap_size, zkw_leaf_params *out) {
    out->circuit_type = circuit_type;
    orc_vk_commitment(base_cap, cap_size, out->basic_circuit_vk_commitment);
    orc_vk_commitment(leaf_cap, cap_size, out->leaf_layer_vk_commitment);
}

static size_t put_params(const zkw_leaf_params *p, uint64_t *o) {
    o[0] = p->circuit_type;
    memcpy(o + 1, p->basic_circuit_vk_commitment, 32);
    memcpy(o + 5, p->leaf_layer_vk_commitment, 32);
    return 9;
}
static size_t put_queue12(const zkw_queue_state12 *q, uint64_t *o) {
    memcpy(o, q->head, 96);
    memcpy(o + 12, q->tail, 96);
    o[24] = q->length;
    return 25;
}

void orc_leaf_vks_and_params_commitment(const zkw_leaf_params *p /* [13] */, uint64_t out[4]) {
    uint64_t enc[9 * 13];
    for (int t = 0; t < 13; t++) put_params(p + t, enc + 9 * t);
    orc_commit_var_length(enc, 9 * 13, out);
}

/* public input of a leaf circuit = commit_variable_length_encodable_item(RecursionLeafInput{params, queue_state}) */
void orc_leaf_public_input(const zkw_leaf_params *params, const zkw_queue_state12 *queue_state, uint64_t out[4]) {
    uint64_t enc[34];
    size_t m = put_params(params, enc);
    m += put_queue12(queue_state, enc + m);
    orc_commit_var_length(enc, m, out);
}

/* one node over chunks[0..n_chunks), n_chunks <= 32: returns 0, or -1 when a chunk is empty / does not chain */
int orc_node_witness(uint8_t branch_circuit_type, const zkw_leaf_params *leaf_layer_params /* [13] */, const uint64_t node_vk_commitment[4],
                     const zkw_queue_state12 *chunks, size_t n_chunks, zkw_queue_state12 *node_state, zkw_queue_tail12 *split_points /* [31] */,
                     uint64_t public_input[4]) {
    if (n_chunks == 0 || n_chunks > 32) return -1;
    zkw_queue_state12 q = chunks[0];
    size_t n_sp = 0;
    for (size_t c = 0; c < n_chunks; c++) {
        if (chunks[c].length == 0) return -1;
        if (c) {
            if (memcmp(q.tail, chunks[c].head, 96)) return -1;
            memcpy(q.tail, chunks[c].tail, 96);
            q.length += chunks[c].length;
        }
        if (n_sp < 31) {
            memset(&split_points[n_sp], 0, sizeof split_points[n_sp]);
            memcpy(split_points[n_sp].tail, chunks[c].tail, 96);
            split_points[n_sp++].length = chunks[c].length;
        }
    }
    for (; n_sp < 31; n_sp++) {
        memset(&split_points[n_sp], 0, sizeof split_points[n_sp]);
        memcpy(split_points[n_sp].tail, q.tail, 96);
    }
    *node_state = q;
    uint64_t enc[1 + 9 * 13 + 4 + 25];
    size_t m = 0;
    enc[m++] = branch_circuit_type;
    for (int t = 0; t < 13; t++) m += put_params(leaf_layer_params + t, enc + m);
    memcpy(enc + m, node_vk_commitment, 32);
    m += 4;
    m += put_queue12(&q, enc + m);
    orc_commit_var_length(enc, m, public_input);
    return 0;
}
