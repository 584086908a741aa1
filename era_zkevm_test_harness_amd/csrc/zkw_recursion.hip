// zkw_recursion.hip — witnesses of the recursion layer that sits on top of the base-layer instances (C++ host code over
// include/zkw.h; the hashing runs in the library's HIP kernels: k_encode_recursion, the queue-chain kernel, k_commit_encodings).
//
// Counterpart of src/witness/recursive_aggregation.rs:
//   compute_encodable_item_from_witness::<AllocatedVerificationKey>   :45-68, :184-206   -> zkw_vk_commitment
//   compute_leaf_params                                               :163-216           -> zkw_compute_leaf_params
//   compute_leaf_vks_and_params_commitment                            :218-240           -> zkw_leaf_vks_and_params_commitment
//   create_leaf_witnesses                                             :71-161            -> zkw_create_leaf_witnesses
//   create_node_witnesses                                             :270-421           -> zkw_create_node_witnesses
// The encodings (`CircuitVarLengthEncodable`, absent crates) are pinned by the reference's committed leaf proofs: the public
// input of leaf_layer_proof_{6,10,15}_0.json is reproduced from vk_{4,8,13}.json, recursion_layer/vk_{6,10,15}.json and the
// public inputs of the base proofs they aggregate (tests/golden/leaf_layer_kat.json).
//   AllocatedVerificationKey  -> the setup_merkle_tree_cap digests, flat (cap_size x 4)
//   RecursionLeafParameters   -> [circuit_type, basic_circuit_vk_commitment(4), leaf_layer_vk_commitment(4)]
//   QueueState<12>            -> [head(12), tail(12), length]
//   RecursionLeafInput        -> [params(9), queue_state(25)]
//   RecursionNodeInput        -> [branch_circuit_type, leaf_layer_parameters(13 x 9), node_layer_vk_commitment(4), queue_state(25)]
#include <cstring>
#include <vector>

#include "../../include/zkw.h"
#include "zkw_internal.h"

namespace {

size_t put_params(const zkw_leaf_params& p, uint64_t* o) {
    o[0] = p.circuit_type;
    memcpy(o + 1, p.basic_circuit_vk_commitment, 32);
    memcpy(o + 5, p.leaf_layer_vk_commitment, 32);
    return 9;
}
size_t put_queue(const zkw_queue_state12& q, uint64_t* o) {
    memcpy(o, q.head, 96);
    memcpy(o + 12, q.tail, 96);
    o[24] = q.length;
    return 25;
}
}  // namespace

extern "C" int zkw_vk_commitment(zkw_ctx* ctx, const uint64_t* setup_merkle_tree_cap, size_t cap_size, uint64_t out[4]) {
    if (!ctx || !setup_merkle_tree_cap || !out || cap_size == 0) return zkw_fail(ZKW_ERR_INVALID, "zkw_vk_commitment: bad argument");
    return zkw_commit_encodings(ctx, setup_merkle_tree_cap, 1, (uint32_t)(4 * cap_size), out);
}

extern "C" int zkw_compute_leaf_params(zkw_ctx* ctx, uint8_t circuit_type, const uint64_t* base_layer_cap, const uint64_t* leaf_layer_cap,
                                       size_t cap_size, zkw_leaf_params* out) {
    if (!ctx || !base_layer_cap || !leaf_layer_cap || !out || cap_size == 0) return zkw_fail(ZKW_ERR_INVALID, "zkw_compute_leaf_params: bad argument");
    if (circuit_type < 1 || circuit_type > ZKW_NUM_BASE_LAYER_CIRCUITS) return zkw_fail(ZKW_ERR_INVALID, "zkw_compute_leaf_params: circuit type %u", circuit_type);
    // the two commitments in one launch (two items of equal length)
    std::vector<uint64_t> enc(8 * cap_size);
    memcpy(enc.data(), base_layer_cap, 32 * cap_size);
    memcpy(enc.data() + 4 * cap_size, leaf_layer_cap, 32 * cap_size);
    uint64_t c[8];
    const int rc = zkw_commit_encodings(ctx, enc.data(), 2, (uint32_t)(4 * cap_size), c);
    if (rc != ZKW_OK) return rc;
    out->circuit_type = circuit_type;
    memcpy(out->basic_circuit_vk_commitment, c, 32);
    memcpy(out->leaf_layer_vk_commitment, c + 4, 32);
    return ZKW_OK;
}

extern "C" int zkw_leaf_vks_and_params_commitment(zkw_ctx* ctx, const zkw_leaf_params* leaf_params /* [13] */, uint64_t out[4]) {
    if (!ctx || !leaf_params || !out) return zkw_fail(ZKW_ERR_INVALID, "zkw_leaf_vks_and_params_commitment: null argument");
    uint64_t enc[9 * ZKW_NUM_BASE_LAYER_CIRCUITS];
    for (int t = 0; t < ZKW_NUM_BASE_LAYER_CIRCUITS; t++) put_params(leaf_params[t], enc + 9 * t);
    return zkw_commit_encodings(ctx, enc, 1, 9 * ZKW_NUM_BASE_LAYER_CIRCUITS, out);
}

extern "C" int zkw_create_leaf_witnesses(zkw_ctx* ctx, const zkw_leaf_params* params, const uint64_t* public_inputs, size_t n,
                                         const uint64_t* queue_tail_in, uint64_t* enc, uint64_t* states, zkw_queue_state12* leaf_states,
                                         uint64_t* leaf_public_inputs, size_t max_leaves, size_t* n_leaves) {
    if (!ctx || !params || !n_leaves || (n && (!public_inputs || !enc || !states))) return zkw_fail(ZKW_ERR_INVALID, "zkw_create_leaf_witnesses: null argument");
    int rc = ZKW_OK;
    if (n && (rc = zkw_encode_recursion_requests(ctx, params->circuit_type, public_inputs, n, enc)) != ZKW_OK) return rc;
    if (n && (rc = zkw_queue_push_chain_full(ctx, enc, n, queue_tail_in, states)) != ZKW_OK) return rc;
    if ((rc = zkw_recursion_queue_split(states, n, ZKW_RECURSION_ARITY, leaf_states, max_leaves, n_leaves)) != ZKW_OK) return rc;
    const size_t leaves = *n_leaves;
    if (leaves == 0) return ZKW_OK;
    if (queue_tail_in) memcpy(leaf_states[0].head, queue_tail_in, 96);  // a queue continued from a non-empty state: its head is that state
    if (!leaf_public_inputs) return ZKW_OK;
    std::vector<uint64_t> in(34 * leaves);
    for (size_t k = 0; k < leaves; k++) {
        uint64_t* o = in.data() + 34 * k;
        o += put_params(*params, o);
        put_queue(leaf_states[k], o);
    }
    return zkw_commit_encodings(ctx, in.data(), leaves, 34, leaf_public_inputs);
}

extern "C" int zkw_create_node_witnesses(zkw_ctx* ctx, uint8_t branch_circuit_type, const zkw_leaf_params* leaf_layer_params /* [13] */,
                                         const uint64_t node_layer_vk_commitment[4], const zkw_queue_state12* chunks, size_t n_chunks,
                                         zkw_queue_state12* node_states, zkw_queue_tail12* split_points, uint64_t* node_public_inputs,
                                         size_t max_nodes, size_t* n_nodes) {
    if (!ctx || !leaf_layer_params || !node_layer_vk_commitment || !n_nodes || !chunks || n_chunks == 0)
        return zkw_fail(ZKW_ERR_INVALID, "zkw_create_node_witnesses: bad argument");  // the reference asserts chunks.len() > 0 (:291)
    const size_t A = ZKW_RECURSION_ARITY, nodes = (n_chunks + A - 1) / A;
    *n_nodes = nodes;
    if (nodes > max_nodes || !node_states || !split_points) return zkw_fail(ZKW_ERR_INVALID, "zkw_create_node_witnesses: %zu nodes, room for %zu", nodes, max_nodes);
    for (size_t k = 0; k < nodes; k++) {
        const size_t first = k * A, end = first + A < n_chunks ? first + A : n_chunks;
        zkw_queue_state12 q = chunks[first];
        zkw_queue_tail12* sp = split_points + k * (A - 1);
        size_t n_sp = 0;
        for (size_t c = first; c < end; c++) {
            if (chunks[c].length == 0) return zkw_fail(ZKW_ERR_INVALID, "zkw_create_node_witnesses: chunk %zu is empty", c);  // :330-332
            if (c > first) {  // RecursionQueueSimulator::merge (circuit_encodings/src/lib.rs:365-377): the tails must chain
                if (memcmp(q.tail, chunks[c].head, 96) != 0) return zkw_fail(ZKW_ERR_CHECK_FAILED, "zkw_create_node_witnesses: chunk %zu does not continue chunk %zu", c, c - 1);
                memcpy(q.tail, chunks[c].tail, 96);
                q.length += chunks[c].length;
            }
            if (n_sp < A - 1) {  // N chunks need N - 1 split points: the 32nd is dropped (:375-377)
                memcpy(sp[n_sp].tail, chunks[c].tail, 96);
                sp[n_sp].length = chunks[c].length;
                sp[n_sp]._pad = 0;
                n_sp++;
            }
        }
        for (; n_sp < A - 1; n_sp++) {  // padding: the merged queue's tail with length 0 (:379-384)
            memcpy(sp[n_sp].tail, q.tail, 96);
            sp[n_sp].length = 0;
            sp[n_sp]._pad = 0;
        }
        node_states[k] = q;
    }
    if (!node_public_inputs) return ZKW_OK;
    const size_t L = 1 + 9 * ZKW_NUM_BASE_LAYER_CIRCUITS + 4 + 25;
    std::vector<uint64_t> in(L * nodes);
    for (size_t k = 0; k < nodes; k++) {
        uint64_t* o = in.data() + L * k;
        *o++ = branch_circuit_type;
        for (int t = 0; t < ZKW_NUM_BASE_LAYER_CIRCUITS; t++) o += put_params(leaf_layer_params[t], o);
        memcpy(o, node_layer_vk_commitment, 32);
        o += 4;
        put_queue(node_states[k], o);
    }
    return zkw_commit_encodings(ctx, in.data(), nodes, (uint32_t)L, node_public_inputs);
}
