"""Oracle restatement of src/witness/recursive_aggregation.rs against the reference's committed leaf proofs (CPU).

tests/golden/leaf_layer_kat.json (made by tests/golden/make_reference_kats.py): for base circuit types 4, 8, 13 the caps of
the base VK and of the leaf VK, the public inputs of the base proofs and the public input of the committed leaf proof.
Reproducing it pins, end to end and against the Rust reference: Poseidon2, `commit_variable_length_encodable_item` (length
specialisation, overwrite absorption, zero padding), RecursionRequest::encoding_witness (recursion_request.rs:13-28), the
full-width queue push (lib.rs:391-429), transform_sponge_like_queue_state (utils.rs:73-85), compute_leaf_params
(recursive_aggregation.rs:163-216) and the field order of RecursionLeafParameters / QueueState / RecursionLeafInput."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLD, "leaf_layer_kat.json")))


@pytest.mark.parametrize("case", KAT["cases"], ids=lambda c: f"type{c['base_circuit_type']}")
def test_leaf_public_input_of_committed_proofs(oracle, case):
    t = case["base_circuit_type"]
    params = oracle.leaf_params(t, case["base_vk_cap"], case["leaf_vk_cap"])
    assert np.array_equal(params["basic_circuit_vk_commitment"][0], oracle.vk_commitment(case["base_vk_cap"]))
    pis = np.array(case["base_public_inputs"], np.uint64)
    enc, states = oracle.recursion_queue(t, pis)
    q = np.zeros(1, oracle.QUEUE_STATE12)
    q["tail"][0] = states[-1]
    q["length"][0] = len(pis)
    assert [int(x) for x in oracle.leaf_public_input(params, q)] == case["leaf_public_input"]
    # a wrong length, a swapped commitment or a non-zero head must not reproduce it
    q2 = q.copy()
    q2["length"][0] += 1
    assert [int(x) for x in oracle.leaf_public_input(params, q2)] != case["leaf_public_input"]
    p2 = params.copy()
    p2["basic_circuit_vk_commitment"], p2["leaf_layer_vk_commitment"] = params["leaf_layer_vk_commitment"], params["basic_circuit_vk_commitment"]
    assert [int(x) for x in oracle.leaf_public_input(p2, q)] != case["leaf_public_input"]


def test_node_witness_merge_and_split_points(oracle):
    """create_node_witnesses (:270-421): 3 leaves of one type merge into one node; split points = the leaves' tails, padded to 31
    with (merged tail, 0); the node's public input commits to 1 + 13 x 9 + 4 + 25 words"""
    rng = np.random.default_rng(5)
    pis = rng.integers(0, 2**63, (70, 4), dtype=np.uint64)
    enc, states = oracle.recursion_queue(8, pis)
    chunks = np.zeros(3, oracle.QUEUE_STATE12)
    for k, (a, b) in enumerate(((0, 32), (32, 64), (64, 70))):
        if a:
            chunks["head"][k] = states[a - 1]
        chunks["tail"][k] = states[b - 1]
        chunks["length"][k] = b - a
    params = np.zeros(13, oracle.LEAF_PARAMS)
    params["circuit_type"] = np.arange(1, 14)
    params["basic_circuit_vk_commitment"] = rng.integers(0, 2**63, (13, 4), dtype=np.uint64)
    params["leaf_layer_vk_commitment"] = rng.integers(0, 2**63, (13, 4), dtype=np.uint64)
    nvk = rng.integers(0, 2**63, 4, dtype=np.uint64)
    st, sp, pi = oracle.node_witness(10, params, nvk, chunks)
    assert int(st["length"]) == 70 and np.array_equal(st["tail"], states[-1]) and not st["head"].any()
    assert [int(x) for x in sp["length"][:4]] == [32, 32, 6, 0]
    assert np.array_equal(sp["tail"][2], states[-1]) and np.array_equal(sp["tail"][30], states[-1])
    u = lambda x: np.atleast_1d(np.asarray(x, dtype=np.uint64))
    flat = np.concatenate([u(10)] + [np.concatenate([u(p["circuit_type"]), p["basic_circuit_vk_commitment"], p["leaf_layer_vk_commitment"]])
                                      for p in params] + [nvk, st["head"], st["tail"], u(70)])
    assert flat.size == 147
    assert np.array_equal(pi, oracle.commit_var_length(flat))
    with pytest.raises(ValueError):
        oracle.node_witness(10, params, nvk, chunks[[0, 2]])  # does not chain
