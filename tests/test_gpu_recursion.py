"""GPU: the recursion-layer witness entry points (csrc/zkw_recursion.hip = src/witness/recursive_aggregation.rs) through the C ABI,
against the reference's committed leaf proofs (tests/golden/leaf_layer_kat.json — no oracle involved) and against the oracle on
seeded multi-leaf / multi-node queues."""
import json
import os

import numpy as np
import pytest

from era_zkevm_test_harness_amd import native as nv

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(GOLD, "leaf_layer_kat.json")))


@pytest.fixture(scope="module")
def ctx():
    c = nv.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("case", KAT["cases"], ids=lambda c: f"type{c['base_circuit_type']}")
def test_leaf_public_input_of_committed_proofs(ctx, case):
    t = case["base_circuit_type"]
    params = nv.compute_leaf_params(ctx, t, case["base_vk_cap"], case["leaf_vk_cap"])
    assert np.array_equal(params["leaf_layer_vk_commitment"][0], nv.vk_commitment(ctx, case["leaf_vk_cap"]))
    w = nv.create_leaf_witnesses(ctx, params, case["base_public_inputs"])
    assert w["leaf_states"].size == 1 and int(w["leaf_states"]["length"][0]) == len(case["base_public_inputs"])
    assert [int(x) for x in w["leaf_public_inputs"][0]] == case["leaf_public_input"]


@pytest.mark.parametrize("n", [0, 1, 32, 33, 100, 1100])
def test_leaf_and_node_witnesses_match_oracle(ctx, oracle, n):
    rng = np.random.default_rng(n)
    pis = rng.integers(0, 2**63, (n, 4), dtype=np.uint64)
    caps = rng.integers(0, 2**63, (2, 16, 4), dtype=np.uint64)
    params = nv.compute_leaf_params(ctx, 9, caps[0], caps[1])
    assert params.tobytes() == oracle.leaf_params(9, caps[0], caps[1]).tobytes()
    w = nv.create_leaf_witnesses(ctx, params, pis)
    enc, states = oracle.recursion_queue(9, pis)
    assert np.array_equal(w["enc"], enc) and np.array_equal(w["states"], states)
    assert w["leaf_states"].tobytes() == nv.recursion_queue_split(states).tobytes()
    for k, q in enumerate(w["leaf_states"]):
        assert np.array_equal(w["leaf_public_inputs"][k], oracle.leaf_public_input(params, q))
    if n == 0:
        return
    all_params = np.zeros(13, nv.LEAF_PARAMS)
    for t in range(13):
        all_params[t] = oracle.leaf_params(t + 1, rng.integers(0, 2**63, (16, 4), dtype=np.uint64), rng.integers(0, 2**63, (16, 4), dtype=np.uint64))[0]
    assert np.array_equal(nv.leaf_vks_and_params_commitment(ctx, all_params), oracle.leaf_vks_and_params_commitment(all_params))
    nvk = nv.vk_commitment(ctx, caps[1])
    nodes = nv.create_node_witnesses(ctx, 11, all_params, nvk, w["leaf_states"])
    for k in range(nodes["node_states"].size):
        st, sp, pi = oracle.node_witness(11, all_params, nvk, w["leaf_states"][32 * k:32 * k + 32])
        assert nodes["node_states"][k].tobytes() == st.tobytes()
        assert nodes["split_points"][k].tobytes() == sp.tobytes()
        assert np.array_equal(nodes["node_public_inputs"][k], pi)
    assert sum(int(x) for x in nodes["node_states"]["length"]) == n


def test_node_witnesses_reject_broken_chains(ctx):
    ch = np.zeros(2, nv.QUEUE_STATE12)
    ch["length"] = 1
    ch["tail"][0][0] = 5  # chunk 1's head (zero) is not chunk 0's tail
    p = np.zeros(13, nv.LEAF_PARAMS)
    with pytest.raises(nv.ZkwError) as e:
        nv.create_node_witnesses(ctx, 3, p, np.zeros(4, np.uint64), ch)
    assert e.value.code == nv.ERR_CHECK_FAILED
    ch["length"][1] = 0
    with pytest.raises(nv.ZkwError) as e:
        nv.create_node_witnesses(ctx, 3, p, np.zeros(4, np.uint64), ch)
    assert e.value.code == nv.ERR_INVALID
