"""Pins against the reference's own committed artifacts (tests/golden, data harvested from
/root/reference/test_proofs — see tests/golden/README.md)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = 0xFFFFFFFF00000001


def _load_kat():
    toks = open(os.path.join(GOLD, "merkle_kat_ram.txt")).read().split()
    n = int(toks[0])
    vals = np.array([int(x) for x in toks[1:]], dtype=np.uint64)
    return vals[: 4 * n].reshape(n, 4), vals[4 * n:].reshape(16, 4)


def test_fixture_shapes_and_canonical_values():
    nodes, cap = _load_kat()
    assert nodes.shape[0] >= 30 and cap.shape == (16, 4)
    assert int(nodes.max()) < P and int(cap.max()) < P
    pis = json.load(open(os.path.join(GOLD, "reference_public_inputs.json")))
    assert len(pis) == 18 and pis["8_0"]["circuit"] == "RAMPermutation"
    assert pis["8_0"]["public_inputs"] == [109708311973601377, 8419656762706556756, 7577600993993766672, 4894740677854192130]


@pytest.mark.xfail(reason="Poseidon2 linear layers are restated from memory of the absent era-boojum crate and do not "
                          "reproduce the reference's Merkle nodes yet: 'parity unpinned' (DESIGN.md section 4)", strict=True)
def test_poseidon2_merkle_node_kat(oracle):
    """Among the level-16 digests of one proof, sibling pairs must hash to a cap digest."""
    nodes, cap = _load_kat()
    capset = {tuple(int(x) for x in c) for c in cap}
    hits = 0
    for a in nodes:
        for b in nodes:
            if a is not b and tuple(int(x) for x in oracle.hash_node(a, b)) in capset:
                hits += 1
    assert hits >= 15


@pytest.mark.xfail(reason="same cause as the Merkle-node KAT: the Poseidon2 restatement is not pinned yet", strict=True)
def test_poseidon2_fri_leaf_kat(oracle):
    """The last FRI oracle of a proof has 16 leaves of 8 elements and a 16-entry cap, so hash_into_leaf(leaf) — one
    permutation of (leaf || 0000), first four words — must itself be a cap entry."""
    kat = json.load(open(os.path.join(GOLD, "fri_leaf_kat_ram.json")))
    capset = {tuple(c) for c in kat["cap"]}
    hits = 0
    for leaf in kat["leaves"]:
        s = np.zeros(12, np.uint64)
        s[:8] = np.array(leaf, dtype=np.uint64)
        if tuple(int(x) for x in oracle.poseidon2(s)[:4]) in capset:
            hits += 1
    assert hits == len(kat["leaves"])


@pytest.mark.xfail(reason="same cause as the Merkle-node KAT (DESIGN.md section 4, round-2 search log)", strict=True)
def test_poseidon2_exact_sibling_pair_kat(oracle):
    """21 exact (left, right) sibling pairs of one proof: each must hash, in one of the two orders, to one of the 31 known
    parents — no search over pairs involved."""
    kat = json.load(open(os.path.join(GOLD, "merkle_pair_kat_ram.json")))
    parents = {tuple(p) for p in kat["parents"]}
    hits = 0
    for a, b in kat["pairs"]:
        a, b = np.array(a, np.uint64), np.array(b, np.uint64)
        if tuple(int(x) for x in oracle.hash_node(a, b)) in parents or tuple(int(x) for x in oracle.hash_node(b, a)) in parents:
            hits += 1
    assert hits == len(kat["pairs"]) == 21


@pytest.mark.xfail(reason="same cause as the Merkle-node KAT (DESIGN.md section 4, round-2 search log)", strict=True)
def test_poseidon2_full_witness_path_kat(oracle):
    """one whole witness_query path: hash_into_leaf of the 150 leaf elements, 17 levels up (either side at every level, the
    index bits being transcript-derived), must reach a witness_oracle_cap entry"""
    kat = json.load(open(os.path.join(GOLD, "witness_path_kat_ram.json")))
    cap = {tuple(c) for c in kat["cap"]}
    cur = {tuple(int(x) for x in oracle.hash_leaf(np.array(kat["leaf_elements"], np.uint64)))}
    for sib in kat["proof"][:12]:  # 2^12 candidates are enough to see whether any prefix survives; the full walk is 2^17
        s = np.array(sib, np.uint64)
        cur = {tuple(int(x) for x in h) for c in cur for h in (oracle.hash_node(np.array(c, np.uint64), s), oracle.hash_node(s, np.array(c, np.uint64)))}
        if len(cur) > 4096:
            break
    # with the right hash exactly one candidate per level is the real node; finish the walk from all candidates
    for sib in kat["proof"][12:]:
        s = np.array(sib, np.uint64)
        cur = {tuple(int(x) for x in h) for c in cur for h in (oracle.hash_node(np.array(c, np.uint64), s), oracle.hash_node(s, np.array(c, np.uint64)))}
    assert cur & cap
