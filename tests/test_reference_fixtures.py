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


def test_poseidon2_fri_leaf_kat(oracle):
    """The last FRI oracle of a proof has 16 leaves of 8 elements and a 16-entry cap, so hash_into_leaf(leaf) — one
    permutation of (leaf || 0000), first four words — must itself be a cap entry."""
    kat = json.load(open(os.path.join(GOLD, "fri_leaf_kat_ram.json")))
    capset = {tuple(c) for c in kat["cap"]}
    hit = {tuple(int(x) for x in oracle.poseidon2(np.array(list(leaf) + [0] * 4, np.uint64))[:4]) for leaf in kat["leaves"]}
    assert hit == capset  # a bijection: 16 distinct leaves, 16 cap entries


def _node(oracle, a, b):
    return tuple(int(x) for x in oracle.hash_node(np.array(a, np.uint64), np.array(b, np.uint64)))


def test_poseidon2_exact_sibling_pair_kat(oracle):
    """21 exact (left, right) sibling pairs of one proof: each hashes, in one of the two orders, to its parent Y, and
    H(Y, uncle) in one of the two orders is a cap entry — two levels, no search over pairs. 20 of the parents are also
    seen directly (as some query's last path element)."""
    kat = json.load(open(os.path.join(GOLD, "merkle_pair_kat_ram.json")))
    parents = {tuple(p) for p in kat["parents"]}
    cap = {tuple(c) for c in kat["cap"]}
    seen = 0
    assert len(kat["pairs"]) == 21
    for e in kat["pairs"]:
        (a, b), u = e["pair"], e["uncle"]
        ys = [y for y in (_node(oracle, a, b), _node(oracle, b, a)) if _node(oracle, y, u) in cap or _node(oracle, u, y) in cap]
        assert len(ys) == 1
        seen += ys[0] in parents
    assert seen == 20


def _leaf(oracle, els):
    return tuple(int(x) for x in oracle.hash_leaf(np.array(els, np.uint64)))


def _walk(oracle, leaf_elements, proof, idx):
    cur = _leaf(oracle, leaf_elements)
    for sib in proof:
        cur = _node(oracle, sib, cur) if idx & 1 else _node(oracle, cur, sib)
        idx >>= 1
    return list(cur), idx


def test_poseidon2_whole_query_paths(oracle):
    """Whole Merkle paths of the reference's committed proofs (tests/golden/make_reference_kats.py): witness (150-160
    elements per leaf), stage-2, quotient and — where the committed VK is the proof's — the setup oracle, whose root is
    `setup_merkle_tree_cap` of setup/*/vk_N.json; then the six FRI oracles at the folded indices. Every digest is
    recomputed: multi-permutation leaf sponges with a zero-padded last chunk, 2..17 node levels, the cap lookup."""
    kats = json.load(open(os.path.join(GOLD, "reference_merkle_paths_kat.json")))
    n_paths = 0
    for k in kats:
        for q in k["queries"]:
            for name, o in q["oracles"].items():
                top, ci = _walk(oracle, o["leaf_elements"], o["proof"], q["index"])
                assert top == k["caps"][name][ci], (k["proof"], q["query"], name)
                n_paths += 1
            idx = q["index"]
            for lvl, o in enumerate(q["fri"]):
                idx >>= 3 if len(o["leaf_elements"]) == 16 else 2
                top, ci = _walk(oracle, o["leaf_elements"], o["proof"], idx)
                assert top == k["fri_caps"][lvl][ci], (k["proof"], q["query"], "fri", lvl)
                n_paths += 1
    assert n_paths == 7 * 6 + 5 * 4 + 2 * 3
    assert sum("setup_query" in k["caps"] for k in kats) == 4


def test_poseidon2_full_witness_path_kat(oracle):
    """one whole witness_query path WITHOUT an index hint: hash_into_leaf of the 150 leaf elements, 17 levels up trying
    either side at every level (the index bits are transcript-derived), must reach a witness_oracle_cap entry"""
    kat = json.load(open(os.path.join(GOLD, "witness_path_kat_ram.json")))
    cap = {tuple(c) for c in kat["cap"]}
    cur = {_leaf(oracle, kat["leaf_elements"])}
    for sib in kat["proof"][:10]:
        cur = {h for c in cur for h in (_node(oracle, c, sib), _node(oracle, sib, c))}
    # 2^10 candidates; finish each greedily is impossible without the index, so finish the walk for all (2^17 digests)
    for sib in kat["proof"][10:]:
        cur = {h for c in cur for h in (_node(oracle, c, sib), _node(oracle, sib, c))}
    assert len(cur & cap) == 1
