#!/usr/bin/env python3
"""Regenerates the reference-held Poseidon2 known answers (run in the authoring container only: reads /root/reference).

Every proof under test_proofs/ holds Merkle paths hashed with `GoldilocksPoseidon2Sponge<AbsorptionModeOverwrite>`
(src/prover_utils.rs:43): leaf = sponge over the leaf elements (rate 8, overwrite, zero-padded last chunk), node = one
permutation of (left || right || 0000), digest = first four state words. Outputs (data only):

  merkle_pair_kat_ram.json      21 exact sibling pairs below the cap of one oracle, their uncle and the 16-entry cap
  reference_merkle_paths_kat.json  whole query paths (leaf elements, siblings, cap) of the witness / stage-2 / quotient /
                                setup oracles and of the six FRI oracles, from four proofs; the setup paths end in the
                                `setup_merkle_tree_cap` of the matching setup/base_layer/vk_N.json. The query index is found
                                here with the oracle and stored as a hint; the test re-derives every digest.
"""
import collections
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as orc  # noqa: E402

REF = "/root/reference"


def node(a, b):
    return tuple(int(x) for x in orc.hash_node(np.array(a, np.uint64), np.array(b, np.uint64)))


def leaf(els):
    return tuple(int(x) for x in orc.hash_leaf(np.array(els, np.uint64)))


def find_index(leaf_elements, proof, cap):
    """depth-first over left/right at every level; the real path is the one ending in a cap entry"""
    cap = [tuple(c) for c in cap]
    cur = {leaf(leaf_elements): 0}
    for lvl, sib in enumerate(proof):
        nxt = {}
        for c, idx in cur.items():
            nxt[node(c, sib)] = idx
            nxt[node(sib, c)] = idx | (1 << lvl)
        cur = nxt
    hits = [(idx, cap.index(c)) for c, idx in cur.items() if c in cap]
    assert len(hits) == 1, hits
    idx, cap_idx = hits[0]
    return idx | (cap_idx << len(proof))


def walk(leaf_elements, proof, idx):
    cur = leaf(leaf_elements)
    for sib in proof:
        cur = node(sib, cur) if idx & 1 else node(cur, sib)
        idx >>= 1
    return cur, idx


def paths_of(proof_file, vk_file, queries, with_setup=True):
    d = json.load(open(os.path.join(REF, proof_file)))
    name = list(d)[0]
    d = d[name]
    vk = json.load(open(os.path.join(REF, vk_file)))
    vk = vk[list(vk)[0]]
    caps = {"witness_query": d["witness_oracle_cap"], "stage_2_query": d["stage_2_oracle_cap"],
            "quotient_query": d["quotient_oracle_cap"], "setup_query": vk["setup_merkle_tree_cap"]}
    if not with_setup:  # setup/*/vk_N.json was regenerated after this proof was made: its cap is not this proof's
        del caps["setup_query"]
    out = {"proof": proof_file, "vk": vk_file, "circuit": name, "caps": caps,
           "fri_caps": [d["fri_base_oracle_cap"]] + d["fri_intermediate_oracles_caps"], "queries": []}
    for qi in queries:
        q = d["queries_per_fri_repetition"][qi]
        idx = find_index(q["quotient_query"]["leaf_elements"], q["quotient_query"]["proof"], caps["quotient_query"])
        rec = {"query": qi, "index": idx, "oracles": {}, "fri": []}
        for k, cap in caps.items():
            top, ci = walk(q[k]["leaf_elements"], q[k]["proof"], idx)
            assert list(top) == cap[ci], (proof_file, qi, k)
            rec["oracles"][k] = {"leaf_elements": q[k]["leaf_elements"], "proof": q[k]["proof"]}
        fidx = idx
        for lvl, fq in enumerate(q["fri_queries"]):
            fidx >>= 3 if len(fq["leaf_elements"]) == 16 else 2  # leaves of 8 extension elements (fold by 8), the last of 4
            top, ci = walk(fq["leaf_elements"], fq["proof"], fidx)
            assert list(top) == out["fri_caps"][lvl][ci], (proof_file, qi, "fri", lvl)
            rec["fri"].append({"leaf_elements": fq["leaf_elements"], "proof": fq["proof"]})
        out["queries"].append(rec)
    return out


def pair_kat():
    d = json.load(open(os.path.join(REF, "test_proofs/base_layer/basic_circuit_proof_8_0.json")))["RAMPermutation"]
    g = collections.defaultdict(set)
    for q in d["queries_per_fri_repetition"]:
        p = q["quotient_query"]["proof"]
        g[tuple(p[16])].add(tuple(p[15]))
    pairs = [{"pair": [list(x) for x in sorted(v)], "uncle": list(k)} for k, v in sorted(g.items()) if len(v) == 2]
    return {"source": "test_proofs/base_layer/basic_circuit_proof_8_0.json, quotient_query paths",
            "statement": "for every entry, Y = H(pair[0] || pair[1]) or H(pair[1] || pair[0]) and H(Y || uncle) or "
                         "H(uncle || Y) is an entry of `cap`; Y is one of `parents` (the 31 level-16 nodes that occur as the "
                         "last path element of some query) for all but the pair under the one level-16 node no query lists",
            "pairs": pairs, "parents": sorted(list(k) for k in g), "cap": d["quotient_oracle_cap"]}


def leaf_layer_kat():
    """what the leaf-layer public inputs are computed from, for the three circuit types whose committed VKs are the
    ones their proofs were made with (setup paths of base AND leaf proof end in the committed caps)"""
    import glob
    out = []
    for base_t in (4, 8, 13):
        leaf_t = base_t + 2
        cap = lambda f: (lambda v: v[list(v)[0]]["setup_merkle_tree_cap"])(json.load(open(os.path.join(REF, f))))
        pis = []
        for f in sorted(glob.glob(os.path.join(REF, f"test_proofs/base_layer/basic_circuit_proof_{base_t}_*.json"))):
            d = json.load(open(f))
            pis.append(d[list(d)[0]]["public_inputs"])
        lp = json.load(open(os.path.join(REF, f"test_proofs/recursion_layer/leaf_layer_proof_{leaf_t}_0.json")))
        out.append({"base_circuit_type": base_t, "leaf_circuit_type": leaf_t,
                    "base_vk_cap": cap(f"setup/base_layer/vk_{base_t}.json"), "leaf_vk_cap": cap(f"setup/recursion_layer/vk_{leaf_t}.json"),
                    "base_public_inputs": pis, "leaf_public_input": lp[list(lp)[0]]["public_inputs"]})
    return {"source": "setup/base_layer/vk_N.json, setup/recursion_layer/vk_{N+2}.json, test_proofs/base_layer/basic_circuit_proof_N_*.json, "
                      "test_proofs/recursion_layer/leaf_layer_proof_{N+2}_0.json for N = 4, 8, 13",
            "statement": "leaf_public_input = commit(RecursionLeafInput{params = compute_leaf_params(N, base vk, leaf vk), queue_state = "
                         "the recursion queue over base_public_inputs})", "cases": out}


if __name__ == "__main__":
    json.dump(leaf_layer_kat(), open(os.path.join(HERE, "leaf_layer_kat.json"), "w"))
    json.dump(pair_kat(), open(os.path.join(HERE, "merkle_pair_kat_ram.json"), "w"))
    out = [paths_of("test_proofs/base_layer/basic_circuit_proof_8_0.json", "setup/base_layer/vk_8.json", [0, 57]),
           paths_of("test_proofs/base_layer/basic_circuit_proof_4_0.json", "setup/base_layer/vk_4.json", [3]),
           paths_of("test_proofs/base_layer/basic_circuit_proof_13_0.json", "setup/base_layer/vk_13.json", [11]),
           paths_of("test_proofs/recursion_layer/leaf_layer_proof_10_0.json", "setup/recursion_layer/vk_10.json", [7]),
           # proofs older than the committed VK of their type (the setup path does not end in vk_N's cap): no setup oracle
           paths_of("test_proofs/base_layer/basic_circuit_proof_1_0.json", "setup/base_layer/vk_1.json", [5], False),
           paths_of("test_proofs/base_layer/basic_circuit_proof_5_0.json", "setup/base_layer/vk_5.json", [9], False)]
    json.dump(out, open(os.path.join(HERE, "reference_merkle_paths_kat.json"), "w"))
    print("ok", [len(o["queries"]) for o in out])
