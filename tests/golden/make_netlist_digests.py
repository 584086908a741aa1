"""Regenerates tests/golden/netlist_trace_digests.json: SHA-256 digests of small oracle-synthesized traces of the netlist
circuits (types 5, 13, 6, 3; "zkw trace v4") on fixed seeds. Types 5, 6, 3 and 13 include their queue section (Poseidon2 rows of the
pops / pushes, include/zkw_netlist_queue.h) and — since round 5 — their closed-form section (include/zkw_netlist_closed_form.h: the words of the
instance record, the commitment sponges, the public input they yield), so the digests also pin the permutation and the encodings.
Run from the repository root:  python tests/golden/make_netlist_digests.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from era_zkevm_test_harness_amd import synthetic  # noqa: E402
from oracle import pyoracle as o  # noqa: E402

N_ROWS = 1 << 18  # the stacked Keccak tables alone are 132 096 rows
ZERO_PI = np.zeros(4, np.uint64)


def digest(trace):
    return hashlib.sha256(np.ascontiguousarray(trace).tobytes()).hexdigest()


def cases():
    out = {}
    for kind, name, cap, synth in ((0, "keccak256_round_function", 6, o.keccak_round_synthesize), (1, "sha256_round_function", 7, o.sha256_round_synthesize)):
        req, mq = synthetic.precompile_trace(kind, 9, seed=3, max_rounds=4)
        tails = o.queue_push_chain_log(o.encode_log_queries(req))[1]
        w = o.precompile_build(kind, req, tails, mq, cap, np.zeros(1, o.QUEUE_STATE12))
        for i in range(w["instances"].size):
            out[f"{name}/capacity{cap}/instance{i}"] = digest(synth(w, i, cap, N_ROWS, public_input=ZERO_PI))
    from oracle import block as ob
    b = synthetic.block_after_vm(seed=2)
    a = ob.create_artifacts_after_vm(b, {ob.CODE_DECOMMITTER: 7})
    w = a["witnesses"]["code_decommitter"]
    for i in range(w["instances"].size):
        out[f"code_decommitter/capacity7/instance{i}"] = digest(o.code_decommitter_synthesize(w, i, 7, N_ROWS, public_input=ZERO_PI))
    q = synthetic.mixed_log_queue(36, seed=8)[:7]
    tr, inst, pi = o.linear_hasher_synthesize(q, o.linear_hasher_queue_state(q), 20, N_ROWS)
    g = o.nl_geometry(13)
    tr[:4, o.linear_hasher_cycles(20) * g["rows_per_cycle"] + 2 * -(-200 // g["general"])] = 0  # the Poseidon2-dependent public input
    out["linear_hasher/capacity20/7messages"] = digest(tr)
    return out


if __name__ == "__main__":
    o.build()
    path = os.path.join(ROOT, "tests", "golden", "netlist_trace_digests.json")
    json.dump(cases(), open(path, "w"), indent=1, sort_keys=True)
    print(open(path).read())
