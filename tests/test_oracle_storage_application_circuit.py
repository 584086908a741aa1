"""CPU: the StorageApplication circuit (type 10) in "zkw trace v4" (tools/gen_storage_application_circuit.py, oracle/netlist_circuit.c):
  * the layout has the REFERENCE's geometry and table volume (storage_apply.rs:28-39,124-140; `total_tables_len` of vk_10.json) and
    holds the reference's capacity (33 tree queries, geometry_config.rs) in 2^20 rows;
  * the walks of an instance end in the roots the tree (oracle/storage_application.c = src/witness/tree/mod.rs) reports, the leaf
    cycles in hashlib's Blake2s-256 of index || value;
  * filled traces satisfy the checker; tampering is caught with the right violation kind."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

from sap_case import be32, storage_application_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ROWS = 1 << 18  # the stacked tables alone need 132 352 rows
WALK = 257


def _gen():
    spec = importlib.util.spec_from_file_location("gen_sa", os.path.join(ROOT, "tools", "gen_storage_application_circuit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_layout_is_the_reference_geometry_table_volume_and_capacity(oracle):
    vk = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vk_parameters.json")))["10"]
    lp = vk["lookup_parameters"]["UseSpecializedColumnsWithTableIdAsConstant"]
    g = oracle.nl_geometry(10)
    assert g["general"] == vk["parameters"]["num_columns_under_copy_permutation"] == 60
    assert (g["width"], g["lookups_per_row"]) == (lp["width"], lp["num_repetitions"]) == (3, 26) and lp["share_table_id"]
    assert g["cols"] == g["general"] + g["width"] * g["lookups_per_row"] + 1  # ONE multiplicity column
    assert g["table_rows"] == vk["total_tables_len"] == 132352  # Xor8, And8, ByteSplit<1, 2, 3, 4, 7>
    assert ((1 << 20) - 8) // (g["rows_per_cycle"] * WALK) >= 33  # cycles_per_storage_application tree queries in 2^20 rows


def test_committed_spec_is_current_and_self_checked(tmp_path):
    mod = _gen()
    spec, path = mod.emit(str(tmp_path / "spec.h"))  # evaluates the netlist against hashlib.blake2s before writing
    assert open(path).read() == open(os.path.join(ROOT, "include", "zkw_storage_application_circuit_spec.h")).read(), \
        "run tools/gen_storage_application_circuit.py"
    st = spec.step_types[0]
    last = 0
    for pos, (j, t) in enumerate(st.slots):  # tables ascend and start a row (what the multiplicity pass relies on)
        assert t.id >= last
        if t.id != last:
            assert pos % spec.R == 0
        last = t.id
    assert not st.unchecked or all(v in [c for g_ in st.gates for c in g_[2]] for v in st.unchecked)


def _state_hash_before(t, spec, cycle):
    """the running hash (cycle state bytes 0..31) as cycle `cycle` copies it: input 0 of the select lookups XOR8(cyc[k], free[k]),
    the netlist's ops 3k (tools/gen_storage_application_circuit.py build())"""
    st = spec.step_types[0]
    rpc = spec.rows_per_cycle()
    out = []
    for k in range(32):
        tb, ins, _, pos = st.ops[3 * k]
        assert tb.name == "XOR8" and ins[0] == ("cyc", k) and ins[1] == ("free", k)
        out.append(int(t[spec.G + 3 * (pos % spec.R), cycle * rpc + 1 + pos // spec.R]))
    return bytes(out)


@pytest.fixture(scope="module")
def spec():
    return _gen().make_spec()


@pytest.fixture(scope="module")
def case(oracle):
    q, tails, tree, idx, paths = storage_application_case(oracle, 7, seed=5)
    q["rw_flag"][:] = [1, 0, 1, 1, 0, 1, 0]  # 4 writes + 3 reads = 11 walks
    ro = q["rw_flag"] == 0
    q["written_value"][ro] = q["read_value"][ro]
    root0 = bytes(tree.root)
    o = oracle.storage_application_build(tree, q, tails, 5)  # capacity 5: cuts after >= 4 tree queries
    return q, o, root0


def test_walks_end_in_the_tree_roots(oracle, case, spec):
    q, o, root0 = case
    g = oracle.nl_geometry(10)
    rpc = g["rows_per_cycle"]
    assert o["instances"].size >= 3
    prev_root = root0
    for k, inst in enumerate(o["instances"]):
        t = oracle.storage_application_synthesize(o, q, k, 5, N_ROWS)
        assert oracle.storage_application_check(t, 5) == (0, (0, 0, 0))
        first, n = int(inst["first_item"]), int(inst["num_items"])
        walks = [(first + i, ph) for i in range(n) for ph in range(2 if q["rw_flag"][first + i] else 1)]
        assert 0 < len(walks) <= 5
        hdr = t[:4, np.arange(5 * WALK) * rpc]
        assert hdr[0].tolist() == [1 if c % WALK == 0 and c < len(walks) * WALK else 0 for c in range(5 * WALK)]  # reset = leaf cycles
        assert hdr[1].tolist() == [0 if c < len(walks) * WALK else 1 for c in range(5 * WALK)]                      # idle
        assert hdr[2].tolist() == [40 if r else 64 for r in hdr[0].tolist()]                                         # t = message length
        # the state before the cycle after a walk's last level = the root the walk arrives at: BND_IN row of the NEXT cycle is
        # not in the trace, but the lookups of that cycle copy it: read it from the first XOR8 inputs (x = cyc[k]) of the cycle
        for w, (i, ph) in enumerate(walks):
            nxt = (w + 1) * WALK
            if nxt < 5 * WALK:
                got = _state_hash_before(t, spec, nxt)
            else:  # the last walk of a full instance: BND_OUT
                got = bytes(int(x) for x in t[:32, 5 * WALK * rpc + 2])
            want = bytes(o["roots"][i]) if (ph == 1 or not q["rw_flag"][i]) else prev_root
            assert got == want, (k, w, i, ph)
            if ph == 1 or not q["rw_flag"][i]:
                prev_root = bytes(o["roots"][i])
        # leaf cycle of the instance's first walk: hashlib's Blake2s-256(index_be || value_be)
        i, ph = walks[0]
        idx = int(o["leaf_indexes"][i])
        leaf = hashlib.blake2s(idx.to_bytes(8, "big") + be32(q["read_value"][i])).digest()
        assert _state_hash_before(t, spec, 1) == leaf


def test_tampering_is_caught(oracle, case, spec):
    q, o, _ = case
    st = spec.step_types[0]
    g = oracle.nl_geometry(10)
    G, rpc = g["general"], g["rows_per_cycle"]
    t = oracle.storage_application_synthesize(o, q, 0, 5, N_ROWS)
    assert int(t[g["cols"] - 1].sum()) == 5 * WALK * len(st.slots)  # every lookup slot counted once
    base = 3 * rpc  # a level cycle
    gi = next(i for i, g_ in enumerate(st.gates) if len(g_[2]) == 5)  # an addition: 4 byte digits + a carry
    grow, gcol = st.gate_pos[gi]
    nk = len(st.gates[gi][0])
    cells = {"lookup_out": ((G + 2, base + 1), 1), "lookup_in_range": ((G, base + 1), 1), "reset": ((0, base), 2),  # (the 255 * reset gate copies the bit: the copy (2) is reported before the header (3)) "mask0": ((2, base), 3),
             "hdr_lookup": ((G + 1, base), 6), "mult": ((g["cols"] - 1, 7), 5), "bnd_out": ((3, 5 * WALK * rpc + 2), 2),  # (a root byte after the last cycle: the closed-form tie to the FSM output objects first, kind 2, then the boundary rule, 4)
             "below": ((G // 2, oracle.nlcf_geometry(10, 5 * WALK)["rows_used"] + 3), 6), "gate_known": ((gcol, base + grow), 2), "gate_digit": ((gcol + nk, base + grow), 2)}
    for name, ((col, row), kind) in cells.items():
        bad = t.copy()
        bad[col, row] = 300 if name == "lookup_in_range" else 2 if name == "reset" else bad[col, row] + 1
        n, first = oracle.storage_application_check(bad, 5)
        assert n > 0 and first[0] == kind, (name, (col, row), n, first)
    # the carry of an addition is copied by its range-check lookup only: the gate's own sum (7) and that copy (2) catch it
    bad = t.copy()
    bad[gcol + nk + 4, base + grow] += 1
    n, first = oracle.storage_application_check(bad, 5)
    assert n >= 1 and first[0] in (2, 7)
    # a sibling byte changed consistently inside ONE lookup (a FREE cell: no copy constraint on it) derails the walk: the
    # lookups that consume that lookup's output no longer copy it
    slot = next(p for p, (j, tb) in enumerate(st.slots) if j is not None and any((not isinstance(r, tuple)) is False and r[0] == "free" and r[1] == 32 for r in st.ops[j][1]))
    col, row = G + 3 * (slot % 26), base + 1 + slot // 26
    bad = t.copy()
    ins = [int(bad[col, row]), int(bad[col + 1, row]) ^ 1]
    bad[col + 1, row], bad[col + 2, row] = ins[1], ins[0] ^ ins[1]
    n, first = oracle.storage_application_check(bad, 5)
    assert n > 0 and first[0] in (2, 5)


def test_a_wrong_path_is_another_tree(oracle, case, spec):
    q, o, _ = case
    bad = dict(o)
    bad["merkle_paths"] = o["merkle_paths"].copy()
    bad["merkle_paths"][0, 17, 3] ^= 1
    # the oracle's fill recomputes the chain itself: a wrong sibling still gives a satisfied trace of a DIFFERENT tree ...
    t = oracle.storage_application_synthesize(bad, q, 0, 5, N_ROWS)
    assert oracle.storage_application_check(t, 5)[0] == 0
    # ... whose first walk no longer arrives at the block's initial root
    good = oracle.storage_application_synthesize(o, q, 0, 5, N_ROWS)
    assert _state_hash_before(t, spec, WALK) != _state_hash_before(good, spec, WALK)
