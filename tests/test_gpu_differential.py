"""Differential suite: every witness builder driven with 200 adversarially SHAPED, seeded cases
(tests/differential_cases.py; corpus and shape statistics in tests/golden/differential_corpus.json) — queue lengths on the instance
boundaries (capacity - 1 / capacity / capacity + 1, one item, several instances), one cell touched by everything or every item its own,
rollback-dense and rollback-free logs, one hash or all-distinct hashes, empty types, timestamps up to 2^32 - 1 — GPU (through the C ABI)
against the CPU oracle, byte for byte. Where a case is not a valid queue both sides must reject it. On the oracle's side the reference's own
structural assertions are checked on every case: the two grand products of a permutation argument end equal (src/witness/utils.rs:654-660),
an empty queue's head equals its tail (circuit_encodings/src/lib.rs:252-254), consecutive instances chain (hidden_fsm_output of i ==
hidden_fsm_input of i + 1, oracle.rs:1380-1390)."""
import json
import os

import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic
from tests import differential_cases as dc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def _corpus(builder):
    with open(os.path.join(ROOT, "tests", "golden", "differential_corpus.json")) as f:
        c = json.load(f)
    rows = [r for r in c["cases"] if r["builder"] == builder]
    assert len(rows) == c["cases_per_builder"] >= 200
    return rows


def _records_equal(g, o, what):
    assert g.size == o.size, what
    assert g.tobytes() == o.tobytes(), what


def _chained(inst, what):
    """hidden_fsm_output of instance i is the hidden_fsm_input of instance i + 1; the first instance starts, the last completes"""
    if inst.size == 0:
        return
    assert int(inst["start_flag"][0]) == 1 and int(inst["completion_flag"][-1]) == 1, what
    assert not inst["start_flag"][1:].any() and not inst["completion_flag"][:-1].any(), what
    for i in range(inst.size - 1):
        assert inst["hidden_fsm_output"][i].tobytes() == inst["hidden_fsm_input"][i + 1].tobytes(), (what, i)


def _both(gpu, cpu, nv):
    """run both sides; a case one of them rejects must be rejected by the other"""
    g_err = o_err = None
    try:
        o = cpu()
    except Exception as e:  # noqa: BLE001 - the oracle signals an invalid queue by raising
        o, o_err = None, e
    try:
        g = gpu()
    except nv.ZkwError as e:
        g, g_err = None, e
    assert (g_err is None) == (o_err is None), (g_err, o_err)
    return g, o


def _each(builder):
    for row in _corpus(builder):
        c = dc.case(builder, row["seed"])
        assert dc.digest(c) == row["digest"], "the generator drifted from the committed corpus: python tests/differential_cases.py"
        yield row["seed"], c


def test_ram(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    for seed, c in _each("ram"):
        w, o = _both(lambda: ctx.compute_ram_circuit_snapshots(c["q"], c["capacity"], c["nondet"]), lambda: oracle.ram_build_instances(c["q"], c["capacity"], c["nondet"]), nv)
        if w is None:
            continue
        n = c["q"].size
        for what, key in ((nv.RAM_SORTED_QUERIES, "sorted_q"), (nv.RAM_UNSORTED_TAILS, "unsorted_tails"), (nv.RAM_SORTED_TAILS, "sorted_tails")):
            assert np.array_equal(w.get(what), o[key]), (seed, key)
        assert np.array_equal(w.get(nv.RAM_CHALLENGES)[0], o["challenges"]), seed
        assert np.array_equal(w.get(nv.RAM_LHS_Z).reshape(2, n), o["lhs_z"]) and np.array_equal(w.get(nv.RAM_RHS_Z).reshape(2, n), o["rhs_z"]), seed
        _records_equal(w.get(nv.RAM_INSTANCES), o["instances"], seed)
        compact, pi = oracle.ram_public_inputs(o["instances"])
        assert np.array_equal(w.get(nv.RAM_PUBLIC_INPUTS), pi) and np.array_equal(w.get(nv.RAM_COMPACT_FORMS), compact), seed
        assert np.array_equal(o["lhs_z"][:, -1], o["rhs_z"][:, -1]), seed  # utils.rs:654-660
        _chained(o["instances"], seed)
        w.free()


def test_decommit_sorter(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    din = np.zeros(1, oracle.QUEUE_STATE12)
    din["tail"], din["head"], din["length"] = synthetic.random_field_elements(3, (12,)), synthetic.random_field_elements(4, (12,)), 9
    for seed, c in _each("decommit_sorter"):
        d = din if c["dedup_in"] else None
        w, o = _both(lambda: ctx.compute_decommitts_sorter_circuit_snapshots(c["q"], c["capacity"], d), lambda: oracle.decommit_sorter_build(c["q"], c["capacity"], d), nv)
        if w is None:
            continue
        for what, key in ((nv.DEC_SORTED_QUERIES, "sorted_q"), (nv.DEC_UNSORTED_TAILS, "unsorted_tails"), (nv.DEC_SORTED_TAILS, "sorted_tails"),
                          (nv.DEC_DEDUP_QUERIES, "dedup_q"), (nv.DEC_DEDUP_TAILS, "dedup_tails"), (nv.DEC_CHALLENGES, "challenges"), (nv.DEC_LHS_Z, "lhs_z"), (nv.DEC_RHS_Z, "rhs_z")):
            assert np.array_equal(w.get(what), o[key]), (seed, key)
        _records_equal(w.get(nv.DEC_INSTANCES), o["instances"], seed)
        assert np.array_equal(np.asarray(o["lhs_z"]).reshape(2, -1)[:, -1], np.asarray(o["rhs_z"]).reshape(2, -1)[:, -1]), seed
        _chained(o["instances"], seed)
        w.free()


def test_events_sorter(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    rin = np.zeros(1, oracle.QUEUE_STATE4)
    rin["tail"], rin["head"], rin["length"] = synthetic.random_field_elements(3, (4,)), synthetic.random_field_elements(4, (4,)), 3
    for seed, c in _each("events_sorter"):
        r = rin if c["result_in"] else None
        w, o = _both(lambda: ctx.compute_events_dedup_and_sort(c["q"], c["capacity"], r), lambda: oracle.events_sorter_build(c["q"], c["capacity"], r), nv)
        if w is None:
            continue
        for what, key in ((nv.EVT_SORTED_QUERIES, "sorted_q"), (nv.EVT_UNSORTED_NEW_TAILS, "unsorted_new_tails"), (nv.EVT_SORTED_NEW_TAILS, "sorted_new_tails"),
                          (nv.EVT_RESULT_QUERIES, "result_q"), (nv.EVT_RESULT_NEW_TAILS, "result_new_tails"), (nv.EVT_CHALLENGES, "challenges"),
                          (nv.EVT_LHS_Z, "lhs_z"), (nv.EVT_RHS_Z, "rhs_z")):
            assert np.array_equal(w.get(what), o[key]), (seed, key)
        _records_equal(w.get(nv.EVT_INSTANCES), o["instances"], seed)
        assert np.array_equal(np.asarray(o["lhs_z"]).reshape(2, -1)[:, -1], np.asarray(o["rhs_z"]).reshape(2, -1)[:, -1]), seed
        _chained(o["instances"], seed)
        w.free()


def test_log_demux(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    for seed, c in _each("log_demux"):
        w, o = _both(lambda: ctx.compute_logs_demux(c["q"], c["capacity"]), lambda: oracle.log_demux_build(c["q"], c["capacity"]), nv)
        if w is None:
            continue
        for what, key in ((nv.DMX_IN_NEW_TAILS, "in_new_tails"), (nv.DMX_OUT_QUERIES, "out_q"), (nv.DMX_OUT_NEW_TAILS, "out_new_tails"), (nv.DMX_OUT_OFFSETS, "out_offsets")):
            assert np.array_equal(w.get(what), o[key]), (seed, key)
        _records_equal(w.get(nv.DMX_INSTANCES), o["instances"], seed)
        _chained(o["instances"], seed)
        w.free()


def test_storage_sorter(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    for seed, c in _each("storage_sorter"):
        w, o = _both(lambda: ctx.compute_storage_dedup_and_sort(c["q"], c["capacity"]), lambda: oracle.storage_sorter_build(c["q"], c["capacity"]), nv)
        if w is None:
            continue
        for what, key in ((nv.STO_SORTED_QUERIES, "sorted_q"), (nv.STO_SORTED_EXT_TS, "sorted_ext_ts"), (nv.STO_UNSORTED_NEW_TAILS, "unsorted_new_tails"),
                          (nv.STO_SORTED_NEW_TAILS, "sorted_new_tails"), (nv.STO_RESULT_QUERIES, "result_q"), (nv.STO_RESULT_NEW_TAILS, "result_new_tails"),
                          (nv.STO_CHALLENGES, "challenges"), (nv.STO_LHS_Z, "lhs_z"), (nv.STO_RHS_Z, "rhs_z")):
            assert np.array_equal(w.get(what), o[key]), (seed, key)
        _records_equal(w.get(nv.STO_INSTANCES), o["instances"], seed)
        assert np.array_equal(np.asarray(o["lhs_z"]).reshape(2, -1)[:, -1], np.asarray(o["rhs_z"]).reshape(2, -1)[:, -1]), seed
        _chained(o["instances"], seed)
        w.free()


def test_decommitter(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv
    from tests.test_oracle_ram import _bytecodes

    mem_in = np.zeros(1, oracle.QUEUE_STATE12)
    mem_in["tail"], mem_in["length"] = synthetic.random_field_elements(9, (12,)), 77
    for seed, c in _each("decommitter"):
        req, tails, words, woff = _bytecodes(oracle, c["n_req"], seed=c["seed"])
        w, o = _both(lambda: ctx.compute_decommitter_circuit_snapshots(req, tails, words, woff, c["capacity"], mem_in),
                     lambda: oracle.decommitter_build(req, tails, words, woff, c["capacity"], mem_in), nv)
        if w is None:
            continue
        for what, key in ((nv.DCM_MEM_QUERIES, "mem_q"), (nv.DCM_MEM_TAILS, "mem_tails"), (nv.DCM_ROUND_STATES, "round_states")):
            assert np.array_equal(w.get(what), o[key]), (seed, key)
        _records_equal(w.get(nv.DCM_INSTANCES), o["instances"], seed)
        w.free()


def test_precompiles(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    for seed, c in _each("precompile"):
        req, mq, kind = c["req"], c["mq"], c["kind"]
        new = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1] if req.size else np.zeros((0, 4), np.uint64)
        mem_in = np.zeros(1, nv.QUEUE_STATE12)
        mem_in["tail"], mem_in["length"] = synthetic.random_field_elements(seed % 97 + 3, (12,)), 12345
        w, o = _both(lambda: ctx._precompile(kind, req, new, mq, c["capacity"], mem_in), lambda: oracle.precompile_build(kind, req, new, mq, c["capacity"], mem_in), nv)
        if w is None:
            continue
        assert w.num_instances == o["instances"].size, seed
        assert np.array_equal(w.get(nv.PRC_MEM_TAILS), o["mem_tails"]), seed
        _records_equal(w.get(nv.PRC_INSTANCES), o["instances"], seed)
        w.free()


def test_linear_hasher(ctx, oracle):
    for seed, c in _each("linear_hasher"):
        assert ctx.compute_linear_keccak256(c["q"]) == oracle.linear_keccak256(c["q"]), seed
