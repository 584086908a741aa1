"""Oracle level: EventsSorter / L1MessagesSorter synthesis ("zkw trace v2", circuit types 11 / 12) is satisfiable, its
boundary rows re-derive the builder's FSM records, and the checker notices tampering."""
import os
import re

import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

P = 0xFFFFFFFF00000001
ROWS_PER_CYCLE = 13


def _bnd(capacity):
    return ROWS_PER_CYCLE * ((capacity + 63) // 64 * 64)


def _slots():
    path = os.path.join(os.path.dirname(__file__), "..", "include", "zkw_events_sorter_circuit_spec.h")
    out = {}
    for m in re.finditer(r"#define ES_(BND_OUT|BND_IN)_(\w+) (\d+)", open(path).read()):
        out.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    return out


@pytest.mark.parametrize("n_forward,rollbacks,capacity,n_rows", [(40, 0.3, 64, 2048), (100, 0.5, 64, 2048), (64, 0.0, 64, 2048),
                                                                 (3, 1.0, 8, 2048), (90, 0.2, 32, 2048)])
def test_oracle_trace_is_satisfied(oracle, n_forward, rollbacks, capacity, n_rows):
    q = synthetic.events_trace(n_forward, rollbacks, seed=n_forward)
    o = oracle.events_sorter_build(q, capacity)
    names = _slots()["BND_OUT"]
    for idx in range(o["instances"].size):
        t = oracle.events_sorter_synthesize(o, idx, capacity, n_rows)
        bad, first = oracle.events_sorter_check(t, capacity)
        assert bad == 0, (idx, first)
        assert int(t.max()) < P and int(t[138].sum()) == 8 * n_rows
        fo = o["instances"][idx]["hidden_fsm_output"]
        bout = t[:, _bnd(capacity) + 1]
        assert [int(bout[names[f"uh{k}"]]) for k in range(4)] == [int(x) for x in fo["initial_unsorted_queue_state"]["head"]]
        assert [int(bout[names[f"sh{k}"]]) for k in range(4)] == [int(x) for x in fo["intermediate_sorted_queue_state"]["head"]]
        assert [int(bout[names["lhs0"]]), int(bout[names["lhs1"]])] == [int(x) for x in fo["lhs_accumulator"]]
        assert [int(bout[names["rhs0"]]), int(bout[names["rhs1"]])] == [int(x) for x in fo["rhs_accumulator"]]
        # the result queue after the flush decision is what the builder hands over
        assert [int(bout[names[f"final_rh{k}"]]) for k in range(4)] == [int(x) for x in fo["final_result_queue_state"]["tail"]]
        assert int(bout[names["final_len_r"]]) == int(fo["final_result_queue_state"]["length"])


def test_checker_notices_tampering(oracle):
    capacity, n_rows = 32, 2048
    q = synthetic.events_trace(25, 0.4, seed=2)
    o = oracle.events_sorter_build(q, capacity)
    t = oracle.events_sorter_synthesize(o, 0, capacity, n_rows)
    assert oracle.events_sorter_check(t, capacity)[0] == 0
    rng = np.random.default_rng(1)
    used = [(c, r) for c in range(138) for r in range(_bnd(capacity) + 56) if t[c, r] != 0]
    for _ in range(40):
        c, r = used[rng.integers(len(used))]
        t2 = t.copy()
        t2[c, r] = (int(t2[c, r]) + 1) % P
        assert oracle.events_sorter_check(t2, capacity)[0] > 0, (c, r)


def test_empty_queue_dummy_instance(oracle):
    """no events: one dummy instance, placeholder FSM input, ONE in the output accumulators (events_sort_dedup.rs:27-76)"""
    o = oracle.events_sorter_build(np.zeros(0, oracle.LOG_QUERY), 16)
    assert o["instances"].size == 1
    t = oracle.events_sorter_synthesize(o, 0, 16, 2048)
    assert oracle.events_sorter_check(t, 16)[0] == 0
    names = _slots()["BND_OUT"]
    bout = t[:, _bnd(16) + 1]
    fo = o["instances"][0]["hidden_fsm_output"]
    assert [int(bout[names[k]]) for k in ("lhs0", "lhs1", "rhs0", "rhs1")] == [1, 1, 1, 1] == [int(x) for x in fo["lhs_accumulator"]] + [int(x) for x in fo["rhs_accumulator"]]


def test_closed_form_section(oracle):
    """challenges, start-flag selection, commitments and the PI row are derived in-trace (gen_ram_circuit.ClosedForm)"""
    from closed_form_case import check_section, events_sorter_tampers

    capacity, n_rows = 32, 2048
    o = oracle.events_sorter_build(synthetic.events_trace(60, 0.3, seed=6), capacity)
    n = o["instances"].size
    assert n >= 3
    check_section(lambda i: oracle.events_sorter_synthesize(o, i, capacity, n_rows), lambda t: oracle.events_sorter_check(t, capacity),
                  oracle.events_sorter_public_inputs(o["instances"])[1], "zkw_events_sorter_circuit_spec.h", "ES", 13, capacity, n,
                  events_sorter_tampers(capacity), challenges=o["challenges"].reshape(2, 21))
