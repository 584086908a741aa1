"""Oracle level: CodeDecommittmentsSorter synthesis ("zkw trace v2", circuit type 2) is satisfiable, its boundary rows
re-derive the builder's FSM records, and the checker notices tampering."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

P = 0xFFFFFFFF00000001
ROWS_PER_CYCLE, N_BND = 7, 4


def _bnd(capacity):
    return ROWS_PER_CYCLE * ((capacity + 63) // 64 * 64)


@pytest.mark.parametrize("n,n_hashes,capacity,n_rows", [(100, 7, 128, 1024), (256, 40, 128, 1024), (300, 3, 128, 1024), (5, 5, 8, 512),
                                                        (64, 1, 64, 512)])
def test_oracle_trace_is_satisfied(oracle, n, n_hashes, capacity, n_rows):
    q = synthetic.decommit_trace(n, n_hashes, seed=n)
    o = oracle.decommit_sorter_build(q, capacity)
    for idx in range(o["instances"].size):
        t = oracle.decommit_sorter_synthesize(o, idx, capacity, n_rows)
        bad, first = oracle.decommit_sorter_check(t, capacity)
        assert bad == 0, (idx, first)
        assert int(t.max()) < P and int(t[148].sum()) == 18 * n_rows
        # BND_OUT (registers carried by the fill itself) == the builder's hidden_fsm_output
        fo = o["instances"][idx]["hidden_fsm_output"]
        bout = t[:, _bnd(capacity) + 1]
        assert np.array_equal(bout[0:12], fo["initial_queue_state"]["head"])
        assert np.array_equal(bout[12:24], fo["sorted_queue_state"]["head"])
        assert [int(bout[36]), int(bout[37])] == [int(fo["initial_queue_state"]["length"]), int(fo["sorted_queue_state"]["length"])]
        assert np.array_equal(bout[39:41], fo["lhs_accumulator"]) and np.array_equal(bout[41:43], fo["rhs_accumulator"])
        # the deduplicated queue after the flush decision: final_rh / final_len_r
        last = idx == o["instances"].size - 1
        want = o["instances"][idx]["final_queue_state"] if last else fo["final_queue_state"]
        names = _slot_names()
        frh = [int(bout[names["BND_OUT"][f"final_rh{k}"]]) for k in range(12)]
        assert frh == [int(x) for x in want["tail"]]
        assert int(bout[names["BND_OUT"]["final_len_r"]]) == int(want["length"])


def _slot_names():
    import os
    import re

    path = os.path.join(os.path.dirname(__file__), "..", "include", "zkw_decommit_sorter_circuit_spec.h")
    out = {}
    for m in re.finditer(r"#define DS_(BND_OUT|BND_IN)_(\w+) (\d+)", open(path).read()):
        out.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    return out


def test_checker_notices_tampering(oracle):
    capacity, n_rows = 64, 512
    q = synthetic.decommit_trace(100, 9, seed=2)
    o = oracle.decommit_sorter_build(q, capacity)
    t = oracle.decommit_sorter_synthesize(o, 0, capacity, n_rows)
    assert oracle.decommit_sorter_check(t, capacity)[0] == 0
    rng = np.random.default_rng(1)
    used = [(c, r) for c in range(148) for r in range(_bnd(capacity) + 54) if t[c, r] != 0]
    for _ in range(40):
        c, r = used[rng.integers(len(used))]
        t2 = t.copy()
        t2[c, r] = (int(t2[c, r]) + 1) % P
        assert oracle.decommit_sorter_check(t2, capacity)[0] > 0, (c, r)


def test_closed_form_section(oracle):
    """the PI row is derived in-trace (sponges over the closed-form input -> compact form -> commitment) and equals the builder's public
    input; the challenges of BND_IN are the challenge sponge's outputs; tampering with any of it is caught"""
    from closed_form_case import decommit_sorter_tampers
    from era_zkevm_test_harness_amd.ram_circuit import spec_macros

    M = spec_macros("zkw_decommit_sorter_circuit_spec.h", "DS")
    capacity, n_rows = 64, 1024
    o = oracle.decommit_sorter_build(synthetic.decommit_trace(150, 9, seed=4), capacity)
    _, pis = oracle.decommit_sorter_public_inputs(o["instances"])
    b = _bnd(capacity)
    assert o["instances"].size == 3
    for idx in range(3):
        t = oracle.decommit_sorter_synthesize(o, idx, capacity, n_rows)
        assert oracle.decommit_sorter_check(t, capacity)[0] == 0
        assert np.array_equal(t[M["PI_pi0"]:M["PI_pi0"] + 4, b + M["ROWOFF_PI"]], pis[idx])
        assert int(t[M["SEL0_flag"], b + M["ROWOFF_SEL0"]]) == (1 if idx == 0 else 0)
        ch = o["challenges"].reshape(2, 9)
        for rep in range(2):
            assert np.array_equal(t[M["CH3_CH3_o0"]:M["CH3_CH3_o0"] + 8, b + M[f"ROWOFF_CH{3 + rep}"]], ch[rep, 1:])
        for name, c, r in decommit_sorter_tampers(capacity):
            t2 = t.copy()
            t2[c, r] = (int(t2[c, r]) + 1) % P
            assert oracle.decommit_sorter_check(t2, capacity)[0] > 0, (idx, name)


def test_unsorted_or_mislabelled_input_is_rejected(oracle):
    q = synthetic.decommit_trace(50, 5, seed=3)
    o = oracle.decommit_sorter_build(q, 64)
    bad = dict(o)
    bad["sorted_q"] = o["sorted_q"].copy()
    bad["sorted_q"]["is_fresh"][1] ^= 1  # freshness no longer marks the first request of a hash
    with pytest.raises(RuntimeError):
        oracle.decommit_sorter_synthesize(bad, 0, 64, 512)


def test_public_inputs_oracle(oracle):
    q = synthetic.decommit_trace(300, 20, seed=8)
    o = oracle.decommit_sorter_build(q, 128)
    compact, pi = oracle.decommit_sorter_public_inputs(o["instances"])
    assert compact.shape == (3, 18) and list(compact[:, 0]) == [1, 0, 0] and list(compact[:, 1]) == [0, 0, 1]
    assert (compact[:, 2:6] == compact[0, 2:6]).all()  # shared observable input
    assert np.array_equal(compact[:-1, 14:18], compact[1:, 10:14])  # FSM chaining
    assert np.array_equal(compact[0, 6:10], compact[1, 6:10]) and not np.array_equal(compact[1, 6:10], compact[2, 6:10])
    for i in range(3):
        assert np.array_equal(pi[i], oracle.commit_var_length(compact[i]))
    t = oracle.decommit_sorter_synthesize(o, 1, 128, 1024)
    assert np.array_equal(t[0:4, 7 * 128 + 3], pi[1])
