"""Oracle level: LogDemuxer synthesis ("zkw trace v2", circuit type 4) is satisfiable, its boundary rows re-derive the
builder's FSM records (the route is re-derived from the encoding's bytes inside the fill), and the checker notices
tampering."""
import os
import re

import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

P = 0xFFFFFFFF00000001
ROWS_PER_CYCLE = 12
QUEUES = ("st", "ev", "l1", "kc", "sh", "ec")


def _bnd(capacity):
    return ROWS_PER_CYCLE * ((capacity + 63) // 64 * 64)


def _slots():
    path = os.path.join(os.path.dirname(__file__), "..", "include", "zkw_log_demux_circuit_spec.h")
    out = {}
    for m in re.finditer(r"#define LD_(BND_OUT|BND_IN)_(\w+) (\d+)", open(path).read()):
        out.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    return out


def check_boundary(t, inst, capacity):
    names = _slots()["BND_OUT"]
    fo = inst["hidden_fsm_output"]
    bout = t[:, _bnd(capacity) + 1]
    assert [int(bout[names[f"ih{k}"]]) for k in range(4)] == [int(x) for x in fo["initial_log_queue_state"]["head"]]
    assert int(bout[names["len_i"]]) == int(fo["initial_log_queue_state"]["length"])
    for c, q in enumerate(QUEUES):
        assert [int(bout[names[f"qt_{q}{k}"]]) for k in range(4)] == [int(x) for x in fo["queue_state"][c]["tail"]], q
        assert int(bout[names[f"ql_{q}"]]) == int(fo["queue_state"][c]["length"]), q


@pytest.mark.parametrize("n,capacity,n_rows", [(100, 64, 1024), (64, 64, 1024), (5, 8, 1024), (200, 70, 2048)])
def test_oracle_trace_is_satisfied(oracle, n, capacity, n_rows):
    q = synthetic.mixed_log_queue(n, seed=n)
    o = oracle.log_demux_build(q, capacity)
    assert int(o["out_offsets"][6]) < n or n < 50  # some precompile calls are dropped
    for idx in range(o["instances"].size):
        t = oracle.log_demux_synthesize(o, idx, capacity, n_rows)
        bad, first = oracle.log_demux_check(t, capacity)
        assert bad == 0, (idx, first)
        assert int(t.max()) < P and int(t[150].sum()) == 14 * n_rows
        check_boundary(t, o["instances"][idx], capacity)


def test_empty_queue(oracle):
    o = oracle.log_demux_build(np.zeros(0, oracle.LOG_QUERY), 16)
    t = oracle.log_demux_synthesize(o, 0, 16, 1024)
    assert oracle.log_demux_check(t, 16)[0] == 0


def test_checker_notices_tampering(oracle):
    capacity, n_rows = 32, 1024
    q = synthetic.mixed_log_queue(30, seed=2)
    o = oracle.log_demux_build(q, capacity)
    t = oracle.log_demux_synthesize(o, 0, capacity, n_rows)
    assert oracle.log_demux_check(t, capacity)[0] == 0
    rng = np.random.default_rng(1)
    used = [(c, r) for c in range(150) for r in range(_bnd(capacity) + 33) if t[c, r] != 0]
    for _ in range(40):
        c, r = used[rng.integers(len(used))]
        t2 = t.copy()
        t2[c, r] = (int(t2[c, r]) + 1) % P
        assert oracle.log_demux_check(t2, capacity)[0] > 0, (c, r)


def test_misrouted_record_is_rejected(oracle):
    """a witness that pushes a record into another queue than its aux byte / address names cannot satisfy row R"""
    capacity, n_rows = 32, 1024
    q = synthetic.mixed_log_queue(20, seed=5)
    o = oracle.log_demux_build(q, capacity)
    t = oracle.log_demux_synthesize(o, 0, capacity, n_rows)
    path = os.path.join(os.path.dirname(__file__), "..", "include", "zkw_log_demux_circuit_spec.h")
    txt = open(path).read()
    col = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define LD_R_(r_\w+) (\d+)", txt)}
    row_r = int(re.search(r"#define LD_ROW_R (\d+)", txt).group(1)) * ((capacity + 63) // 64 * 64)
    i = next(i for i in range(20) if t[col["r_ev"], row_r + i] == 1)
    t[col["r_ev"], row_r + i] = 0
    t[col["r_l1"], row_r + i] = 1
    assert oracle.log_demux_check(t, capacity)[0] > 0


def test_public_input_row_and_compact_forms(oracle):
    """the PI row carries the commitment of the compact closed-form input; flags and the shared observable input
    behave as CircuitMaker::process prescribes (postprocessing/mod.rs:353-369)"""
    capacity, n_rows = 32, 1024
    q = synthetic.mixed_log_queue(80, seed=7)
    o = oracle.log_demux_build(q, capacity)
    compact, pi = oracle.log_demux_public_inputs(o["instances"])
    n_inst = o["instances"].size
    assert n_inst == 3 and compact[0, 0] == 1 and compact[-1, 1] == 1 and compact[1, 0] == 0 and compact[0, 1] == 0
    assert all((compact[i, 2:6] == compact[0, 2:6]).all() for i in range(n_inst))  # one observable input per block
    assert all((compact[i, 10:14] == compact[i - 1, 14:18]).all() for i in range(1, n_inst))  # fsm_in = previous fsm_out
    assert len({tuple(p) for p in pi.tolist()}) == n_inst
    t = oracle.log_demux_synthesize(o, 1, capacity, n_rows)
    assert t[:4, _bnd(capacity) + 2].tolist() == pi[1].tolist()


def test_closed_form_section(oracle):
    """start-flag selection, commitments and the PI row are derived in-trace (gen_ram_circuit.ClosedForm); no challenges in this circuit"""
    from closed_form_case import check_section, log_demux_tampers

    capacity, n_rows = 64, 2048
    o = oracle.log_demux_build(synthetic.mixed_log_queue(170, seed=6), capacity)
    n = o["instances"].size
    assert n == 3
    check_section(lambda i: oracle.log_demux_synthesize(o, i, capacity, n_rows), lambda t: oracle.log_demux_check(t, capacity),
                  oracle.log_demux_public_inputs(o["instances"])[1], "zkw_log_demux_circuit_spec.h", "LD", 12, capacity, n,
                  log_demux_tampers(capacity))
