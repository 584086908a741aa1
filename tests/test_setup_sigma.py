"""CPU (host arithmetic of libzkw): zkw_setup_copy_permutation of the queue circuits — sigma is a permutation, every
oracle-synthesized trace satisfies trace[cell] == trace[sigma[cell]], and sigma's classes are the link classes of the
spec-driven checker: bumping a cell that sigma moves is a copy violation (kind 4) for the oracle's checker too, bumping a
cell sigma fixes is never one."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import native as nv, synthetic

N_ROWS = 1 << 15


def _cases(oracle):
    w = oracle.ram_build_instances(synthetic.ram_trace(2983, seed=5), 1000, 0)
    yield 8, 1000, [oracle.ram_synthesize(w, i, 1000, N_ROWS) for i in (0, 2)], oracle.ram_check
    d = oracle.decommit_sorter_build(synthetic.decommit_trace(700, 60, seed=3), 300)
    yield 2, 300, [oracle.decommit_sorter_synthesize(d, i, 300, N_ROWS) for i in (0, d["instances"].size - 1)], oracle.decommit_sorter_check
    lq = synthetic.mixed_log_queue(900, seed=4)
    m = oracle.log_demux_build(lq, 400)
    yield 4, 400, [oracle.log_demux_synthesize(m, i, 400, N_ROWS) for i in (0, m["instances"].size - 1)], oracle.log_demux_check
    e = oracle.events_sorter_build(synthetic.events_trace(500, 0.3, seed=6), 300)
    yield 11, 300, [oracle.events_sorter_synthesize(e, i, 300, N_ROWS) for i in (0, e["instances"].size - 1)], oracle.events_sorter_check


def test_sigma_is_the_link_structure(oracle):
    rng = np.random.default_rng(1)
    for ctype, cap, traces, check in _cases(oracle):
        sigma = nv.setup_copy_permutation(ctype, cap, N_ROWS)
        flat = sigma.reshape(-1)
        ident = np.arange(flat.size, dtype=np.uint64)
        assert np.array_equal(np.sort(flat), ident), ctype                       # a permutation
        moved = np.flatnonzero(flat != ident)
        assert moved.size > 100 * cap // 10
        for t in traces:
            body = t[:sigma.shape[0]].reshape(-1)
            assert np.array_equal(body, body[flat]), ctype                       # satisfied traces satisfy sigma
        t = traces[0]
        G = sigma.shape[0]
        for cell in rng.choice(moved, 6, replace=False):                         # a cell in a copy cycle
            bad = t.copy()
            bad[int(cell) // N_ROWS, int(cell) % N_ROWS] += 1
            body = bad[:G].reshape(-1)
            assert not np.array_equal(body, body[flat])
            n, first = check(bad, cap)
            assert n > 0
        used_fixed = np.flatnonzero((flat == ident) & (t[:G].reshape(-1) != 0))
        for cell in rng.choice(used_fixed, 4, replace=False):                    # a used cell outside every copy cycle
            bad = t.copy()
            bad[int(cell) // N_ROWS, int(cell) % N_ROWS] += 1
            body = bad[:G].reshape(-1)
            assert np.array_equal(body, body[flat])                              # sigma does not see it ...
            n, first = check(bad, cap)
            assert n == 0 or first[0] != 4, (ctype, int(cell), first)            # ... and neither do the checker's links


def test_sigma_of_the_netlist_circuits(oracle):
    """types 5, 13, 6, 3, 10: the classes come from the netlists' operand references — traces satisfy sigma, a bumped cell of a
    cycle is a copy violation (kind 2) or a broken relation (kind 1 / 7) for the oracle's checker, a free witness byte is in
    no cycle"""
    from tests.test_setup_selectors import _netlist_cases

    rng = np.random.default_rng(2)
    checks = {5: oracle.keccak_round_check, 13: lambda t, c: oracle.linear_hasher_check(t, oracle.linear_hasher_cycles(c)),
              6: oracle.sha256_round_check, 3: oracle.code_decommitter_check, 10: oracle.storage_application_check}
    n_rows = 1 << 18
    for ctype, cap, trace, col0, width, lpr in _netlist_cases(oracle):
        sigma = nv.setup_copy_permutation(ctype, cap, n_rows)
        G = sigma.shape[0]
        assert G == col0 + width * lpr
        flat = sigma.reshape(-1)
        ident = np.arange(flat.size, dtype=np.uint64)
        assert np.array_equal(np.sort(flat), ident), ctype
        body = trace[:G].reshape(-1)
        assert np.array_equal(body, body[flat]), ctype
        moved = np.flatnonzero(flat != ident)
        assert moved.size > 10000
        for cell in rng.choice(moved, 5, replace=False):
            bad = trace.copy()
            bad[int(cell) // n_rows, int(cell) % n_rows] += 1
            b = bad[:G].reshape(-1)
            assert not np.array_equal(b, b[flat])
            n, first = checks[ctype](bad, cap)
            assert n > 0 and first[0] in (1, 2, 7), (ctype, int(cell), first)


def test_sigma_of_the_ecrecover_circuit(oracle):
    """type 7: Keccak-f netlist + queue section + the EC section (cells with a tape reference are copies of the value's home cell, an input
    byte's home of the read query's value byte, the netlist's key / mask / ok elements of EC home cells): the oracle's trace satisfies
    sigma, a bumped cell of a cycle is caught by the oracle's checker, cells of all three parts are in cycles"""
    cap, n_rows = 2, 1 << 18
    req, mq = synthetic.precompile_trace(2, 3, seed=4)
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
    b = oracle.precompile_build(2, req, tails, mq, cap, np.zeros(1, nv.QUEUE_STATE12))
    t = oracle.ecrecover_synthesize(b, 0, cap, n_rows)
    assert oracle.ecrecover_check(t, cap)[0] == 0
    sigma = nv.setup_copy_permutation(7, cap, n_rows)
    G = sigma.shape[0]
    assert G == 128
    flat = sigma.reshape(-1)
    ident = np.arange(flat.size, dtype=np.uint64)
    assert np.array_equal(np.sort(flat), ident)
    body = t[:G].reshape(-1)
    assert np.array_equal(body, body[flat])
    moved = np.flatnonzero(flat != ident)
    g = oracle.ec_geometry(cap)
    rows = moved % n_rows
    lay = nv.circuit_layout(7, cap)
    in_ec = (rows >= g["first_row"]) & (rows < g["first_row"] + cap * g["rows_per_cycle"])
    in_queue = (rows >= int(lay["queue_first_row"])) & (rows < g["first_row"])
    assert in_ec.sum() > 100000 and in_queue.sum() > 500 and (~in_ec & ~in_queue).sum() > 10000
    rng = np.random.default_rng(3)
    for cell in np.concatenate([rng.choice(moved[in_ec], 6, replace=False), rng.choice(moved[in_queue], 2, replace=False)]):
        bad = t.copy()
        bad[int(cell) // n_rows, int(cell) % n_rows] += 1
        bb = bad[:G].reshape(-1)
        assert not np.array_equal(bb, bb[flat])
        assert oracle.ecrecover_check(bad, cap)[0] > 0, int(cell)
    # an EC-section cell that holds a value and is in no cycle: a constant or a value used once (its own home); sigma fixes it
    ec_cells = np.flatnonzero((flat == ident) & (body != 0))
    ec_cells = ec_cells[(ec_cells % n_rows >= g["first_row"]) & (ec_cells % n_rows < g["first_row"] + cap * g["rows_per_cycle"])]
    assert ec_cells.size > 0


def test_sigma_rejects_what_it_cannot_describe():
    with pytest.raises(nv.ZkwError):
        nv.setup_copy_permutation(1, 0, 1 << 20)   # MainVM: no layout
    with pytest.raises(nv.ZkwError):
        nv.setup_copy_permutation(8, 136714, 1 << 19)
