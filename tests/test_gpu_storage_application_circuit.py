"""GPU: the StorageApplication circuit (type 10) in "zkw trace v4" through the C ABI (zkw_storage_application_synthesize: k_sap_walk_prepare +
the netlist engine, csrc/netlist_kernels.cuh) against the oracle (oracle/netlist_circuit.c): traces cell for cell incl. ragged and dummy
instances, the two checkers violation for violation on tampered cells of every region, production geometry (33 tree queries, 2^20 rows)
through the GPU checker, a slot dirtied before the synthesis."""
import ctypes

import numpy as np
import pytest

from sap_case import storage_application_case

pytestmark = pytest.mark.gpu
N_ROWS = 1 << 18
WALK = 257


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def _build(ctx, oracle, n, capacity, seed):
    q, tails, tree, idx, paths = storage_application_case(oracle, n, seed=seed)
    w = ctx.decompose_into_storage_application_witnesses(q, tails, idx, paths, tree.root, tree.next_enumeration_index, capacity)
    o = oracle.storage_application_build(tree, q, tails, capacity)  # mutates the tree
    return q, w, o


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return hip


@pytest.mark.parametrize("n,capacity", [(9, 5), (0, 4), (1, 3)])
def test_traces_match_oracle(ctx, oracle, n, capacity):
    from era_zkevm_test_harness_amd import native

    q, w, o = _build(ctx, oracle, n, capacity, seed=n + 3)
    ni = w.num_instances
    assert ni == o["instances"].size
    lay = native.circuit_layout(10, capacity)
    g = oracle.nl_geometry(10)
    assert int(lay["num_columns"]) == native.SA_COLS == g["cols"] and int(lay["total_table_rows"]) == g["table_rows"] == 132352
    assert int(lay["rows_per_cycle"]) == g["rows_per_cycle"] and int(lay["rows_used"]) == capacity * WALK * g["rows_per_cycle"] + 2 * 2 + 1 + oracle.nlcf_geometry(10, capacity * WALK)["rows"]
    t = native.Trace(ctx, N_ROWS, ni, n_cols=native.SA_COLS)
    garbage = np.full((native.SA_COLS, N_ROWS), 0x1234567, np.uint64)  # the slot's previous tenant
    ctx.synchronize()
    assert _hip().hipMemcpy(t.device_ptr(0), garbage.ctypes.data, garbage.nbytes, 1) == 0
    ctx.synthesize_storage_application(w, t, 0, ni, 0)
    pis = oracle.closed_form_public_inputs(10, o["instances"])[1]
    for i in range(ni):
        exp = oracle.storage_application_synthesize(o, q, i, capacity, N_ROWS)
        got = t.get(i)
        assert np.array_equal(got, exp), np.argwhere(got != exp)[:4]
        assert ctx.check_if_satisfied_storage_application(t, i, capacity) == (0, (0, 0, 0))
        assert got[:4, int(lay["public_input_row"][0])].tolist() == pis[i].tolist()
    # a second synthesis into the same slots (the slot keeps its layout tag: nothing is cleared) gives the same cells
    ctx.synthesize_storage_application(w, t, 0, ni, 0)
    assert np.array_equal(t.get(ni - 1), oracle.storage_application_synthesize(o, q, ni - 1, capacity, N_ROWS))
    t.free()
    w.free()


def test_tamper_parity(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    capacity = 4
    q, w, o = _build(ctx, oracle, 6, capacity, seed=21)
    t = native.Trace(ctx, N_ROWS, 1, n_cols=native.SA_COLS)
    base = oracle.storage_application_synthesize(o, q, 0, capacity, N_ROWS)
    g = oracle.nl_geometry(10)
    rpc, G, cols = g["rows_per_cycle"], g["general"], g["cols"]
    used = capacity * WALK * rpc
    rng = np.random.default_rng(10)
    cells = [(int(rng.integers(0, G)), int(rng.integers(0, used))) for _ in range(14)]                     # general-purpose cells: gates, headers, empties
    cells += [(int(rng.integers(G, cols - 1)), int(rng.integers(0, used))) for _ in range(14)]              # lookup cells
    cells += [(cols - 1, int(rng.integers(0, g["table_rows"]))) for _ in range(4)] + [(cols - 1, g["table_rows"] + 5)]  # multiplicities
    cells += [(int(rng.integers(0, G)), used + k) for k in range(0, 2 * 2 + 3)]                             # boundary rows, PI, below
    cells += [(0, rpc), (1, rpc), (2, rpc), (3, 2 * rpc), (0, WALK * rpc), (2, WALK * rpc)]                 # reset / idle / masks of level and leaf cycles
    from nlcf_cells import closed_form_cells
    cells += closed_form_cells(oracle, 10, capacity * WALK, rng)  # the closed-form section: words, commitment sponges, the PI row
    hip = _hip()
    n_flagged = 0
    for col, row in cells:
        bad = base.copy()
        bad[col, row] = bad[col, row] + 1 if rng.random() < 0.7 else 300
        ctx.synchronize()
        assert hip.hipMemcpy(t.device_ptr(0), bad.ctypes.data, bad.nbytes, 1) == 0
        got = ctx.check_if_satisfied_storage_application(t, 0, capacity)
        want = oracle.storage_application_check(bad, capacity)
        assert got == want, ((col, row), got, want)
        n_flagged += got[0] > 0
    assert n_flagged >= len(cells) - 8  # (PI cells and the FREE inputs of lookups are under no constraint of their own)
    t.free()
    w.free()


def test_production_geometry(ctx, oracle):
    """2^20 rows at the reference's capacity (33 tree queries = 8 481 cycles): the instances of a block of 60 queries synthesized in one
    call and checked on the GPU; the multiplicity column counts every lookup slot once; the walks arrive at the tree's roots"""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 33, 1 << 20
    q, w, o = _build(ctx, oracle, 60, capacity, seed=77)
    ni = w.num_instances
    assert ni >= 2
    lay = native.circuit_layout(10)
    assert int(lay["capacity"]) == capacity and bool(lay["fits"]) and int(lay["rows_used"]) <= n_rows
    t = native.Trace(ctx, n_rows, ni, n_cols=native.SA_COLS)
    ctx.synthesize_storage_application(w, t, 0, ni, 0)
    g = oracle.nl_geometry(10)
    slots = oracle.nl_slots_per_cycle(10)
    for i in range(ni):
        assert ctx.check_if_satisfied_storage_application(t, i, capacity) == (0, (0, 0, 0))
    got = t.get(0)
    assert int(got[g["cols"] - 1].sum()) == capacity * WALK * slots
    # BND_OUT of a full instance = the root after its last query (the last walk is a read's or a write's second)
    inst = o["instances"][0]
    last = int(inst["first_item"]) + int(inst["num_items"]) - 1
    n_walks = sum(2 if q["rw_flag"][int(inst["first_item"]) + k] else 1 for k in range(int(inst["num_items"])))
    bnd_out = capacity * WALK * g["rows_per_cycle"] + 2
    assert bytes(int(x) for x in got[:32, bnd_out]) == bytes(o["roots"][last]), n_walks
    exp = oracle.storage_application_synthesize(o, q, 0, capacity, n_rows)
    assert np.array_equal(got, exp)
    t.free()
    w.free()


def test_lane_per_cycle_fill_is_identical(ctx, oracle):
    """zkw_set_netlist_fill_form(ctx, 1): the lane-per-cycle form of the netlist fill (k_nl_walk + k_nl_expand) gives the same cells
    as the oracle for a StorageApplication and a Sha256RoundFunction instance, garbage in the slot before"""
    from era_zkevm_test_harness_amd import native, synthetic

    ctx.set_netlist_fill_form(1)
    try:
        q, w, o = _build(ctx, oracle, 7, 4, seed=31)
        t = native.Trace(ctx, N_ROWS, 1, n_cols=native.SA_COLS)
        garbage = np.full((native.SA_COLS, N_ROWS), 77, np.uint64)
        ctx.synchronize()
        assert _hip().hipMemcpy(t.device_ptr(0), garbage.ctypes.data, garbage.nbytes, 1) == 0
        ctx.synthesize_storage_application(w, t, 1, 1, 0)
        assert np.array_equal(t.get(0), oracle.storage_application_synthesize(o, q, 1, 4, N_ROWS))
        assert ctx.check_if_satisfied_storage_application(t, 0, 4) == (0, (0, 0, 0))
        t.free()
        w.free()
        req, mq = synthetic.precompile_trace(1, 9, seed=3, max_rounds=4)
        tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
        mem_in = np.zeros(1, native.QUEUE_STATE12)
        ws, os_ = ctx._precompile(1, req, tails, mq, 70, mem_in), oracle.precompile_build(1, req, tails, mq, 70, mem_in)
        t = native.Trace(ctx, N_ROWS, 1, n_cols=native.SC_COLS)
        ctx.synthesize_sha256_round_function(ws, t, 0, 1, 0)
        assert np.array_equal(t.get(0), oracle.sha256_round_synthesize(os_, 0, 70, N_ROWS))
        t.free()
        ws.free()
    finally:
        ctx.set_netlist_fill_form(0)
