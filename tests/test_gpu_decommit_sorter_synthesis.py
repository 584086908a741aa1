"""GPU parity: CodeDecommittmentsSorter synthesis through the C ABI vs the oracle's trace, cell by cell, and the GPU
satisfiability checker vs the oracle's on clean and tampered traces."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,n_hashes,capacity,n_rows", [(5, 5, 8, 512), (100, 7, 128, 1024), (256, 40, 128, 1024), (300, 3, 128, 1024),
                                                        (64, 1, 64, 512), (3000, 400, 1024, 8192)])
def test_trace_matches_oracle(ctx, oracle, n, n_hashes, capacity, n_rows):
    from era_zkevm_test_harness_amd import native

    q = synthetic.decommit_trace(n, n_hashes, seed=n)
    o = oracle.decommit_sorter_build(q, capacity)
    w = ctx.compute_decommitts_sorter_circuit_snapshots(q, capacity)
    n_inst = o["instances"].size
    compact, pis = oracle.decommit_sorter_public_inputs(o["instances"])
    assert np.array_equal(w.get(native.DEC_COMPACT_FORMS), compact) and np.array_equal(w.get(native.DEC_PUBLIC_INPUTS), pis)
    t = native.Trace(ctx, n_rows, n_inst)
    ctx.synthesize_decommit_sorter(w, t)
    for idx in range(n_inst):
        got = t.get(idx)
        exp = oracle.decommit_sorter_synthesize(o, idx, capacity, n_rows)
        if not np.array_equal(got, exp):
            bad = np.argwhere(got != exp)
            raise AssertionError(f"instance {idx}: {len(bad)} cells differ, first (col, row) = {bad[:8].tolist()}")
        assert ctx.check_if_satisfied_decommit_sorter(t, idx, capacity)[0] == 0
    t.free()


def test_production_geometry(ctx, oracle):
    """capacity 117 500 in a 2^20-row trace: one full instance and a ragged last one."""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 117500, 1 << 20
    q = synthetic.decommit_trace(capacity + 5000, 3000, seed=9)
    w = ctx.compute_decommitts_sorter_circuit_snapshots(q, capacity)
    t = native.Trace(ctx, n_rows, 2)
    ctx.synthesize_decommit_sorter(w, t)
    for idx in range(2):
        bad, first = ctx.check_if_satisfied_decommit_sorter(t, idx, capacity)
        assert bad == 0, (idx, first)
        mult = t.get(idx, 148, 1)[0]
        assert int(mult.sum()) == 18 * n_rows and not mult[256:].any()
    t.free()


def test_gpu_checker_flags_tampering(ctx, oracle):
    import torch

    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 128, 1024
    q = synthetic.decommit_trace(200, 11, seed=3)
    w = ctx.compute_decommitts_sorter_circuit_snapshots(q, capacity)
    t = native.Trace(ctx, n_rows, 1)
    ctx.synthesize_decommit_sorter(w, t, 0, 1)
    assert ctx.check_if_satisfied_decommit_sorter(t, 0, capacity)[0] == 0
    host = t.get(0)
    rng = np.random.default_rng(2)
    used = np.argwhere(host[:148, :7 * 128 + 54] != 0)
    base = native.load().zkw_trace_device_ptr(t.handle, 0)
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    for _ in range(25):
        c, r = used[rng.integers(len(used))]
        addr = base + (int(c) * n_rows + int(r)) * 8
        old = np.array([host[c, r]], np.uint64)
        new = np.array([(int(host[c, r]) + 1) % P], np.uint64)
        torch.cuda.synchronize()
        hip.hipMemcpy(C.c_void_p(addr), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
        assert ctx.check_if_satisfied_decommit_sorter(t, 0, capacity)[0] > 0, (c, r)
        hip.hipMemcpy(C.c_void_p(addr), old.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
    assert ctx.check_if_satisfied_decommit_sorter(t, 0, capacity)[0] == 0
    # the closed-form section: challenges, start-flag selection, commitments, public input — same verdict as the oracle's checker
    from closed_form_case import decommit_sorter_tampers
    for name, c, r in decommit_sorter_tampers(capacity):
        addr = base + (int(c) * n_rows + int(r)) * 8
        new = np.array([(int(host[c, r]) + 1) % P], np.uint64)
        hip.hipMemcpy(C.c_void_p(addr), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
        bad, first = ctx.check_if_satisfied_decommit_sorter(t, 0, capacity)
        h2 = host.copy()
        h2[c, r] = new[0]
        obad, ofirst = oracle.decommit_sorter_check(h2, capacity)
        assert bad > 0 and bad == obad and first[0] == ofirst[0], (name, bad, first, obad, ofirst)
        hip.hipMemcpy(C.c_void_p(addr), np.array([host[c, r]], np.uint64).ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
    assert ctx.check_if_satisfied_decommit_sorter(t, 0, capacity)[0] == 0
    t.free()


def test_slot_reuse_keeps_the_zero_cells(ctx, oracle):
    """A slot that already holds this layout keeps every cell that is zero in all of its traces (padding rows, region gaps, the columns a row
    type does not use, multiplicity rows >= 256): consecutive instances — full ones, then a ragged last one — and a change of capacity
    (another layout: everything is rewritten) into ONE slot, each compared cell for cell."""
    from era_zkevm_test_harness_amd import native

    n_rows = 2048
    q = synthetic.decommit_trace(300, 40, seed=6)
    t = native.Trace(ctx, n_rows, 1)
    for capacity in (128, 64, 128):
        w = ctx.compute_decommitts_sorter_circuit_snapshots(q, capacity)
        o = oracle.decommit_sorter_build(q, capacity)
        for idx in range(o["instances"].size):
            ctx.synthesize_decommit_sorter(w, t, idx, 1, 0)
            assert np.array_equal(t.get(0), oracle.decommit_sorter_synthesize(o, idx, capacity, n_rows)), (capacity, idx)
        w.free()
    t.free()
