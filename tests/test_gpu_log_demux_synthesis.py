"""GPU parity: LogDemuxer synthesis through the C ABI vs the oracle's trace, cell by cell, and the GPU satisfiability
checker on clean and tampered traces."""
import ctypes as C

import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
LD_COLS = 151


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,capacity,n_rows", [(5, 8, 1024), (100, 64, 1024), (64, 64, 1024), (200, 70, 2048), (3000, 1000, 16384), (0, 16, 1024)])
def test_trace_matches_oracle(ctx, oracle, n, capacity, n_rows):
    from era_zkevm_test_harness_amd import native

    q = synthetic.mixed_log_queue(n, seed=n) if n else np.zeros(0, oracle.LOG_QUERY)
    o = oracle.log_demux_build(q, capacity)
    w = ctx.compute_logs_demux(q, capacity)
    n_inst = o["instances"].size
    t = native.Trace(ctx, n_rows, n_inst, n_cols=LD_COLS)
    assert t.n_cols == LD_COLS
    ctx.synthesize_log_demux(w, t)
    for idx in range(n_inst):
        got = t.get(idx)
        exp = oracle.log_demux_synthesize(o, idx, capacity, n_rows)
        if not np.array_equal(got, exp):
            bad = np.argwhere(got != exp)
            raise AssertionError(f"instance {idx}: {len(bad)} cells differ, first (col, row) = {bad[:8].tolist()}")
        assert ctx.check_if_satisfied_log_demux(t, idx, capacity)[0] == 0
    t.free()


def test_needs_a_wide_trace(ctx):
    from era_zkevm_test_harness_amd import native

    w = ctx.compute_logs_demux(synthetic.mixed_log_queue(20, seed=1), 32)
    t = native.Trace(ctx, 1024, 1)  # 149 columns
    with pytest.raises(native.ZkwError):
        ctx.synthesize_log_demux(w, t)
    t.free()


def test_production_geometry(ctx, oracle):
    """capacity 58 750 in a 2^20-row trace: one full instance and a ragged last one."""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 58750, 1 << 20
    q = synthetic.mixed_log_queue(70000, seed=9)
    w = ctx.compute_logs_demux(q, capacity)
    t = native.Trace(ctx, n_rows, 2, n_cols=LD_COLS)
    ctx.synthesize_log_demux(w, t)
    for idx in range(2):
        bad, first = ctx.check_if_satisfied_log_demux(t, idx, capacity)
        assert bad == 0, (idx, first)
        mult = t.get(idx, 150, 1)[0]
        assert int(mult.sum()) == 14 * n_rows and not mult[256:].any()
    t.free()


def test_gpu_checker_flags_tampering(ctx, oracle):
    import torch

    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 64, 1024
    q = synthetic.mixed_log_queue(50, seed=3)
    w = ctx.compute_logs_demux(q, capacity)
    t = native.Trace(ctx, n_rows, 1, n_cols=LD_COLS)
    ctx.synthesize_log_demux(w, t, 0, 1)
    assert ctx.check_if_satisfied_log_demux(t, 0, capacity)[0] == 0
    host = t.get(0)
    rng = np.random.default_rng(2)
    used = np.argwhere(host[:150, :12 * 64 + 33] != 0)
    base = native.load().zkw_trace_device_ptr(t.handle, 0)
    hip = C.CDLL("libamdhip64.so")
    for _ in range(25):
        c, r = used[rng.integers(len(used))]
        addr = base + (int(c) * n_rows + int(r)) * 8
        old = np.array([host[c, r]], np.uint64)
        new = np.array([(int(host[c, r]) + 1) % P], np.uint64)
        torch.cuda.synchronize()
        hip.hipMemcpy(C.c_void_p(addr), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
        assert ctx.check_if_satisfied_log_demux(t, 0, capacity)[0] > 0, (c, r)
        hip.hipMemcpy(C.c_void_p(addr), old.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
    assert ctx.check_if_satisfied_log_demux(t, 0, capacity)[0] == 0
    # the closed-form section: challenges, start-flag selection, commitments, public input — same verdict as the oracle's checker
    from closed_form_case import gpu_tamper_parity, log_demux_tampers
    gpu_tamper_parity(base, n_rows, host, lambda: ctx.check_if_satisfied_log_demux(t, 0, capacity), lambda h: oracle.log_demux_check(h, capacity),
                      log_demux_tampers(capacity))
    t.free()


def test_compact_forms_and_public_inputs(ctx, oracle):
    """a20: compact closed-form inputs and public-input commitments of every instance, GPU vs oracle"""
    from era_zkevm_test_harness_amd import native as nv

    q = synthetic.mixed_log_queue(100, seed=11)
    o = oracle.log_demux_build(q, 32)
    w = ctx.compute_logs_demux(q, 32)
    compact, pi = oracle.log_demux_public_inputs(o["instances"])
    assert o["instances"].size >= 3
    assert np.array_equal(w.get(nv.DMX_COMPACT_FORMS), compact)
    assert np.array_equal(w.get(nv.DMX_PUBLIC_INPUTS), pi)


def test_slot_reuse_keeps_the_zero_cells(ctx, oracle):
    """as tests/test_gpu_decommit_sorter_synthesis.py::test_slot_reuse_keeps_the_zero_cells, for the log demultiplexer"""
    from era_zkevm_test_harness_amd import native

    n_rows = 2048
    q = synthetic.mixed_log_queue(200, seed=8)
    t = native.Trace(ctx, n_rows, 1, n_cols=LD_COLS)
    for capacity in (70, 32, 70):
        w = ctx.compute_logs_demux(q, capacity)
        o = oracle.log_demux_build(q, capacity)
        for idx in range(o["instances"].size):
            ctx.synthesize_log_demux(w, t, idx, 1, 0)
            assert np.array_equal(t.get(0), oracle.log_demux_synthesize(o, idx, capacity, n_rows)), (capacity, idx)
        w.free()
    t.free()
