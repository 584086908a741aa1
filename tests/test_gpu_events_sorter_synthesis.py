"""GPU parity: EventsSorter / L1MessagesSorter synthesis through the C ABI vs the oracle's trace, cell by cell, and the
GPU satisfiability checker on clean and tampered traces."""
import ctypes as C

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


@pytest.mark.parametrize("n_forward,rollbacks,capacity,n_rows", [(3, 1.0, 8, 2048), (40, 0.3, 64, 2048), (100, 0.5, 64, 2048),
                                                                 (64, 0.0, 64, 2048), (90, 0.2, 32, 2048), (2000, 0.3, 700, 16384)])
def test_trace_matches_oracle(ctx, oracle, n_forward, rollbacks, capacity, n_rows):
    from era_zkevm_test_harness_amd import native

    q = synthetic.events_trace(n_forward, rollbacks, seed=n_forward)
    o = oracle.events_sorter_build(q, capacity)
    w = ctx.compute_events_dedup_and_sort(q, capacity)
    n_inst = o["instances"].size
    t = native.Trace(ctx, n_rows, n_inst)
    ctx.synthesize_events_sorter(w, t)
    for idx in range(n_inst):
        got = t.get(idx)[:139]
        exp = oracle.events_sorter_synthesize(o, idx, capacity, n_rows)
        if not np.array_equal(got, exp):
            bad = np.argwhere(got != exp)
            raise AssertionError(f"instance {idx}: {len(bad)} cells differ, first (col, row) = {bad[:8].tolist()}")
        assert ctx.check_if_satisfied_events_sorter(t, idx, capacity)[0] == 0
    t.free()


def test_production_geometry(ctx, oracle):
    """capacity 31 287 in a 2^20-row trace: one full instance and a ragged last one."""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 31287, 1 << 20
    q = synthetic.events_trace(30000, 0.2, seed=9)
    assert q.size > capacity
    w = ctx.compute_events_dedup_and_sort(q, capacity)
    t = native.Trace(ctx, n_rows, 2)
    ctx.synthesize_events_sorter(w, t)
    for idx in range(2):
        bad, first = ctx.check_if_satisfied_events_sorter(t, idx, capacity)
        assert bad == 0, (idx, first)
        mult = t.get(idx, 138, 1)[0]
        assert int(mult.sum()) == 8 * n_rows and not mult[256:].any()
    t.free()


def test_gpu_checker_flags_tampering(ctx, oracle):
    import torch

    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 64, 2048
    q = synthetic.events_trace(40, 0.4, seed=3)
    w = ctx.compute_events_dedup_and_sort(q, capacity)
    t = native.Trace(ctx, n_rows, 1)
    ctx.synthesize_events_sorter(w, t, 0, 1)
    assert ctx.check_if_satisfied_events_sorter(t, 0, capacity)[0] == 0
    host = t.get(0)
    rng = np.random.default_rng(2)
    used = np.argwhere(host[:138, :13 * 64 + 56] != 0)
    base = native.load().zkw_trace_device_ptr(t.handle, 0)
    hip = C.CDLL("libamdhip64.so")
    for _ in range(25):
        c, r = used[rng.integers(len(used))]
        addr = base + (int(c) * n_rows + int(r)) * 8
        old = np.array([host[c, r]], np.uint64)
        new = np.array([(int(host[c, r]) + 1) % P], np.uint64)
        torch.cuda.synchronize()
        hip.hipMemcpy(C.c_void_p(addr), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
        assert ctx.check_if_satisfied_events_sorter(t, 0, capacity)[0] > 0, (c, r)
        hip.hipMemcpy(C.c_void_p(addr), old.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
    assert ctx.check_if_satisfied_events_sorter(t, 0, capacity)[0] == 0
    # the closed-form section: challenges, start-flag selection, commitments, public input — same verdict as the oracle's checker
    from closed_form_case import gpu_tamper_parity, events_sorter_tampers
    gpu_tamper_parity(base, n_rows, host, lambda: ctx.check_if_satisfied_events_sorter(t, 0, capacity), lambda h: oracle.events_sorter_check(h, capacity),
                      events_sorter_tampers(capacity))
    t.free()


def test_compact_forms_and_public_inputs(ctx, oracle):
    """a20: compact closed-form inputs and public-input commitments of every instance, GPU vs oracle"""
    from era_zkevm_test_harness_amd import native as nv

    q = synthetic.events_trace(80, 0.3, seed=11)
    o = oracle.events_sorter_build(q, 32)
    w = ctx.compute_events_dedup_and_sort(q, 32)
    compact, pi = oracle.events_sorter_public_inputs(o["instances"])
    assert o["instances"].size >= 3
    assert np.array_equal(w.get(nv.EVT_COMPACT_FORMS), compact)
    assert np.array_equal(w.get(nv.EVT_PUBLIC_INPUTS), pi)


def test_empty_queue_dummy_instance(ctx, oracle):
    """no records: the one dummy instance of the reference, synthesized and satisfied, bit-exact vs the oracle"""
    from era_zkevm_test_harness_amd import native

    q = np.zeros(0, oracle.LOG_QUERY)
    o = oracle.events_sorter_build(q, 16)
    w = ctx.compute_events_dedup_and_sort(q, 16)
    assert w.num_instances == 1
    t = native.Trace(ctx, 2048, 1)
    ctx.synthesize_events_sorter(w, t)
    assert np.array_equal(t.get(0)[:139], oracle.events_sorter_synthesize(o, 0, 16, 2048))
    assert ctx.check_if_satisfied_events_sorter(t, 0, 16)[0] == 0
    t.free()


def test_slot_reuse_keeps_the_padding_rows(ctx, oracle):
    """as tests/test_gpu_ram_synthesis.py::test_slot_reuse_keeps_the_padding_rows, for the events sorter's tail kernel"""
    from era_zkevm_test_harness_amd import native

    n_rows = 4096
    q = synthetic.events_trace(150, 0.3, seed=5)
    t = native.Trace(ctx, n_rows, 1)
    for capacity in (64, 32, 64):
        w = ctx.compute_events_dedup_and_sort(q, capacity)
        o = oracle.events_sorter_build(q, capacity)
        for idx in range(o["instances"].size):
            ctx.synthesize_events_sorter(w, t, idx, 1, 0)
            assert np.array_equal(t.get(0)[:139], oracle.events_sorter_synthesize(o, idx, capacity, n_rows)), (capacity, idx)
        w.free()
    t.free()
