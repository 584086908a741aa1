"""GPU: zkw_check_copy_permutation — traces synthesized on the GPU satisfy the sigma columns of zkw_setup_copy_permutation
(RAMPermutation and LogDemuxer), a bumped cell of a copy cycle is reported at its position."""
import ctypes as C

import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def test_gpu_traces_satisfy_sigma(ctx):
    import torch

    from era_zkevm_test_harness_amd import native

    n_rows = 1 << 15
    w = ctx.compute_ram_circuit_snapshots(synthetic.ram_trace(2983, seed=5), 1000, 0)
    t = native.Trace(ctx, n_rows, 3)
    ctx.synthesize_ram(w, t, 0, 3, 0)
    sigma = native.setup_copy_permutation(8, 1000, n_rows)
    for slot in range(3):
        assert ctx.check_copy_permutation(t, slot, sigma) == (0, (0, 0, 0))
    flat = sigma.reshape(-1)
    moved = np.flatnonzero(flat != np.arange(flat.size, dtype=np.uint64))
    host = t.get(1)
    base = native.load().zkw_trace_device_ptr(t.handle, 1)
    hip = C.CDLL("libamdhip64.so")
    for cell in np.random.default_rng(3).choice(moved, 8, replace=False):
        c, r = int(cell) // n_rows, int(cell) % n_rows
        new, old = np.array([int(host[c, r]) + 1], np.uint64), np.array([host[c, r]], np.uint64)
        torch.cuda.synchronize()
        hip.hipMemcpy(C.c_void_p(base + (c * n_rows + r) * 8), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
        n, first = ctx.check_copy_permutation(t, 1, sigma)
        assert 1 <= n <= 2 and first[0] == 4
        hip.hipMemcpy(C.c_void_p(base + (c * n_rows + r) * 8), old.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
    assert ctx.check_copy_permutation(t, 1, sigma)[0] == 0
    t.free()
    w.free()
    d = ctx.compute_logs_demux(synthetic.mixed_log_queue(900, seed=4), 400)
    t = native.Trace(ctx, n_rows, 1, n_cols=151)
    ctx.synthesize_log_demux(d, t, 0, 1, 0)
    assert ctx.check_copy_permutation(t, 0, native.setup_copy_permutation(4, 400, n_rows)) == (0, (0, 0, 0))
    t.free()
    d.free()


def test_gpu_netlist_traces_satisfy_sigma(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    n_rows = 1 << 18
    for kind, ctype, cap, cols, synth in ((0, 5, 6, native.KC_COLS, ctx.synthesize_keccak_round_function),
                                          (1, 6, 7, native.SC_COLS, ctx.synthesize_sha256_round_function)):
        req, mq = synthetic.precompile_trace(kind, 9, seed=3, max_rounds=4)
        tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
        w = ctx._precompile(kind, req, tails, mq, cap, np.zeros(1, native.QUEUE_STATE12))
        t = native.Trace(ctx, n_rows, w.num_instances, n_cols=cols)
        synth(w, t)
        sigma = native.setup_copy_permutation(ctype, cap, n_rows)
        for slot in range(w.num_instances):
            assert ctx.check_copy_permutation(t, slot, sigma) == (0, (0, 0, 0)), (ctype, slot)
        t.free()
        w.free()
