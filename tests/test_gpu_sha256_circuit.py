"""GPU: Sha256RoundFunction synthesis ("zkw trace v3", csrc/sha256_circuit_kernels.cuh: byte lookups + 32-bit ADD gates) —
the builder's round records and the filled trace cell-exact against the oracle, the GPU checker against the oracle's on
clean and tampered traces, production geometry (2^20 rows, capacity 2206) through the GPU checker."""
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


def _build(ctx, oracle, n_req, capacity, seed, max_rounds=4):
    from era_zkevm_test_harness_amd import native

    req, mq = synthetic.precompile_trace(1, n_req, seed=seed, max_rounds=max_rounds)
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1] if n_req else np.zeros((0, 4), np.uint64)
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    o = oracle.precompile_build(1, req, tails, mq, capacity, mem_in)
    w = ctx._precompile(1, req, tails, mq, capacity, mem_in)
    return o, w


@pytest.mark.parametrize("n_req,capacity", [(9, 7), (0, 3), (2, 40)])
def test_sha256_round_function_trace_matches_the_oracle(ctx, oracle, n_req, capacity):
    from era_zkevm_test_harness_amd import native

    n_rows = 1 << 16
    o, w = _build(ctx, oracle, n_req, capacity, seed=3)
    assert w.get(native.PRC_SHA256_ROUNDS).tobytes() == o["sha256_rounds"].tobytes()
    ni = w.num_instances
    assert ni == o["instances"].size
    t = native.Trace(ctx, n_rows, ni, n_cols=native.SC_COLS)
    ctx.synthesize_sha256_round_function(w, t)
    for i in range(ni):
        exp = oracle.sha256_round_synthesize(o, i, capacity, n_rows)
        got = t.get(i)
        if not np.array_equal(got, exp):
            c, r = np.argwhere(got != exp)[0]
            raise AssertionError(f"instance {i}: first difference at column {c} row {r}: {got[c, r]} != {exp[c, r]}")
        assert ctx.check_if_satisfied_sha256_round_function(t, i, capacity) == (0, (0, 0, 0))
    t.free()
    w.free()


def test_gpu_checker_flags_tampering_like_the_oracle(ctx, oracle):
    import torch

    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 5, 1 << 16
    o, w = _build(ctx, oracle, 6, capacity, seed=11)
    t = native.Trace(ctx, n_rows, 1, n_cols=native.SC_COLS)
    ctx.synthesize_sha256_round_function(w, t, 1, 1)
    assert ctx.check_if_satisfied_sha256_round_function(t, 0, capacity)[0] == 0
    host = t.get(0)
    base = native.load().zkw_trace_device_ptr(t.handle, 0)
    hip = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(5)
    cyc = oracle.SC_ROWS_PER_CYCLE
    cells = [(0, cyc), (2, cyc), (24, cyc + 5), (28, cyc + 5), (0, cyc + 5), (40, cyc + 5), (70, cyc + 200), (86, cyc), (128, 77),
             (3, capacity * cyc + 1), (50, capacity * cyc + 9)]
    used = np.argwhere(host[:128, :capacity * cyc] != 0)
    cells += [(int(c), int(r)) for c, r in used[rng.integers(len(used), size=40)]]
    for c, r in cells:
        addr = base + (c * n_rows + r) * 8
        old = np.array([host[c, r]], np.uint64)
        new = np.array([int(host[c, r]) + 1], np.uint64)
        torch.cuda.synchronize()
        hip.hipMemcpy(C.c_void_p(addr), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
        bad = host.copy()
        bad[c, r] += 1
        n, first = ctx.check_if_satisfied_sha256_round_function(t, 0, capacity)
        on, ofirst = oracle.sha256_round_check(bad, capacity)
        assert n > 0 and on > 0 and n == on and first == ofirst, ((c, r), n, first, on, ofirst)
        hip.hipMemcpy(C.c_void_p(addr), old.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
    assert ctx.check_if_satisfied_sha256_round_function(t, 0, capacity)[0] == 0
    t.free()
    w.free()


def test_production_geometry(ctx, oracle):
    """2^20 rows, capacity 2206 (geometry_config.rs): a full instance and a partly idle one satisfied; multiplicities add up"""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 2206, 1 << 20
    o, w = _build(ctx, oracle, 900, capacity, seed=21, max_rounds=6)
    ni = w.num_instances
    assert ni >= 2
    t = native.Trace(ctx, n_rows, 1, n_cols=native.SC_COLS)
    for i in (0, ni - 1):
        ctx.synthesize_sha256_round_function(w, t, i, 1, 0)
        assert ctx.check_if_satisfied_sha256_round_function(t, 0, capacity) == (0, (0, 0, 0)), i
        mult = t.get(0, 128, 10)
        assert int(mult.sum()) == 14 * (oracle.SC_ROWS_PER_CYCLE - 1) * capacity and not mult[:, 65536:].any()
    n = int(o["instances"]["num_rounds"][ni - 1])
    first = int(o["instances"]["first_round"][ni - 1])
    bnd = capacity * oracle.SC_ROWS_PER_CYCLE
    out = t.get(0, 0, 32)[:, bnd + 1].astype(np.uint8)
    assert out.tobytes() == o["sha256_rounds"]["state_after"][first + n - 1].astype("<u4").tobytes()
    t.free()
    w.free()


@pytest.mark.parametrize("capacity,n_rows", [(7, 1 << 16), (2845, 1 << 20)])
def test_code_decommitter_circuit(ctx, oracle, capacity, n_rows):
    """type 3 (CodeDecommitter): round records and traces cell-exact against the oracle, satisfied; the reference's capacity
    (2845 rounds in 2^20 rows) included"""
    from era_zkevm_test_harness_amd import block as blk, native

    big = capacity > 100
    b = synthetic.block_production(seed=3) if big else synthetic.block_after_vm(seed=2)
    dec = ctx.compute_decommitts_sorter_circuit_snapshots(b["decommit_queries"], 117500 if big else 5)
    dq, dt = dec.get(native.DEC_DEDUP_QUERIES), dec.get(native.DEC_DEDUP_TAILS)
    codes = [b["bytecodes"][h.tobytes()] for h in dq["hash"]]
    woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
    words = np.concatenate(codes)
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    w = ctx.compute_decommitter_circuit_snapshots(dq, dt, words, woff, capacity, mem_in)
    o = oracle.decommitter_build(dq, dt, words, woff, capacity, mem_in)
    assert w.get(native.DCM_SHA256_ROUNDS).tobytes() == o["sha256_rounds"].tobytes()
    ni = w.num_instances
    assert ni == o["instances"].size and (ni >= 3 or big)
    t = native.Trace(ctx, n_rows, 1, n_cols=native.DC_COLS)
    for i in sorted({0, ni - 1}):
        ctx.synthesize_code_decommitter(w, t, i, 1, 0)
        assert ctx.check_if_satisfied_code_decommitter(t, 0, capacity) == (0, (0, 0, 0)), i
        exp = oracle.code_decommitter_synthesize(o, i, capacity, n_rows)
        got = t.get(0)
        if not np.array_equal(got, exp):
            c, r = np.argwhere(got != exp)[0]
            raise AssertionError(f"instance {i}: first difference at column {c} row {r}: {got[c, r]} != {exp[c, r]}")
    t.free()
    w.free()
    dec.free()
