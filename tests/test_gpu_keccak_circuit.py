"""GPU: Keccak256RoundFunction synthesis ("zkw trace v3", csrc/keccak_circuit_kernels.cuh) — cell-exact against the oracle's
fill (oracle/keccak_circuit.c) on every instance of a small block, the GPU checker against the oracle's on clean and
tampered traces (same violation kind), production geometry (2^20 rows, capacity 293) through the GPU checker."""
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

    req, mq = synthetic.precompile_trace(0, n_req, seed=seed, max_rounds=max_rounds)
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1] if n_req else np.zeros((0, 4), np.uint64)
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    o = oracle.precompile_build(0, req, tails, mq, capacity, mem_in)
    w = ctx._precompile(0, req, tails, mq, capacity, mem_in)
    return o, w


@pytest.mark.parametrize("n_req,capacity", [(9, 6), (0, 3), (2, 30)])
def test_keccak_round_function_trace_matches_the_oracle(ctx, oracle, n_req, capacity):
    from era_zkevm_test_harness_amd import native

    n_rows = 1 << 16
    o, w = _build(ctx, oracle, n_req, capacity, seed=3)
    ni = w.num_instances
    assert ni == o["instances"].size
    t = native.Trace(ctx, n_rows, ni, n_cols=native.KC_COLS)
    ctx.synthesize_keccak_round_function(w, t)
    for i in range(ni):
        exp = oracle.keccak_round_synthesize(o, i, capacity, n_rows)
        got = t.get(i)
        assert got.shape == exp.shape
        if not np.array_equal(got, exp):
            c, r = np.argwhere(got != exp)[0]
            raise AssertionError(f"instance {i}: first difference at column {c} row {r}: {got[c, r]} != {exp[c, r]}")
        assert ctx.check_if_satisfied_keccak_round_function(t, i, capacity) == (0, (0, 0, 0))
        assert oracle.keccak_round_check(got, capacity)[0] == 0
    t.free()
    w.free()


def test_gpu_checker_flags_tampering_like_the_oracle(ctx, oracle):
    import torch

    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 4, 1 << 16
    o, w = _build(ctx, oracle, 5, capacity, seed=11)
    t = native.Trace(ctx, n_rows, 1, n_cols=native.KC_COLS)
    ctx.synthesize_keccak_round_function(w, t, 1, 1)
    assert ctx.check_if_satisfied_keccak_round_function(t, 0, capacity)[0] == 0
    host = t.get(0)
    base = native.load().zkw_trace_device_ptr(t.handle, 0)
    hip = C.CDLL("libamdhip64.so")
    rng = np.random.default_rng(5)
    cyc = oracle.KC_ROWS_PER_CYCLE
    cells = [(0, cyc), (2, cyc), (7, cyc + 9), (86, 0), (128, 77), (3, capacity * cyc + 3), (50, capacity * cyc + 9)]
    used = np.argwhere(host[86:128, :capacity * cyc] != 0)
    cells += [(int(c) + 86, int(r)) for c, r in used[rng.integers(len(used), size=40)]]
    for c, r in cells:
        addr = base + (c * n_rows + r) * 8
        old = np.array([host[c, r]], np.uint64)
        new = np.array([int(host[c, r]) + 1], np.uint64)
        torch.cuda.synchronize()
        hip.hipMemcpy(C.c_void_p(addr), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
        bad = host.copy()
        bad[c, r] += 1
        n, first = ctx.check_if_satisfied_keccak_round_function(t, 0, capacity)
        on, ofirst = oracle.keccak_round_check(bad, capacity)
        assert n > 0 and on > 0 and n == on and first == ofirst, ((c, r), n, first, on, ofirst)
        hip.hipMemcpy(C.c_void_p(addr), old.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1)
    assert ctx.check_if_satisfied_keccak_round_function(t, 0, capacity)[0] == 0
    t.free()
    w.free()


def test_production_geometry(ctx, oracle):
    """2^20 rows, capacity 293 (geometry_config.rs), two full instances and a partly idle one: satisfied; the multiplicity
    columns sum to 14 lookups per table row of the cycle region"""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 293, 1 << 20
    o, w = _build(ctx, oracle, 160, capacity, seed=21, max_rounds=6)
    ni = w.num_instances
    assert ni >= 2
    t = native.Trace(ctx, n_rows, 2, n_cols=native.KC_COLS)
    for i in (0, ni - 1):
        ctx.synthesize_keccak_round_function(w, t, i, 1, 0)
        assert ctx.check_if_satisfied_keccak_round_function(t, 0, capacity) == (0, (0, 0, 0)), i
        mult = t.get(0, 128, 9)
        assert int(mult.sum()) == 14 * (oracle.KC_ROWS_PER_CYCLE - 1) * capacity and not mult[:, 65536:].any()
    # the last cells of BND_OUT are the sponge state after the instance's last round
    n = int(o["instances"]["num_rounds"][ni - 1])
    first = int(o["instances"]["first_round"][ni - 1])
    bnd = capacity * oracle.KC_ROWS_PER_CYCLE
    tr = t.get(0, 0, 86)
    out = np.concatenate([tr[:86, bnd + 3], tr[:86, bnd + 4], tr[:28, bnd + 5]]).astype(np.uint8)
    assert out.tobytes() == o["keccak_rounds"]["state_after"][first + n - 1].tobytes()
    t.free()
    w.free()


@pytest.mark.parametrize("n_msg,capacity,n_rows", [(0, 4, 1 << 16), (7, 20, 1 << 16), (20, 20, 1 << 16), (774, 774, 1 << 20), (301, 774, 1 << 20)])
def test_linear_hasher_circuit(ctx, oracle, n_msg, capacity, n_rows):
    """type 13 (LinearHasher): cell-exact against the oracle, satisfied, the record and the public input identical; the
    reference's capacity (774 messages = 501 cycles in 2^20 rows) included"""
    from era_zkevm_test_harness_amd import native

    q = synthetic.mixed_log_queue(4 * n_msg + 8, seed=n_msg + 1)[:n_msg]
    qs = np.zeros(1, native.QUEUE_STATE4)
    qs["tail"] = synthetic.random_field_elements(n_msg + 2, (4,))
    qs["length"] = n_msg
    t = native.Trace(ctx, n_rows, 1, n_cols=native.KC_COLS)
    rec, pi = ctx.synthesize_linear_hasher(q, qs, capacity, t, 0)
    cycles = native.linear_hasher_cycles(capacity)
    assert ctx.check_if_satisfied_keccak_round_function(t, 0, cycles) == (0, (0, 0, 0))
    exp, orec, opi = oracle.linear_hasher_synthesize(q, qs, capacity, n_rows)
    assert rec.tobytes() == orec.tobytes() and pi.tolist() == opi.tolist()
    assert rec["keccak256_hash"][0].tobytes() == ctx.compute_linear_keccak256(q)
    got = t.get(0)
    if not np.array_equal(got, exp):
        c, r = np.argwhere(got != exp)[0]
        raise AssertionError(f"first difference at column {c} row {r}: {got[c, r]} != {exp[c, r]}")
    t.free()


def test_linear_hasher_rejects_too_many_messages(ctx):
    from era_zkevm_test_harness_amd import native

    q = synthetic.mixed_log_queue(64, seed=2)[:9]
    t = native.Trace(ctx, 1 << 16, 1, n_cols=native.KC_COLS)
    with pytest.raises(native.ZkwError) as ei:
        ctx.synthesize_linear_hasher(q, np.zeros(1, native.QUEUE_STATE4), 8, t, 0)
    assert ei.value.code == native.ERR_INVALID
    t.free()
