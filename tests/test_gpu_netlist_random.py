"""GPU: randomized parity of the netlist circuits (types 5, 6, 3) against the oracle — request counts, round counts and
capacities drawn so that instance boundaries fall on request boundaries, inside requests and on the last round, incl.
capacity 1 and capacities larger than the whole block. Every instance: round records, trace cell for cell, both checkers."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu
N_ROWS = 1 << 18  # the stacked Keccak tables alone are 132 096 rows


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def _dirty(ctx, t, n_cols):
    """garbage in the whole slot: the synthesis must leave no cell of a previous tenant behind (it zeroes only what its fill
    does not write: nl_synthesize, zkw_precompiles.hip)"""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    ctx.synchronize()
    assert hip.hipMemset(t.device_ptr(0), 0xA5, n_cols * N_ROWS * 8) == 0
    assert hip.hipDeviceSynchronize() == 0


def _compare(ctx, native, t, i, got_fn, exp, check_gpu, check_orc, capacity):
    got = t.get(0)
    if not np.array_equal(got, exp):
        c, r = np.argwhere(got != exp)[0]
        raise AssertionError(f"instance {i}: first difference at column {c} row {r}: {got[c, r]} != {exp[c, r]}")
    assert check_gpu(t, 0, capacity) == (0, (0, 0, 0)) and check_orc(exp, capacity)[0] == 0


@pytest.mark.parametrize("seed", range(6))
def test_precompile_circuits_random(ctx, oracle, seed):
    from era_zkevm_test_harness_amd import native

    rng = np.random.default_rng(100 + seed)
    for kind, cols, rounds_key, synth, check, osynth, ocheck, max_cap in (
            (0, native.KC_COLS, "keccak_rounds", ctx.synthesize_keccak_round_function, ctx.check_if_satisfied_keccak_round_function,
             oracle.keccak_round_synthesize, oracle.keccak_round_check, 30),
            (1, native.SC_COLS, "sha256_rounds", ctx.synthesize_sha256_round_function, ctx.check_if_satisfied_sha256_round_function,
             oracle.sha256_round_synthesize, oracle.sha256_round_check, 120)):
        n_req = int(rng.integers(1, 9))
        req, mq = synthetic.precompile_trace(kind, n_req, seed=int(rng.integers(1 << 30)), max_rounds=int(rng.integers(1, 7)))
        tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
        mem_in = np.zeros(1, native.QUEUE_STATE12)
        total = int(oracle.precompile_build(kind, req, tails, mq, 1 << 20, mem_in)["instances"]["num_rounds"].sum())
        capacity = int(rng.choice([1, max(1, total // 2), total, total + 3, int(rng.integers(1, max_cap))]))
        o = oracle.precompile_build(kind, req, tails, mq, capacity, mem_in)
        w = ctx._precompile(kind, req, tails, mq, capacity, mem_in)
        assert w.get(native.PRC_KECCAK_ROUNDS if kind == 0 else native.PRC_SHA256_ROUNDS).tobytes() == o[rounds_key].tobytes()
        t = native.Trace(ctx, N_ROWS, 1, n_cols=cols)
        for i in sorted({0, w.num_instances // 2, w.num_instances - 1}):
            _dirty(ctx, t, cols)
            synth(w, t, i, 1, 0)
            _compare(ctx, native, t, i, None, osynth(o, i, capacity, N_ROWS), check, ocheck, capacity)
        t.free()
        w.free()


@pytest.mark.parametrize("seed", range(3))
def test_code_decommitter_random(ctx, oracle, seed):
    from era_zkevm_test_harness_amd import native

    rng = np.random.default_rng(200 + seed)
    b = synthetic.block_after_vm(seed=30 + seed, max_code_words=int(rng.integers(3, 40)))
    dec = ctx.compute_decommitts_sorter_circuit_snapshots(b["decommit_queries"], 5)
    dq, dt = dec.get(native.DEC_DEDUP_QUERIES), dec.get(native.DEC_DEDUP_TAILS)
    codes = [b["bytecodes"][h.tobytes()] for h in dq["hash"]]
    woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
    words = np.concatenate(codes)
    total = sum((c.shape[0] + 1) // 2 for c in codes)
    capacity = int(rng.choice([1, total // 3 + 1, total, total + 2]))
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    w = ctx.compute_decommitter_circuit_snapshots(dq, dt, words, woff, capacity, mem_in)
    o = oracle.decommitter_build(dq, dt, words, woff, capacity, mem_in)
    assert w.get(native.DCM_SHA256_ROUNDS).tobytes() == o["sha256_rounds"].tobytes()
    t = native.Trace(ctx, N_ROWS, 1, n_cols=native.DC_COLS)
    for i in sorted({0, w.num_instances // 2, w.num_instances - 1}):
        _dirty(ctx, t, native.DC_COLS)
        ctx.synthesize_code_decommitter(w, t, i, 1, 0)
        _compare(ctx, native, t, i, None, oracle.code_decommitter_synthesize(o, i, capacity, N_ROWS), ctx.check_if_satisfied_code_decommitter,
                 oracle.code_decommitter_check, capacity)
    t.free()
    w.free()
    dec.free()


@pytest.mark.parametrize("seed", range(4))
def test_storage_application_random(ctx, oracle, seed):
    """type 10: random numbers of slots, read / write mixes and capacities (2 = one write per instance .. larger than the block):
    every instance cut the reference's chunking rule makes, dirty slots, both checkers"""
    from era_zkevm_test_harness_amd import native
    from sap_case import storage_application_case

    rng = np.random.default_rng(300 + seed)
    n = int(rng.integers(1, 12))
    q, tails, tree, idx, paths = storage_application_case(oracle, n, seed=400 + seed)
    q["rw_flag"] = (rng.random(n) < rng.random()).astype(np.uint8)
    ro = q["rw_flag"] == 0
    q["written_value"][ro] = q["read_value"][ro]
    capacity = int(rng.choice([2, 3, 5, 7]))  # 7 walks x 257 cycles x 122 rows < 2^18 rows
    w = ctx.decompose_into_storage_application_witnesses(q, tails, idx, paths, tree.root, tree.next_enumeration_index, capacity)
    o = oracle.storage_application_build(tree, q, tails, capacity)
    assert w.num_instances == o["instances"].size
    t = native.Trace(ctx, N_ROWS, 1, n_cols=native.SA_COLS)
    for i in sorted({0, w.num_instances // 2, w.num_instances - 1}):
        _dirty(ctx, t, native.SA_COLS)
        ctx.synthesize_storage_application(w, t, i, 1, 0)
        _compare(ctx, native, t, i, None, oracle.storage_application_synthesize(o, q, i, capacity, N_ROWS),
                 ctx.check_if_satisfied_storage_application, oracle.storage_application_check, capacity)
    t.free()
    w.free()
