"""GPU: the ECRecover circuit (type 7) through the C ABI against the oracle (oracle/ecrecover_circuit.c): traces cell for cell —
Keccak-f netlist over the recovered key, queue section, EC section (secp256k1 over 16-bit limbs) — on real signatures and on the
failures the precompile has; recovered addresses against public vectors; checker verdict parity on tampered cells of every region
(a limb, a carry, a quotient limb, a FixedBaseMul table cell, a Xor8 range cell, inputs, outputs, the links); the reference's
capacity (7 requests in 2^20 rows) through the GPU checker."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu
N_ROWS = 1 << 18


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def _build(ctx, oracle, n_req, cap, seed=3, req_mq=None):
    from era_zkevm_test_harness_amd import native

    req, mq = req_mq if req_mq is not None else synthetic.precompile_trace(2, n_req, seed=seed)
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    return ctx._precompile(2, req, tails, mq, cap, mem_in), oracle.precompile_build(2, req, tails, mq, cap, mem_in)


def test_traces_match_oracle_and_tamper_parity(ctx, oracle):
    import ctypes

    from era_zkevm_test_harness_amd import native

    cap = 2
    w, o = _build(ctx, oracle, 5, cap)
    ni = w.num_instances
    assert ni == 3 == o["instances"].size
    lay = native.circuit_layout(7, cap)
    g = oracle.ec_geometry(cap)
    assert int(lay["num_columns"]) == native.EK_COLS == oracle.nl_geometry(7)["cols"] and int(lay["total_table_rows"]) == 197632
    assert int(lay["ec_first_row"]) == g["first_row"] and int(lay["ec_rows_per_cycle"]) == g["rows_per_cycle"]
    t = native.Trace(ctx, N_ROWS, ni, n_cols=native.EK_COLS)
    ctx.synthesize_ecrecover(w, t, 0, ni, 0)
    for i in range(ni):
        exp = oracle.ecrecover_synthesize(o, i, cap, N_ROWS)
        got = t.get(i)
        assert np.array_equal(got, exp), np.argwhere(got != exp)[:4]
        assert ctx.check_if_satisfied_ecrecover(t, i, cap) == (0, (0, 0, 0))
        assert ctx.check_if_satisfied(7, t, i, cap) == (0, (0, 0, 0))  # the type-dispatching entry point
        assert got[:4, int(lay["public_input_row"][0])].tolist() == oracle.closed_form_public_inputs(7, o["instances"])[1][i].tolist()
    # the same instances through zkw_synthesize, into a slot that keeps its layout tag
    ctx.synthesize(7, w, t, 2, 1, 0)
    assert np.array_equal(t.get(0), oracle.ecrecover_synthesize(o, 2, cap, N_ROWS))
    # tamper parity on instance 0
    base = oracle.ecrecover_synthesize(o, 0, cap, N_ROWS)
    rng = np.random.default_rng(7)
    cells = [oracle.ec_cell(cap, 0, "glob", 320), oracle.ec_cell(cap, 0, "glob", 321), oracle.ec_cell(cap, 1, "in", 70), oracle.ec_cell(cap, 0, "in", 3),
             oracle.ec_cell(cap, 0, "key", 5), oracle.ec_cell(cap, 1, "key", 63), oracle.ec_cell(cap, 0, "glob", 40), oracle.ec_cell(cap, 0, "glob", 300)]
    c0, r0 = oracle.ec_cell(cap, 0, "mul", 2)
    cells += [(c0 + 3, r0), (c0 + 20, r0), (c0 + 40, r0), (c0 + 47, r0), (c0 + 50, r0), (c0 + 70, r0), (79, r0)]  # a, b, q, q's top limb, r, a carry, the spare cell
    fix_row = g["first_row"] + g["rows_per_cycle"] - 17 - 22 * 32  # first row of the first FIX segment: FixedBaseMul<0, 0>
    cells += [(80, fix_row), (81, fix_row), (82, fix_row), (83, fix_row), (81, fix_row + 5)]  # the byte, x word, y word, a padding slot
    cells += [(int(rng.integers(0, 80)), int(rng.integers(g["first_row"], g["rows_used"]))) for _ in range(12)]
    cells += [(int(rng.integers(80, 128)), int(rng.integers(g["first_row"], g["rows_used"]))) for _ in range(12)]
    cells += [(128, 5), (128, 131072 + 7), (128, 131072 + 256 * 9)]  # multiplicities: Xor8, two FixedBaseMul tables
    cells += [(int(rng.integers(0, 128)), int(rng.integers(0, 2 * 2073))) for _ in range(6)]  # the netlist
    qg = oracle.nlq_geometry(7, cap)
    cells += [(int(rng.integers(0, 80)), int(rng.integers(qg["first_row"], qg["rows_used"]))) for _ in range(8)]  # the queue section
    cells += [oracle.nlq_cell(7, cap, 0, 3, -1, 0, 6 + 2), oracle.nlq_cell(7, cap, 1, 5, -1, 0, 6), oracle.nlq_cell(7, cap, 0, 6, -1, 0, 6 + 19)]  # a read byte, ok, an address byte
    from nlcf_cells import closed_form_cells
    cells += closed_form_cells(oracle, 7, cap, rng)  # the closed-form section below the EC section
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n_flagged = 0
    for col, row in cells:
        bad = base.copy()
        bad[col, row] = bad[col, row] + 1 if rng.random() < 0.7 else 300
        ctx.synchronize()
        assert hip.hipMemcpy(t.device_ptr(0), bad.ctypes.data, bad.nbytes, 1) == 0
        got = ctx.check_if_satisfied_ecrecover(t, 0, cap)
        want = oracle.ecrecover_check(bad, cap)
        assert got == want, ((col, row), got, want)
        n_flagged += got[0] > 0
    assert n_flagged >= len(cells) - 3
    t.free()
    w.free()


def test_public_vectors_and_failures(ctx, oracle):
    """the addresses the circuit writes: go-ethereum's ecrecover precompile vector, the key-1 address, and every failure mode"""
    from era_zkevm_test_harness_amd import native, secp256k1 as ec

    h0 = 0x456E9AEA5E197A1F1AF7A3E85A3212FA4049A3BA34C2289B4C860FC0B0C64EF3
    v1, r1, s1 = ec.sign(h0, 1, 0xC0FFEE)
    x = 5
    while ec.lift_x(x, 0) is not None:
        x += 1
    cases = [(h0, 1, 0x9242685BF161793CC25603C231BC2F568EB630EA16AA137D2664AC8038825608, 0x4F8AE3BD7535248D0BD448298CC2E2071E56992D0774DC340C368AE950852ADA),
             (h0, v1, r1, s1), (h0, v1, ec.N, s1), (h0, v1, r1, 0), (h0, 0, x, s1), (0, v1, r1, s1), (h0, v1, r1, ec.N + 3)]
    want = [ec.ecrecover(*c) for c in cases]
    assert want[0] == (1, 0x7156526FBD7A3C72969B54F64E42C10FBB768C8A) and want[1] == (1, 0x7E5F4552091A69125D5DFCB7B8C2659029395BDF)
    assert [wnt[0] for wnt in want] == [1, 1, 0, 0, 0, 1, 0]
    req, mq = synthetic.precompile_trace(2, len(cases), seed=9)
    for k, (c, (ok, addr)) in enumerate(zip(cases, want)):
        for j, val in enumerate(c + (ok, addr)):
            mq["value"][6 * k + j] = np.frombuffer(int(val).to_bytes(32, "little"), "<u4")
    cap = len(cases)
    w, o = _build(ctx, oracle, 0, cap, req_mq=(req, mq))
    assert w.num_instances == 1
    t = native.Trace(ctx, N_ROWS, 1, n_cols=native.EK_COLS)
    ctx.synthesize_ecrecover(w, t, 0, 1, 0)
    assert ctx.check_if_satisfied_ecrecover(t, 0, cap) == (0, (0, 0, 0))
    got = t.get(0)
    assert np.array_equal(got, oracle.ecrecover_synthesize(o, 0, cap, N_ROWS))
    # the written values as the trace holds them: the value bytes of the two writes of every cycle (queue section)
    for k, (ok, addr) in enumerate(want):
        okc = [got[oracle.nlq_cell(7, cap, k, 5, -1, 0, 6 + b)] for b in range(32)]
        adc = [got[oracle.nlq_cell(7, cap, k, 6, -1, 0, 6 + b)] for b in range(32)]
        assert int.from_bytes(bytes(int(b) for b in okc), "little") == ok and int.from_bytes(bytes(int(b) for b in adc), "little") == addr
    # a write query that does not hold the circuit's result is a violation (the copy constraint of the digest link)
    mq2 = mq.copy()
    mq2["value"][6 * 1 + 5][0] ^= 1
    w2, _o2 = _build(ctx, oracle, 0, cap, req_mq=(req, mq2))
    ctx.synthesize_ecrecover(w2, t, 0, 1, 0)
    assert ctx.check_if_satisfied_ecrecover(t, 0, cap)[0] > 0
    t.free()
    w.free()
    w2.free()


def test_production_geometry(ctx, oracle):
    """2^20 rows at the reference's capacity (7 requests per instance): three instances in one call, the last one partial"""
    from era_zkevm_test_harness_amd import native

    cap = int(native.circuit_geometry(7)["capacity"])
    assert cap == 7
    lay = native.circuit_layout(7)
    assert int(lay["fits"]) == 1 and int(lay["rows_used"]) == 197632  # the tables' rows: 7 cycles use fewer
    w, o = _build(ctx, oracle, 17, cap, seed=11)
    assert w.num_instances == 3
    n_rows = 1 << 20
    t = native.Trace(ctx, n_rows, 3, n_cols=native.EK_COLS)
    ctx.synthesize_ecrecover(w, t, 0, 3, 0)
    for i in range(3):
        assert ctx.check_if_satisfied_ecrecover(t, i, cap) == (0, (0, 0, 0))
    exp = oracle.ecrecover_synthesize(o, 2, cap, n_rows)
    assert np.array_equal(t.get(2), exp)
    t.free()
    w.free()


def test_capacity_beyond_eight_cycles_per_instance(ctx, oracle):
    """capacity 9: the tape of an instance interleaves its cycles sixteen to a value (the stride is the capacity rounded up to 8) and the row
    kernel takes two groups of eight cycles — trace == oracle cell for cell, a full instance and one with idle cycles"""
    from era_zkevm_test_harness_amd import native

    cap = 9
    w, o = _build(ctx, oracle, 11, cap, seed=21)
    assert w.num_instances == 2 == o["instances"].size
    t = native.Trace(ctx, N_ROWS, 2, n_cols=native.EK_COLS)
    ctx.synthesize_ecrecover(w, t, 0, 2, 0)
    for i in range(2):
        assert np.array_equal(t.get(i), oracle.ecrecover_synthesize(o, i, cap, N_ROWS)), i
        assert ctx.check_if_satisfied_ecrecover(t, i, cap) == (0, (0, 0, 0))
    t.free()
    w.free()
