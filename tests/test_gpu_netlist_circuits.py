"""GPU: the four netlist circuits in "zkw trace v4" (csrc/netlist_kernels.cuh) through the C ABI against the oracle
(oracle/netlist_circuit.c): Keccak256RoundFunction (5), Sha256RoundFunction (6), CodeDecommitter (3), L1MessagesHasher (13) on the
reference's geometry and table sets. Traces cell for cell; the two checkers violation for violation on tampered cells of every
region (header, lookup inputs / outputs, gate cells, multiplicities, boundary rows, padding); production geometry (2^20 rows at
the reference's capacities) through the GPU checker."""
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


def _precompile(ctx, oracle, kind, n_req, cap, seed=3, max_rounds=4):
    from era_zkevm_test_harness_amd import native

    req, mq = synthetic.precompile_trace(kind, n_req, seed=seed, max_rounds=max_rounds)
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    return ctx._precompile(kind, req, tails, mq, cap, mem_in), oracle.precompile_build(kind, req, tails, mq, cap, mem_in)


CIRCUITS = {5: (0, "KC_COLS", "synthesize_keccak_round_function", "check_if_satisfied_keccak_round_function", "keccak_round_synthesize", "keccak_round_check", 6),
            6: (1, "SC_COLS", "synthesize_sha256_round_function", "check_if_satisfied_sha256_round_function", "sha256_round_synthesize", "sha256_round_check", 7)}


@pytest.mark.parametrize("ct", [5, 6])
def test_traces_match_oracle_and_tamper_parity(ctx, oracle, ct):
    from era_zkevm_test_harness_amd import native

    kind, cols, synth, check, osynth, ocheck, cap = CIRCUITS[ct]
    cols = getattr(native, cols)
    w, o = _precompile(ctx, oracle, kind, 9, cap)
    ni = w.num_instances
    assert ni >= 3 and ni == o["instances"].size
    lay = native.circuit_layout(ct, cap)
    assert int(lay["num_columns"]) == cols and int(lay["total_table_rows"]) == oracle.nl_geometry(ct)["table_rows"]
    t = native.Trace(ctx, N_ROWS, ni, n_cols=cols)
    getattr(ctx, synth)(w, t, 0, ni, 0)
    for i in range(ni):
        exp = getattr(oracle, osynth)(o, i, cap, N_ROWS)
        got = t.get(i)
        assert np.array_equal(got, exp), np.argwhere(got != exp)[:4]
        assert getattr(ctx, check)(t, i, cap) == (0, (0, 0, 0))
        assert got[:4, int(lay["public_input_row"][0])].tolist() == oracle.closed_form_public_inputs(ct, o["instances"])[1][i].tolist()
    # tamper parity on instance 1: random cells of every region + the structured ones; upload the tampered slot and compare verdicts
    base = getattr(oracle, osynth)(o, 1, cap, N_ROWS)
    g = oracle.nl_geometry(ct)
    rpc, G = g["rows_per_cycle"], g["general"]
    rng = np.random.default_rng(ct)
    used = cap * rpc
    cells = [(int(rng.integers(0, G)), int(rng.integers(0, used))) for _ in range(14)]                      # general-purpose cells: gates, headers, empties
    cells += [(int(rng.integers(G, cols - 1)), int(rng.integers(0, used))) for _ in range(14)]               # lookup cells
    cells += [(cols - 1, int(rng.integers(0, g["table_rows"]))) for _ in range(4)] + [(cols - 1, g["table_rows"] + 5)]  # multiplicities
    cells += [(int(rng.integers(0, G)), used + k) for k in range(0, 2 * -(-oracle.nl_spec_state(ct) // G) + 3)]          # boundary rows, PI, below
    cells += [(0, rpc), (1, rpc), (2, rpc), (3, 2 * rpc)]                                                   # reset / idle / masks
    if oracle.nlq_geometry(ct, cap)["has"]:  # the queue section: flags, linked nibbles, encodings, states, Poseidon2 variables, QBND, unused
        qg = oracle.nlq_geometry(ct, cap)
        qc = lambda *a, **k: oracle.nlq_cell(ct, cap, *a, **k)  # noqa: E731
        last = qg["ops"] - 1  # the digest write
        cells += [qc(2, 1), qc(2, 0), qc(3, last), qc(2, 1, -1, 0, 16), qc(2, last, -1, 0, 20), qc(2, 2, -1, 0, 1), qc(3, 1, -1, 1, 4), qc(3, 1, -1, 2, 9), qc(3, 2, -1, 3, 2),
                  qc(3, 1, 0, 0, 3), qc(3, 1, 0, 0, 70), qc(3, last, 0, 0, 129), qc(2, 0, -1, 0, 5), qc(2, 0, 1, 0, 40), qc(4, 0, 2, 0, 6), qc(4, 0, -1, 3, 1),
                  qc(0, qg["ops"], k=1), qc(0, qg["ops"], k=4 + 12 + 1), qc(0, qg["ops"], k=G - 1), (G - 1, qc(2, 1)[1]), (G + 2, qc(2, 1)[1]),
                  (int(rng.integers(0, G)), qg["rows_used"] + 2)]
        cells += [(int(rng.integers(0, G)), int(rng.integers(qg["first_row"], qg["rows_used"]))) for _ in range(10)]
    from nlcf_cells import closed_form_cells
    cells += closed_form_cells(oracle, ct, cap, rng)  # the closed-form section: flags, words, ties, sponges, the PI row
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n_flagged = 0
    for col, row in cells:
        bad = base.copy()
        bad[col, row] = bad[col, row] + 1 if rng.random() < 0.7 else 300
        ctx.synchronize()
        assert hip.hipMemcpy(t.device_ptr(0), bad.ctypes.data, bad.nbytes, 1) == 0
        got = getattr(ctx, check)(t, 0, cap)
        want = getattr(oracle, ocheck)(bad, cap)
        assert got == want, ((col, row), got, want)
        n_flagged += got[0] > 0
    assert n_flagged >= len(cells) - 2  # (nearly everything is constrained — since round 5 the PI row too)
    t.free()
    w.free()


def test_linear_hasher_matches_oracle(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    for n, cap in ((0, 4), (7, 20), (20, 20)):
        q = synthetic.random_log_queries(max(n, 1), seed=n + 1)[:n]
        qs = oracle.linear_hasher_queue_state(q)  # the queue's own state: the closed-form section ties the pops to its head AND its tail
        t = native.Trace(ctx, N_ROWS, 1, n_cols=native.LH_COLS)
        rec, pi = ctx.synthesize_linear_hasher(q, qs, cap, t, 0)
        exp, orec, opi = oracle.linear_hasher_synthesize(q, qs, cap, N_ROWS)
        assert np.array_equal(t.get(0), exp) and rec.tobytes() == orec.tobytes() and np.array_equal(pi, opi)
        assert rec["keccak256_hash"][0].tobytes() == oracle.linear_keccak256(q)
        assert ctx.check_if_satisfied_linear_hasher(t, 0, cap) == (0, (0, 0, 0))
        t.free()


def test_linear_hasher_batch_equals_single_calls(ctx, oracle):
    """zkw_linear_hasher_synthesize_batch: ragged queues (incl. an empty one and a full one) in one call == the oracle per queue"""
    from era_zkevm_test_harness_amd import native

    cap, sizes = 20, (7, 0, 20, 1, 13)
    queues = [synthetic.random_log_queries(max(n, 1), seed=40 + k)[:n] for k, n in enumerate(sizes)]
    states = np.concatenate([oracle.linear_hasher_queue_state(q) for q in queues])
    t = native.Trace(ctx, N_ROWS, len(sizes) + 1, n_cols=native.LH_COLS)
    rec, pi = ctx.synthesize_linear_hasher_batch(queues, states, cap, t, 1)
    for k, q in enumerate(queues):
        exp, orec, opi = oracle.linear_hasher_synthesize(q, states[k:k + 1], cap, N_ROWS)
        assert np.array_equal(t.get(1 + k), exp), k
        assert rec[k:k + 1].tobytes() == orec.tobytes() and np.array_equal(pi[k], opi)
        assert ctx.check_if_satisfied_linear_hasher(t, 1 + k, cap) == (0, (0, 0, 0))
    with pytest.raises(native.ZkwError):
        ctx.synthesize_linear_hasher_batch(queues, states, cap, t, 2)  # one slot short
    # the queues' states handed in (zkw_linear_hasher_synthesize_batch_with_tails, what zkw_block_synthesize does): the same traces;
    # a non-empty queue head is where the pops of the queue section start; wrong states are a broken chain, which the checker reports
    states[2:3] = oracle.linear_hasher_queue_state(queues[2], [9, 8, 7, 6])
    tails = [oracle.queue_push_chain_log(oracle.encode_log_queries(q), states["head"][k])[1] for k, q in enumerate(queues)]
    ctx.synthesize_linear_hasher_batch(queues, states, cap, t, 1, tails=tails)
    for k, q in enumerate(queues):
        exp, _orec, _opi = oracle.linear_hasher_synthesize(q, states[k:k + 1], cap, N_ROWS)
        assert np.array_equal(t.get(1 + k), exp), k
    bad = [x.copy() for x in tails]
    bad[0][3, 1] += 1
    ctx.synthesize_linear_hasher_batch(queues, states, cap, t, 1, tails=bad)
    assert ctx.check_if_satisfied_linear_hasher(t, 1, cap)[0] > 0 and ctx.check_if_satisfied_linear_hasher(t, 2, cap)[0] == 0
    t.free()


def test_dummy_instances_of_empty_queues(ctx, oracle):
    """no request at all: one instance of idle cycles, satisfied, equal to the oracle's"""
    from era_zkevm_test_harness_amd import native

    for ct in (5, 6):
        kind, cols, synth, check, osynth, ocheck, cap = CIRCUITS[ct]
        w, o = _precompile(ctx, oracle, kind, 0, cap)
        assert w.num_instances == 1
        t = native.Trace(ctx, N_ROWS, 1, n_cols=getattr(native, cols))
        getattr(ctx, synth)(w, t, 0, 1, 0)
        assert np.array_equal(t.get(0), getattr(oracle, osynth)(o, 0, cap, N_ROWS)) and getattr(ctx, check)(t, 0, cap)[0] == 0
        t.free()
        w.free()


@pytest.mark.parametrize("ct,n_req,max_rounds", [(5, 700, 2), (6, 3000, 3)])
def test_production_geometry(ctx, oracle, ct, n_req, max_rounds):
    """2^20 rows at the reference's capacity (293 / 2206 cycles): two full instances synthesized in one call and checked on the GPU;
    the multiplicity column counts every lookup slot once"""
    from era_zkevm_test_harness_amd import native

    kind, cols, synth, check, osynth, ocheck, _ = CIRCUITS[ct]
    cap = int(native.circuit_geometry(ct)["capacity"])
    w, _o = _precompile(ctx, oracle, kind, n_req, cap, seed=11, max_rounds=max_rounds)
    assert w.num_instances >= 2
    n_rows = 1 << 20
    lay = native.circuit_layout(ct)
    assert int(lay["fits"]) == 1 and int(lay["rows_used"]) <= n_rows
    t = native.Trace(ctx, n_rows, 2, n_cols=getattr(native, cols))
    getattr(ctx, synth)(w, t, 0, 2, 0)
    g = oracle.nl_geometry(ct)
    for i in range(2):
        assert getattr(ctx, check)(t, i, cap) == (0, (0, 0, 0))
        mult = t.get(i, getattr(native, cols) - 1, 1)[0]
        assert int(mult.sum()) == cap * oracle.nl_slots_per_cycle(ct) and not mult[g["table_rows"]:].any()
    t.free()
    w.free()


def test_code_decommitter_over_a_block(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    b = synthetic.block_after_vm(seed=2)
    cap = 7
    dec = ctx.compute_decommitts_sorter_circuit_snapshots(b["decommit_queries"], 5)
    dq, dt = dec.get(native.DEC_DEDUP_QUERIES), dec.get(native.DEC_DEDUP_TAILS)
    codes = [b["bytecodes"][h.tobytes()] for h in dq["hash"]]
    woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    w = ctx.compute_decommitter_circuit_snapshots(dq, dt, np.concatenate(codes), woff, cap, mem_in)
    o = oracle.decommitter_build(dq, dt, np.concatenate(codes), woff, cap, mem_in)
    ni = w.num_instances
    t = native.Trace(ctx, N_ROWS, ni, n_cols=native.DC_COLS)
    ctx.synthesize_code_decommitter(w, t, 0, ni, 0)
    for i in range(ni):
        assert np.array_equal(t.get(i), oracle.code_decommitter_synthesize(o, i, cap, N_ROWS))
        assert ctx.check_if_satisfied_code_decommitter(t, i, cap) == (0, (0, 0, 0))
    t.free()
    w.free()
    dec.free()


@pytest.mark.parametrize("ct", [3, 13])
def test_queue_section_tamper_parity_decommitter_and_linear_hasher(ctx, oracle, ct):
    """the two checkers agree (violation count and smallest code) on tampered cells of the queue section of types 3 and 13 — the
    full-width pop, the recomposition gates and cycle-dependent links of the L1 messages, the relations between operations — and on a
    tampered hashed cell of the netlist, which the links must notice"""
    import ctypes

    from era_zkevm_test_harness_amd import native

    rng = np.random.default_rng(100 + ct)
    if ct == 3:
        from oracle import block as ob

        b = synthetic.block_after_vm(seed=2)
        cap = cycles = 7
        o = ob.create_artifacts_after_vm(b, {ob.CODE_DECOMMITTER: cap})["witnesses"]["code_decommitter"]
        base = oracle.code_decommitter_synthesize(o, 1, cap, N_ROWS)
        check, ocheck, cols = ctx.check_if_satisfied_code_decommitter, oracle.code_decommitter_check, native.DC_COLS
        ocap = cap
    else:
        cap = 20
        cycles = oracle.linear_hasher_cycles(cap)
        q = synthetic.mixed_log_queue(60, seed=8)[:13]
        base, _rec, _pi = oracle.linear_hasher_synthesize(q, oracle.linear_hasher_queue_state(q), cap, N_ROWS)
        check, ocheck, cols = ctx.check_if_satisfied_linear_hasher, oracle.linear_hasher_check, native.LH_COLS
        ocap = cycles
    assert ocheck(base, ocap) == (0, (0, 0, 0))
    qg = oracle.nlq_geometry(ct, cycles)
    G = oracle.nl_geometry(ct)["general"]
    rpc = oracle.nl_geometry(ct)["rows_per_cycle"]
    qc = lambda *a, **k: oracle.nlq_cell(ct, cycles, *a, **k)  # noqa: E731
    cells = [qc(2, 0), qc(2, 1), qc(1, 0, -1, 0, 3), qc(1, 0, -1, 0, 10), qc(1, 1, -1, 0, 30), qc(2, 0, -1, 1, 2), qc(2, 0, -1, 2, 1), qc(2, 0, -1, 3, 0),
             qc(2, 0, 0, 0, 5), qc(2, 0, 0, 0, 77), qc(3, 1, 0, 0, 129), qc(0, qg["ops"], k=1), qc(0, qg["ops"], k=G - 1), (G - 1, qc(2, 0)[1]), (G + 1, qc(2, 1)[1])]
    if ct == 13:
        cells += [qc(1, 0, -1, 0, 80), qc(1, 1, -1, 0, 108), qc(1, 0, -1, 0, 18), qc(1, 0, 2, 0, 40)]  # a written_value byte, a tx byte, tx_number, the third permutation
    cells += [(int(rng.integers(0, G)), int(rng.integers(qg["first_row"], qg["rows_used"]))) for _ in range(20)]
    cells += [(int(rng.integers(0, cols - 1)), int(rng.integers(rpc, 3 * rpc))) for _ in range(8)]  # the netlist of cycles 1-2: hashed cells among them
    from nlcf_cells import closed_form_cells
    cells += closed_form_cells(oracle, ct, cycles, rng)
    t = native.Trace(ctx, N_ROWS, 1, n_cols=cols)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n_flagged = 0
    for col, row in cells:
        bad = base.copy()
        bad[col, row] += 1
        ctx.synchronize()
        assert hip.hipMemcpy(t.device_ptr(0), bad.ctypes.data, bad.nbytes, 1) == 0
        got, want = check(t, 0, cap), ocheck(bad, ocap)
        assert got == want, ((col, row), got, want)
        n_flagged += got[0] > 0
    assert n_flagged >= len(cells) - 8
    t.free()


def test_production_geometry_decommitter_and_linear_hasher(ctx, oracle):
    """2^20 rows at the reference capacities (2845 SHA-256 rounds / 774 messages) with the queue sections: synthesized and checked on the GPU"""
    from era_zkevm_test_harness_amd import native

    n_rows = 1 << 20
    for ct in (3, 13):
        lay = native.circuit_layout(ct)
        assert int(lay["fits"]) == 1 and int(lay["queue_rows_per_cycle"]) > 0 and int(lay["rows_used"]) <= n_rows
    # CodeDecommitter: the bytecodes of a synthetic block at capacity 2845
    b = synthetic.block_after_vm(seed=2)
    cap = int(native.circuit_geometry(3)["capacity"])
    dec = ctx.compute_decommitts_sorter_circuit_snapshots(b["decommit_queries"], 5)
    dq, dt = dec.get(native.DEC_DEDUP_QUERIES), dec.get(native.DEC_DEDUP_TAILS)
    codes = [b["bytecodes"][h.tobytes()] for h in dq["hash"]]
    woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
    w = ctx.compute_decommitter_circuit_snapshots(dq, dt, np.concatenate(codes), woff, cap, np.zeros(1, native.QUEUE_STATE12))
    t = native.Trace(ctx, n_rows, 1, n_cols=native.DC_COLS)
    ctx.synthesize_code_decommitter(w, t, 0, 1, 0)
    assert ctx.check_if_satisfied_code_decommitter(t, 0, cap) == (0, (0, 0, 0))
    t.free(); w.free(); dec.free()
    # L1MessagesHasher: a full queue of 774 messages, the queue's states handed in
    cap = int(native.circuit_geometry(13)["capacity"])
    q = synthetic.mixed_log_queue(4 * cap + 100, seed=5)[:cap]
    assert q.size == cap
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(q))[1]
    t = native.Trace(ctx, n_rows, 1, n_cols=native.LH_COLS)
    rec, _pi = ctx.synthesize_linear_hasher_batch([q], oracle.linear_hasher_queue_state(q), cap, t, 0, tails=[tails])
    assert rec["keccak256_hash"][0].tobytes() == oracle.linear_keccak256(q)
    assert ctx.check_if_satisfied_linear_hasher(t, 0, cap) == (0, (0, 0, 0))
    lay = native.circuit_layout(13)
    qb = t.get(0, 0, 8)[:, int(lay["queue_first_row"])]  # QBND: the head before (zeros) | after = the state after the last push
    assert qb[:4].tolist() == [0, 0, 0, 0] and qb[4:8].tolist() == np.asarray(tails[-1]).tolist()
    t.free()


@pytest.mark.parametrize("ct", [5, 6])
def test_slot_reuse_keeps_no_stale_cells(ctx, oracle, ct):
    """a slot whose layout tag matches is NOT cleared (zkw_precompiles.hip, nl_synthesize_with) — the path the block ring
    and bench.py take. Every instance of a queue is synthesized INTO THE SAME SLOT one after the other without the slot's pointer
    being taken in between (taking it resets the tag), a full instance followed by the partial last one and then by instance 0 again,
    from two different request queues: each time the slot must equal the oracle's trace of that instance cell for cell."""
    from era_zkevm_test_harness_amd import native

    kind, cols, synth, check, osynth, ocheck, cap = CIRCUITS[ct]
    cols = getattr(native, cols)
    t = native.Trace(ctx, N_ROWS, 1, n_cols=cols)
    wa, oa = _precompile(ctx, oracle, kind, 9, cap, seed=3)
    wb, ob = _precompile(ctx, oracle, kind, 10, cap, seed=21, max_rounds=3)
    assert wa.num_instances >= 3 and wb.num_instances >= 2
    order = [(wa, oa, 0), (wa, oa, wa.num_instances - 1), (wb, ob, wb.num_instances - 1), (wb, ob, 0), (wa, oa, 1), (wa, oa, wa.num_instances - 1)]
    for w, o, i in order:
        getattr(ctx, synth)(w, t, i, 1, 0)  # instance i into slot 0; t.get() copies through the library (no zkw_trace_device_ptr)
        exp = getattr(oracle, osynth)(o, i, cap, N_ROWS)
        got = t.get(0)
        assert np.array_equal(got, exp), (i, np.argwhere(got != exp)[:4])
        assert getattr(ctx, check)(t, 0, cap) == (0, (0, 0, 0))
    t.free()
    wa.free()
    wb.free()


def test_type_dispatching_entry_points(ctx, oracle):
    """zkw_synthesize / zkw_check_satisfied (the reference's `match` in ZkSyncBaseLayerCircuit::synthesis, base_layer/mod.rs:286-323):
    the same cells and verdicts as the per-type functions for the netlist types incl. the L1MessagesHasher's record-of-pointers
    witness; MainVM and unknown types are refused"""
    import ctypes as C

    from era_zkevm_test_harness_amd import native

    for ct in (5, 6):
        kind, cols, synth, check, osynth, ocheck, cap = CIRCUITS[ct]
        w, o = _precompile(ctx, oracle, kind, 6, cap)
        t = native.Trace(ctx, N_ROWS, 2, n_cols=getattr(native, cols))
        ctx.synthesize(ct, w, t, 1, 2, 0)
        for k in range(2):
            assert np.array_equal(t.get(k), getattr(oracle, osynth)(o, 1 + k, cap, N_ROWS))
            assert ctx.check_if_satisfied(ct, t, k, cap) == (0, (0, 0, 0))
        with pytest.raises(native.ZkwError):
            ctx.synthesize(1, w, t, 0, 1, 0)
        with pytest.raises(native.ZkwError):
            ctx.synthesize(14, w, t, 0, 1, 0)
        t.free()
        w.free()
    # type 13: instance k = queue k of a zkw_linear_hasher_witness
    cap, sizes = 20, (7, 0, 13)
    queues = [synthetic.random_log_queries(max(n, 1), seed=60 + k)[:n] for k, n in enumerate(sizes)]
    flat = np.concatenate(queues)
    off = np.array([0, 7, 7, 20], np.uint64)
    states = np.concatenate([oracle.linear_hasher_queue_state(q) for q in queues])

    class LHW(C.Structure):
        _fields_ = [("messages", C.c_void_p), ("message_offsets", C.c_void_p), ("n_queues", C.c_size_t), ("queue_states", C.c_void_p),
                    ("message_tails", C.c_void_p), ("capacity", C.c_uint32), ("records_out", C.c_void_p), ("public_inputs_out", C.c_void_p)]

    pis = np.zeros((3, 4), np.uint64)
    lw = LHW(flat.ctypes.data, off.ctypes.data, 3, states.ctypes.data, None, cap, None, pis.ctypes.data)
    t = native.Trace(ctx, N_ROWS, 2, n_cols=native.LH_COLS)
    ctx.synthesize(13, C.addressof(lw), t, 1, 2, 0)  # queues 1 and 2 -> slots 0 and 1
    for k in range(2):
        exp, _rec, opi = oracle.linear_hasher_synthesize(queues[1 + k], states[1 + k:2 + k], cap, N_ROWS)
        assert np.array_equal(t.get(k), exp) and np.array_equal(pis[1 + k], opi)
        assert ctx.check_if_satisfied(13, t, k, cap) == (0, (0, 0, 0))
    with pytest.raises(native.ZkwError):
        ctx.synthesize(13, C.addressof(lw), t, 2, 2, 0)  # queue 3 of 3
    t.free()


def test_forged_fsm_words_get_the_oracles_verdict(ctx, oracle):
    """the closed-form section's gated ties (docs/KERNELS.md 3.22): an FSM word forged in the instance record with the whole section recomputed
    (the oracle's fill) is ONE violation of kind 7 on the oracle — the GPU checker must say the same about the same trace (types 6, 3, 5, 10)"""
    import ctypes

    from era_zkevm_test_harness_amd import native
    from oracle import block as ob
    from sap_case import storage_application_case

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def verdicts(cols, check, ocheck, trace, cap, ocap=None):
        t = native.Trace(ctx, N_ROWS, 1, n_cols=cols)
        ctx.synchronize()
        assert hip.hipMemcpy(t.device_ptr(0), trace.ctypes.data, trace.nbytes, 1) == 0
        got, want = check(t, 0, cap), ocheck(trace, cap if ocap is None else ocap)
        t.free()
        return got, want

    def forge(w, i, path, flip=False):
        inst = w["instances"].copy()
        rec = inst
        for p in path[:-1]:
            rec = rec[p]
        rec[path[-1]][i] = rec[path[-1]][i] ^ 1 if flip else rec[path[-1]][i] + 1
        f = dict(w)
        f["instances"] = inst
        return f

    for kind, ct, cap, cols, synth, check, ocheck, fields in (
            (1, 6, 3, native.SC_COLS, oracle.sha256_round_synthesize, ctx.check_if_satisfied_sha256_round_function, oracle.sha256_round_check,
             [("hidden_fsm_input", "input_offset"), ("hidden_fsm_output", "num_rounds"), ("hidden_fsm_input", "output_page")]),
            (0, 5, 2, native.KC_COLS, oracle.keccak_round_synthesize, ctx.check_if_satisfied_keccak_round_function, oracle.keccak_round_check,
             [("hidden_fsm_input", "output_offset"), ("hidden_fsm_output", "output_page")])):
        req, mq = synthetic.precompile_trace(kind, 9, seed=3, max_rounds=4)
        tails = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
        w = oracle.precompile_build(kind, req, tails, mq, cap, np.zeros(1, native.QUEUE_STATE12))
        n = w["instances"].size
        cont = next(i for i in range(1, n - 1) if int(synth(w, i, cap, N_ROWS)[oracle.nlq_cell(ct, cap, 0, 0)]) == 0)
        for path in fields:
            got, want = verdicts(cols, check, ocheck, synth(forge(w, cont, path), cont, cap, N_ROWS), cap)
            assert want[0] == 1 and want[1][0] == 7 and got == want, (ct, path, got, want)
        got, want = verdicts(cols, check, ocheck, synth(forge(w, cont, ("hidden_fsm_output", "log_queue_state", "tail"), flip=False), cont, cap, N_ROWS), cap)
        assert want[0] >= 1 and got == want, (ct, "far end", got, want)
    a = ob.create_artifacts_after_vm(synthetic.block_after_vm(seed=2), {ob.CODE_DECOMMITTER: 2})
    w = a["witnesses"]["code_decommitter"]
    n = w["instances"].size
    cont = next(i for i in range(1, n - 1) if int(oracle.code_decommitter_synthesize(w, i, 2, N_ROWS)[oracle.nlq_cell(3, 2, 0, 0)]) == 0)
    for path in (("hidden_fsm_input", "current_index"), ("hidden_fsm_output", "timestamp")):
        got, want = verdicts(native.DC_COLS, ctx.check_if_satisfied_code_decommitter, oracle.code_decommitter_check,
                             oracle.code_decommitter_synthesize(forge(w, cont, path), cont, 2, N_ROWS), 2)
        assert want[0] == 1 and want[1][0] == 7 and got == want, (3, path, got, want)
    sq, stails, tree, _idx, _paths = storage_application_case(oracle, 7, seed=9)
    sap = oracle.storage_application_build(tree, sq, stails, 3)
    inst = sap["instances"].copy()
    inst["hidden_fsm_output"]["current_root_hash"][1][5] ^= 1
    f = dict(sap)
    f["instances"] = inst
    got, want = verdicts(native.SA_COLS, ctx.check_if_satisfied_storage_application, oracle.storage_application_check,
                         oracle.storage_application_synthesize(f, sq, 1, 3, N_ROWS), 3)
    assert want[0] == 2 and want[1][0] == 7 and got == want, (10, got, want)
