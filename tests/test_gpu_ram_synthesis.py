"""GPU synthesis of RAMPermutation instances through the C ABI: the emitted trace (every column, every
row, multiplicities included) is bit-exact against the oracle's; the GPU satisfiability check accepts it
and rejects tampered traces; production geometry (2^20 rows, capacity 136 714) via the GPU checker."""
import ctypes as C

import numpy as np

from era_zkevm_test_harness_amd.ram_circuit import boundary_row
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


def _trace(n, seed, pages=3, indices=16, heap_writes=2):
    q = synthetic.ram_trace(n, seed=seed, pages=pages, indices=indices)
    k = min(n, heap_writes)
    q["page"][:k] = 10
    q["index"][:k] = 1000 + np.arange(k)
    q["timestamp"][:k] = 0
    q["rw_flag"][:k] = 1
    q["value_is_pointer"][:k] = 0
    mem = {}
    for rec in q:
        key = (int(rec["page"]), int(rec["index"]))
        if rec["rw_flag"]:
            mem[key] = (rec["value"].copy(), rec["value_is_pointer"])
        elif key in mem:
            rec["value"], rec["value_is_pointer"] = mem[key]
        else:
            rec["value"], rec["value_is_pointer"] = 0, 0
    return q


@pytest.mark.parametrize("n,capacity,n_rows", [(1, 8, 512), (100, 128, 1024), (256, 128, 1024), (700, 256, 2048),
                                               (5000, 2048, 1 << 14)])
def test_trace_matches_oracle(ctx, oracle, n, capacity, n_rows):
    from era_zkevm_test_harness_amd import native

    q = _trace(n, seed=n)
    w = ctx.compute_ram_circuit_snapshots(q, capacity, 2)
    o = oracle.ram_build_instances(q, capacity, 2)
    k = w.num_instances
    t = native.Trace(ctx, n_rows, k)
    ctx.synthesize_ram(w, t)
    for idx in range(k):
        got = t.get(idx)
        exp = oracle.ram_synthesize(o, idx, capacity, n_rows)
        if not np.array_equal(got, exp):
            cols, rows = np.nonzero(got != exp)
            raise AssertionError(f"instance {idx}: {cols.size} cells differ, first at col {cols[0]} row {rows[0]}: "
                                 f"{got[cols[0], rows[0]]} vs {exp[cols[0], rows[0]]}")
        assert oracle.ram_check(got, capacity)[0] == 0          # oracle checker accepts the GPU trace
        bad, first = ctx.check_if_satisfied_ram(t, idx, capacity)  # GPU checker accepts it too
        assert bad == 0, first
    t.free()
    w.free()


def test_gpu_checker_rejects_tampering(ctx, oracle):
    import torch

    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 128, 1024
    q = _trace(200, seed=3)
    w = ctx.compute_ram_circuit_snapshots(q, capacity, 2)
    t = native.Trace(ctx, n_rows, 1)
    ctx.synthesize_ram(w, t, 0, 1)
    assert ctx.check_if_satisfied_ram(t, 0, capacity)[0] == 0
    host = t.get(0)
    rng = np.random.default_rng(2)
    used = np.argwhere(host[:148, :boundary_row(capacity) + 40] != 0)
    kinds = set()
    base = native.load().zkw_trace_device_ptr(t.handle, 0)
    for _ in range(25):
        c, r = used[rng.integers(len(used))]
        addr = base + (int(c) * n_rows + int(r)) * 8
        old = np.array([host[c, r]], np.uint64)
        new = np.array([(int(host[c, r]) + 1) % P], np.uint64)
        torch.cuda.synchronize()
        _hip_copy(addr, new)
        bad, first = ctx.check_if_satisfied_ram(t, 0, capacity)
        obad, ofirst = oracle.ram_check(_with(host, c, r, new[0]), capacity)
        assert bad > 0 and obad > 0, (c, r)
        assert bad == obad, (c, r, bad, obad)  # both checkers count the same violated relations
        kinds.add(first[0])
        _hip_copy(addr, old)
    assert ctx.check_if_satisfied_ram(t, 0, capacity)[0] == 0
    assert len(kinds) >= 2
    # the closed-form section: challenges, start-flag selection, commitments, public input (what the reference derives in-circuit)
    from closed_form_case import ram_closed_form_tampers
    for name, c, r in ram_closed_form_tampers(capacity):
        addr = base + (int(c) * n_rows + int(r)) * 8
        new = np.array([(int(host[c, r]) + 1) % P], np.uint64)
        _hip_copy(addr, new)
        bad, first = ctx.check_if_satisfied_ram(t, 0, capacity)
        obad, ofirst = oracle.ram_check(_with(host, c, r, new[0]), capacity)
        assert bad > 0 and bad == obad and first[0] == ofirst[0], (name, bad, first, obad, ofirst)
        _hip_copy(addr, np.array([host[c, r]], np.uint64))
    assert ctx.check_if_satisfied_ram(t, 0, capacity)[0] == 0
    # multiplicity and padding
    for (c, r) in ((148, 3), (7, n_rows - 1)):
        addr = base + (c * n_rows + r) * 8
        _hip_copy(addr, np.array([host[c, r] + 1], np.uint64))
        assert ctx.check_if_satisfied_ram(t, 0, capacity)[0] > 0
        _hip_copy(addr, np.array([host[c, r]], np.uint64))
    t.free()
    w.free()


def _with(a, c, r, v):
    b = a.copy()
    b[c, r] = v
    return b


def _hip_copy(dev_addr, host_arr):
    import torch

    src = torch.from_numpy(host_arr.view(np.int64)).cuda()
    hip = C.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(C.c_void_p(dev_addr), C.c_void_p(src.data_ptr()), C.c_size_t(host_arr.nbytes), C.c_int(3))
    assert rc == 0
    torch.cuda.synchronize()


def test_production_geometry(ctx):
    """2^20 rows, capacity 136 714 (BASELINE configs): two instances (one full, one padded) synthesised and
    accepted by the GPU checker; size-independent properties of the emitted columns."""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 136714, 1 << 20
    q = synthetic.ram_trace(capacity + 20000, seed=11)
    w = ctx.compute_ram_circuit_snapshots(q, capacity, 0)
    assert w.num_instances == 2
    t = native.Trace(ctx, n_rows, 2)
    ctx.synthesize_ram(w, t)
    inst = w.get(native.RAM_INSTANCES)
    for idx in range(2):
        bad, first = ctx.check_if_satisfied_ram(t, idx, capacity)
        assert bad == 0, (idx, first)
        mult = t.get(idx, 148, 1)[0]
        assert int(mult.sum()) == 15 * n_rows and not mult[256:].any()
        bout = t.get(idx, 0, 40)[:, boundary_row(capacity) + 1]
        fo = inst[idx]["hidden_fsm_output"]
        assert np.array_equal(bout[0:12], fo["current_unsorted_queue_state"]["head"])
        assert np.array_equal(bout[26:28], fo["lhs_accumulator"]) and np.array_equal(bout[28:30], fo["rhs_accumulator"])
        assert np.array_equal(bout[30:33].astype(np.uint32), fo["previous_sorting_key"])
    t.free()
    w.free()


def test_baseline_config0_ram_2pow16(ctx, oracle):
    """BASELINE.json configs[0] on the GPU: 2^16-row RAMPermutation trace bit-exact vs the CPU-only run."""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 8192, 1 << 16
    q = synthetic.ram_trace(capacity, seed=1)
    w = ctx.compute_ram_circuit_snapshots(q, capacity, 0)
    o = oracle.ram_build_instances(q, capacity, 0)
    t = native.Trace(ctx, n_rows, 1)
    ctx.synthesize_ram(w, t)
    assert np.array_equal(t.get(0), oracle.ram_synthesize(o, 0, capacity, n_rows))
    assert ctx.check_if_satisfied_ram(t, 0, capacity)[0] == 0
    t.free()
    w.free()


def test_chain_window_smaller_than_the_batch(oracle, monkeypatch):
    """the grand-product chains are recomputed per group of blocks (builder) and per synthesis call: with a window of
    one block the batch of 5 blocks goes through 5 groups, and a synthesis call that spans blocks is split"""
    from era_zkevm_test_harness_amd import native

    monkeypatch.setenv("ZKW_Z_WINDOW_ITEMS", "1")
    ctx = native.Context(0)
    capacity, n_rows = 300, 1 << 12
    sizes = [700, 300, 1000, 50, 620]
    qs = [synthetic.ram_trace(n, seed=30 + k) for k, n in enumerate(sizes)]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    w = ctx.compute_ram_circuit_snapshots(np.concatenate(qs), capacity, 0, block_offsets=offs)
    os_ = [oracle.ram_build_instances(q, capacity, 0) for q in qs]
    exp_inst = np.concatenate([o["instances"] for o in os_])
    assert w.get(native.RAM_INSTANCES).tobytes() == exp_inst.tobytes()
    n_inst = exp_inst.size
    t = native.Trace(ctx, n_rows, n_inst)
    ctx.synthesize_ram(w, t)  # one call over all 5 blocks
    k = 0
    for o in os_:
        for i in range(o["instances"].size):
            assert np.array_equal(t.get(k), oracle.ram_synthesize(o, i, capacity, n_rows)), k
            k += 1
    # the ABI arrays are still whole
    lz = w.get(native.RAM_LHS_Z)
    for b, o in enumerate(os_):
        lo, n = int(offs[b]), sizes[b]
        assert np.array_equal(lz.reshape(-1)[2 * lo:2 * (lo + n)].reshape(2, n), o["lhs_z"])
    t.free()
    w.free()
    ctx.close()


def test_witness_outlives_later_builds_in_host_pointer_mode(ctx, oracle):
    """host-pointer mode: the witness owns a device copy of its queries (the kernels re-encode them at synthesis time),
    so a later build on the same context — which reuses the context's staging buffers — does not disturb it"""
    from era_zkevm_test_harness_amd import native

    capacity, n_rows = 400, 1 << 12
    q1, q2 = synthetic.ram_trace(1000, seed=71), synthetic.ram_trace(1000, seed=72)
    w1 = ctx.compute_ram_circuit_snapshots(q1, capacity, 0)
    w2 = ctx.compute_ram_circuit_snapshots(q2, capacity, 0)  # overwrites the staging copy of q1
    o1 = oracle.ram_build_instances(q1, capacity, 0)
    t = native.Trace(ctx, n_rows, w1.num_instances)
    ctx.synthesize_ram(w1, t)
    for i in range(w1.num_instances):
        assert np.array_equal(t.get(i), oracle.ram_synthesize(o1, i, capacity, n_rows)), i
    assert np.array_equal(w1.get(native.RAM_SORTED_QUERIES), o1["sorted_q"])
    assert np.array_equal(w1.get(native.RAM_UNSORTED_ENC), o1["unsorted_enc"])
    assert np.array_equal(w1.get(native.RAM_RHS_Z).reshape(2, -1), o1["rhs_z"])
    t.free(); w1.free(); w2.free()


def test_slot_reuse_keeps_the_padding_rows(ctx, oracle):
    """a slot whose previous tenant had the same layout keeps its zero padding rows (the tail kernel skips them): instance after instance
    into ONE slot without touching the device pointer, a ragged last instance after full ones, then another capacity (the tag no longer
    matches: everything is cleared), then back — every trace equals the oracle's cell for cell"""
    from era_zkevm_test_harness_amd import native

    n_rows = 2048
    q = _trace(700, seed=9)
    t = native.Trace(ctx, n_rows, 1)
    for capacity in (256, 128, 256):
        w = ctx.compute_ram_circuit_snapshots(q, capacity, 2)
        o = oracle.ram_build_instances(q, capacity, 2)
        assert w.num_instances >= 3
        for idx in range(w.num_instances):  # the last one is ragged
            ctx.synthesize_ram(w, t, idx, 1, 0)
            assert np.array_equal(t.get(0), oracle.ram_synthesize(o, idx, capacity, n_rows)), (capacity, idx)
        w.free()
    t.free()
