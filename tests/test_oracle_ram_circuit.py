"""Oracle synthesis of RAMPermutation instances (zkw trace v1): the filled trace satisfies every
constraint / copy link / lookup of the spec, and tampering is detected (the reference's tests run
`check_if_satisfied` on every emitted circuit, src/tests/mod.rs:130-259)."""
import numpy as np

from era_zkevm_test_harness_amd.ram_circuit import boundary_row, spec_macros
import pytest

from era_zkevm_test_harness_amd import synthetic
from closed_form_case import ram_closed_form_tampers

P = 0xFFFFFFFF00000001


def _build(oracle, n, capacity, seed=5):
    q = synthetic.ram_trace(n, seed=seed, pages=3, indices=16)
    k = min(n, 2)  # bootloader-heap writes at timestamp 0 to cells nothing else touches
    q["page"][:k] = 10
    q["index"][:k] = 1000 + np.arange(k)
    q["timestamp"][:k] = 0
    q["rw_flag"][:k] = 1
    q["value_is_pointer"][:k] = 0
    # the cells those two queries originally wrote may now be read before any write: regenerate validity
    mem = {}
    for rec in q:
        key = (int(rec["page"]), int(rec["index"]))
        if rec["rw_flag"]:
            mem[key] = (rec["value"].copy(), rec["value_is_pointer"])
        elif key in mem:
            rec["value"], rec["value_is_pointer"] = mem[key]
        else:
            rec["value"], rec["value_is_pointer"] = 0, 0
    return oracle.ram_build_instances(q, capacity, 2)


def test_flattened_poseidon_matches_permutation(oracle):
    s = synthetic.random_field_elements(3, (12,))
    slots = oracle.poseidon2_flattened(s)
    assert np.array_equal(slots[:12], s)
    assert np.array_equal(slots[118:], oracle.poseidon2(s))


@pytest.mark.parametrize("n,capacity,n_rows", [(100, 128, 1024), (256, 128, 1024), (300, 128, 1 << 12), (1, 8, 512)])
def test_oracle_trace_is_satisfied(oracle, n, capacity, n_rows):
    o = _build(oracle, n, capacity)
    for idx in range(o["instances"].size):
        t = oracle.ram_synthesize(o, idx, capacity, n_rows)
        bad, first = oracle.ram_check(t, capacity)
        assert bad == 0, (idx, first)
        assert int(t.max()) < P
        # multiplicities sum to the number of lookup cells
        assert int(t[148].sum()) == 15 * n_rows
        # the boundary-out registers are the instance's hidden_fsm_output
        fo = o["instances"][idx]["hidden_fsm_output"]
        bout = boundary_row(capacity) + 1
        assert np.array_equal(t[0:12, bout], fo["current_unsorted_queue_state"]["head"])
        assert np.array_equal(t[24:26, bout].astype(np.uint32),
                              [fo["current_unsorted_queue_state"]["length"], fo["current_sorted_queue_state"]["length"]])
        assert np.array_equal(t[26:28, bout], fo["lhs_accumulator"])
        assert np.array_equal(t[28:30, bout], fo["rhs_accumulator"])
        assert np.array_equal(t[30:33, bout].astype(np.uint32), fo["previous_sorting_key"])
        assert int(t[39, bout]) == int(fo["num_nondeterministic_writes"])


def test_tampering_is_detected(oracle):
    capacity, n_rows = 64, 512
    o = _build(oracle, 100, capacity)
    t = oracle.ram_synthesize(o, 0, capacity, n_rows)
    assert oracle.ram_check(t, capacity)[0] == 0
    rng = np.random.default_rng(1)
    kinds = set()
    used = [(c, r) for c in range(148) for r in range(boundary_row(capacity) + 40) if t[c, r] != 0]
    for _ in range(60):
        c, r = used[rng.integers(len(used))]
        t2 = t.copy()
        t2[c, r] = (int(t2[c, r]) + 1) % P
        bad, first = oracle.ram_check(t2, capacity)
        assert bad > 0, (c, r)
        kinds.add(first[0])
    assert {1, 2} <= kinds  # constraint and poseidon violations both occur
    t2 = t.copy(); t2[39, boundary_row(capacity) + 1] += 1  # BND_OUT.cnt is held by a copy link only
    assert oracle.ram_check(t2, capacity)[1][0] == 4
    # a wrong multiplicity and a dirty padding row
    t2 = t.copy(); t2[148, 0] -= 1
    assert oracle.ram_check(t2, capacity)[0] > 0
    t2 = t.copy(); t2[5, n_rows - 1] = 7
    assert oracle.ram_check(t2, capacity)[0] > 0


def test_closed_form_section(oracle):
    """What the reference's circuit computes in-trace, this trace states: the PI row is the commitment of the compact form of the
    closed-form input (utils.rs:269-306), the challenges of BND_IN come out of the sponge over the observable input's tails and lengths
    (utils.rs:498-550), BND_IN is start ? observable input : hidden FSM input. Instance 0 starts the block, instance 1 continues it."""
    capacity, n_rows = 64, 1024
    o = _build(oracle, 150, capacity)
    M = spec_macros()
    _, pis = oracle.ram_public_inputs(o["instances"])
    for idx in (0, 1, 2):
        t = oracle.ram_synthesize(o, idx, capacity, n_rows)
        assert oracle.ram_check(t, capacity)[0] == 0
        b = boundary_row(capacity)
        assert np.array_equal(t[M["PI_pi0"]:M["PI_pi0"] + 4, b + M["ROWOFF_PI"]], pis[idx])
        assert int(t[M["SEL0_start"], b + M["ROWOFF_SEL0"]]) == (1 if idx == 0 else 0)
        ch = o["challenges"].reshape(2, 9)
        for rep in range(2):
            assert np.array_equal(t[M["CH3_CH3_o0"]:M["CH3_CH3_o0"] + 8, b + M[f"ROWOFF_CH{3 + rep}"]], ch[rep, 1:])
        for name, c, r in ram_closed_form_tampers(capacity):
            t2 = t.copy()
            t2[c, r] = (int(t2[c, r]) + 1) % P
            assert oracle.ram_check(t2, capacity)[0] > 0, (idx, name)


def test_invalid_memory_is_rejected(oracle):
    capacity = 64
    q = synthetic.ram_trace(60, seed=8, pages=2, indices=4)
    reads = np.flatnonzero(q["rw_flag"] == 0)
    q["value"][reads[0], 0] ^= 1  # a read that does not return the last write
    o = oracle.ram_build_instances(q, capacity, 0)
    with pytest.raises(RuntimeError):
        oracle.ram_synthesize(o, 0, capacity, 512)


def test_baseline_config0_ram_2pow16_cpu_only(oracle):
    """BASELINE.json configs[0]: RAMPermutation base circuit, 2^16 rows, CPU-only synthesize. Capacity 8 192
    (SURVEY section 8d), seeded trace; witness + synthesis + satisfiability on the oracle alone."""
    capacity, n_rows = 8192, 1 << 16
    q = synthetic.ram_trace(capacity, seed=1)
    o = oracle.ram_build_instances(q, capacity, 0)
    assert o["instances"].size == 1
    t = oracle.ram_synthesize(o, 0, capacity, n_rows)
    bad, first = oracle.ram_check(t, capacity)
    assert bad == 0, first
    assert t.shape == (149, n_rows) and int(t[148].sum()) == 15 * n_rows
    fo = o["instances"][0]["hidden_fsm_output"]
    assert np.array_equal(fo["lhs_accumulator"], fo["rhs_accumulator"])
