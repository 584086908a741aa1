"""CPU: the oracle's restatement of the pre-builder half of create_artifacts_from_tracer (oracle/vm_trace.py =
callstack_handler.rs:174-460 + oracle.rs:233-843) on synthetic nested-call traces with reverts: the reference's own asserts
hold (they are restated as asserts), plus the structural facts the MainVM circuit relies on."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic
from oracle import vm_trace


@pytest.mark.parametrize("seed,n,p_panic", [(0, 60, 0.3), (1, 400, 0.3), (2, 400, 0.0), (3, 400, 1.0), (4, 1500, 0.5), (5, 0, 0.0)])
def test_log_queue_and_callstack_replay_invariants(oracle, seed, n, p_panic):
    ev, q, e = synthetic.vm_events(n, seed=seed, p_panic=p_panic)
    a = vm_trace.create_artifacts_before_builders(ev, q, e)
    n_rw = int(q["rw_flag"].sum())
    fq = a["flat_queries"]
    assert fq.size == q.size + n_rw  # every write appears twice: applied and as its rollback twin
    # the chain: each item's old tail is the previous item's new tail, the first starts from the empty queue
    assert np.array_equal(a["flat_old_tails"][1:], a["flat_new_tails"][:-1]) and not a["flat_old_tails"][:1].any()
    n_orig = a["original_log_queue_length"]
    # the original (applied) queue holds every query once plus the rollbacks of panicked frames, in pairs with their forward
    rb = fq["rollback"].astype(bool)
    assert not rb[n_orig:].size or rb[n_orig:].all()  # what follows the applied part is rollbacks only
    applied_rollbacks = int(rb[:n_orig].sum())
    assert n_orig == q.size + applied_rollbacks and fq.size - n_orig == n_rw - applied_rollbacks
    if p_panic == 0.0:
        assert applied_rollbacks == 0
    # every rollback follows its forward twin (same timestamp) in the flat order
    first_seen = {}
    for i, x in enumerate(fq):
        t = int(x["timestamp"])
        if x["rollback"]:
            assert t in first_seen
        else:
            assert t not in first_seen
            first_seen[t] = i
    # one rollback tail per frame, frame 0 and frames without rollbacks start from the global end of the log
    tails = a["rollback_queue_initial_tails_for_new_frames"]
    n_push = int((ev["kind"] == 1).sum())
    assert len(tails) == a["monotonic_frame_counter"] == n_push + 1
    assert np.array_equal(tails[0][1], a["global_end_of_storage_log"]) and tails[0][0] == 0
    assert [c for c, _ in tails] == sorted(c for c, _ in tails)
    # callstack witnesses: one per push and per pop, at distinct cycles, ending on the empty stack
    w = a["callstack_values_witnesses"]
    assert w["is_push"].size == 2 * n_push and int(w["depth"][-1]) == 0 and not w["new_state"][-1].any()
    assert np.all(np.diff(w["cycles"].astype(np.int64)) > 0)
    rc, rs = a["callstack_sponge_encoding_ranges"]
    assert rc.size == 2 * n_push + 1 and rc[0] == 0 and not rs[0].any() and np.array_equal(rs[1:], w["new_state"])
    # the storage-log history ends with everything merged into frame 0: forward = the whole applied queue
    hist = a["history_of_storage_log_states"]
    last = hist[-1][1]
    assert last["frame_idx"] == 0 and last["forward_length"] == n_orig
    if n_orig:
        assert np.array_equal(last["forward_tail"], a["flat_new_tails"][n_orig - 1])
    assert last["rollback_length"] == fq.size - n_orig
    assert np.array_equal(last["rollback_tail"], a["global_end_of_storage_log"])
    if fq.size > n_orig:  # the surviving rollbacks run from the end of the applied queue to the global end
        assert np.array_equal(last["rollback_head"], a["flat_old_tails"][n_orig])
    # head segments: one per write, in cycle order, each the tail BEFORE the write's rollback twin
    assert len(a["rollback_queue_head_segments"]) == n_rw


def test_unbalanced_trace_is_rejected(oracle):
    ev, q, e = synthetic.vm_events(50, seed=9)
    with pytest.raises(AssertionError):
        vm_trace.create_artifacts_before_builders(ev[:-1], q, e)  # the bootloader frame never exits
