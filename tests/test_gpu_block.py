"""GPU, end to end: the post-VM half of create_artifacts_from_tracer (src/witness/oracle.rs:928-1130) over one synthetic
block — every witness builder in the reference's order with the shared queues threaded through them, every instance of
the six synthesized circuit types filled and checked, one recursion queue per circuit type."""
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


@pytest.mark.parametrize("seed", [1, 2])
def test_block_after_vm(ctx, oracle, seed):
    from era_zkevm_test_harness_amd import block as blk, native as nv

    b = synthetic.block_after_vm(seed=seed)
    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9}
    a = blk.create_artifacts_after_vm(ctx, b, caps)
    w = a["witnesses"]
    # the memory queue: VM, then code words, then keccak / sha256 / ecrecover queries; RAM sees exactly that queue
    total = a["memory_queries"].size
    assert total == b["vm_memory_queries"].size + w["code_decommitter"].get(nv.DCM_MEM_QUERIES).size + sum(m.size for m in b["precompile_memory_queries"])
    ram_inst = w["ram_permutation"].get(nv.RAM_INSTANCES)
    assert int(a["memory_queue_state"]["length"][0]) == total == int(ram_inst[0]["unsorted_queue_initial_state"]["length"])
    assert np.array_equal(ram_inst[0]["unsorted_queue_initial_state"]["tail"], a["memory_queue_state"]["tail"][0])
    assert ram_inst.size == -(-total // 1000) > 3
    # the demuxer's six output queues are the input queues of the sorters and the precompiles
    dmx_out = w["log_demuxer"].get(nv.DMX_INSTANCES)[-1]["output_queue_state"]
    sto_inst = w["storage_sorter"].get(nv.STO_INSTANCES)
    evt_inst = w["events_sorter"].get(nv.EVT_INSTANCES)
    l1_inst = w["l1_messages_sorter"].get(nv.EVT_INSTANCES)
    for k, qs in ((0, sto_inst[0]["unsorted_log_queue_state"]), (1, evt_inst[0]["initial_log_queue_state"]), (2, l1_inst[0]["initial_log_queue_state"])):
        assert np.array_equal(qs["tail"], dmx_out[k]["tail"]) and int(qs["length"]) == int(dmx_out[k]["length"]) > 0, k
    off = a["demuxed_offsets"]
    assert [int(off[k + 1] - off[k]) for k in (3, 4, 5)] == [5, 4, 3]
    # the same builders on the CPU oracle agree on every public input (spot check: the types with a commitment row)
    for ctype, key, fn in ((blk.LOG_DEMUXER, "log_demuxer", oracle.log_demux_public_inputs),
                           (blk.STORAGE_SORTER, "storage_sorter", oracle.storage_sorter_public_inputs),
                           (blk.EVENTS_SORTER, "events_sorter", oracle.events_sorter_public_inputs)):
        what = {blk.LOG_DEMUXER: nv.DMX_INSTANCES, blk.STORAGE_SORTER: nv.STO_INSTANCES, blk.EVENTS_SORTER: nv.EVT_INSTANCES}[ctype]
        assert np.array_equal(fn(w[key].get(what))[1], a["public_inputs"][ctype])
    assert a["l1_messages_pubdata_hash"] == oracle.linear_keccak256(w["l1_messages_sorter"].get(nv.EVT_RESULT_QUERIES))
    # every instance of the six synthesized circuit types: filled, satisfied, PI row = the instance's public input
    done = blk.synthesize_and_check(ctx, a, 1 << 15)
    assert done[blk.RAM_PERMUTATION] == ram_inst.size and done[blk.STORAGE_SORTER] == sto_inst.size >= 3
    assert done[blk.LOG_DEMUXER] >= 4 and done[blk.DECOMMITS_SORTER] >= 2 and done[blk.EVENTS_SORTER] >= 2 and done[blk.L1_MESSAGES_SORTER] >= 2
    # one recursion queue per circuit type: encoding [type, pi0..3, 0, 0, 0], chained from the empty queue
    for ctype, (enc, states) in a["recursion_queues"].items():
        pi = a["public_inputs"][ctype]
        assert enc.shape == (pi.shape[0], 8) and (enc[:, 0] == ctype).all() and np.array_equal(enc[:, 1:5], pi) and not enc[:, 5:].any()
        assert np.array_equal(states, oracle.queue_push_chain_full(enc))
    for wit in w.values():
        wit.free()


def test_block_with_empty_queues(ctx, oracle):
    """a block without events, L1 messages, keccak and ecrecover calls: the builders of the empty queues emit their
    dummy instances (events_sort_dedup.rs:27-76), everything else composes as before"""
    from era_zkevm_test_harness_amd import block as blk, native as nv

    b = synthetic.block_after_vm(seed=3, n_events=0, n_l1_messages=0, n_precompile_calls=(0, 3, 0), n_storage=60)
    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9}
    a = blk.create_artifacts_after_vm(ctx, b, caps)
    w = a["witnesses"]
    assert w["events_sorter"].num_instances == 1 and w["l1_messages_sorter"].num_instances == 1
    assert a["l1_messages_pubdata_hash"] == oracle.linear_keccak256(np.zeros(0, oracle.LOG_QUERY))
    ram_inst = w["ram_permutation"].get(nv.RAM_INSTANCES)
    assert np.array_equal(ram_inst[0]["unsorted_queue_initial_state"]["tail"], a["memory_queue_state"]["tail"][0])
    done = blk.synthesize_and_check(ctx, a, 1 << 15)
    assert done[blk.EVENTS_SORTER] == 1 and done[blk.L1_MESSAGES_SORTER] == 1 and done[blk.STORAGE_SORTER] >= 2
    for wit in w.values():
        wit.free()
