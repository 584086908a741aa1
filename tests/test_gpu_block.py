"""GPU, end to end: the post-VM half of create_artifacts_from_tracer (src/witness/oracle.rs:928-1130) over one synthetic
block — every witness builder in the reference's order with the shared queues threaded through them, every instance of
the synthesized circuit types filled and checked, one recursion queue per circuit type."""
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
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
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
    # every instance of the six queue circuit types (block.synthesize_and_check): filled, satisfied, PI row = the instance's public input
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
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
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


def _tree_for(oracle, dedup_queries, seed):
    """a storage tree that holds, before the block, exactly what the block's first reads of every slot expect"""
    tree = oracle.Tree()
    rng = np.random.default_rng(seed)
    for _ in range(10):
        tree.insert_leaf(rng.bytes(32), rng.bytes(32))
    for q in dedup_queries:
        if q["read_value"].any():
            tree.insert_leaf(oracle.derive_final_address(q), b"".join(int(x).to_bytes(4, "big") for x in q["read_value"][::-1]))
    return tree


@pytest.mark.parametrize("seed", [1, 4])
def test_block_sequencer_matches_builder_by_builder(ctx, oracle, seed):
    """zkw_block_run (the C++ sequencer: builders as a dependency graph on their own streams / threads, the memory queue
    hashed once) against the same builders called one by one in the reference's order (block.create_artifacts_after_vm):
    every instance record, public input, recursion queue and shared queue state identical; then every instance
    synthesized in emission order through the circuit callback and checked."""
    from era_zkevm_test_harness_amd import block as blk, native as nv

    b = synthetic.block_after_vm(seed=seed)
    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.STORAGE_APPLICATION: 5, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
    a = blk.create_artifacts_after_vm(ctx, b, caps)
    w = a["witnesses"]
    dedup = w["storage_sorter"].get(nv.STO_RESULT_QUERIES)
    tree = _tree_for(oracle, dedup, seed)
    root0, next0 = tree.root, tree.next_enumeration_index

    def tree_answers(q):
        assert q.tobytes() == dedup.tobytes()
        idx = np.zeros(q.size, np.uint64)
        paths = np.zeros((q.size, 256, 32), np.uint8)
        for i in range(q.size):
            idx[i], _, paths[i] = tree.get_leaf(oracle.derive_final_address(q[i]))
        return idx, paths

    B = nv.Block(0, b, caps, storage_tree=tree_answers, storage_initial_root=root0, storage_next_enumeration_index=next0)
    # shared queues
    assert B.memory_queue_length == a["memory_queries"].size
    assert B.memory_queue_state().tobytes() == a["memory_queue_state"].tobytes()
    assert np.array_equal(B.demuxed_offsets().astype(np.int64), a["demuxed_offsets"])
    assert B.l1_messages_hash() == a["l1_messages_pubdata_hash"]
    # instance records of every builder
    pairs = ((blk.DECOMMITS_SORTER, nv.DEC_INSTANCES, "decommits_sorter"), (blk.CODE_DECOMMITTER, nv.DCM_INSTANCES, "code_decommitter"),
             (blk.LOG_DEMUXER, nv.DMX_INSTANCES, "log_demuxer"), (blk.KECCAK256, nv.PRC_INSTANCES, "keccak256"),
             (blk.SHA256, nv.PRC_INSTANCES, "sha256"), (blk.ECRECOVER, nv.PRC_INSTANCES, "ecrecover"),
             (blk.RAM_PERMUTATION, nv.RAM_INSTANCES, "ram_permutation"), (blk.STORAGE_SORTER, nv.STO_INSTANCES, "storage_sorter"),
             (blk.EVENTS_SORTER, nv.EVT_INSTANCES, "events_sorter"), (blk.L1_MESSAGES_SORTER, nv.EVT_INSTANCES, "l1_messages_sorter"))
    for ctype, what, key in pairs:
        got, exp = B.witness_get(ctype, what, np.uint8), w[key].get(what)
        assert B.num_instances(ctype) == w[key].num_instances, key
        assert got.tobytes() == exp.tobytes(), key
    # the storage application over the callback's answers == the builder called directly == the oracle
    qt = w["storage_sorter"].get(nv.STO_RESULT_NEW_TAILS)
    idx, paths = tree_answers(dedup)
    sap = ctx.decompose_into_storage_application_witnesses(dedup, qt, idx, paths, root0, next0, caps[blk.STORAGE_APPLICATION])
    assert B.witness_get(blk.STORAGE_APPLICATION, nv.SAP_INSTANCES, np.uint8).tobytes() == sap.get(nv.SAP_INSTANCES).tobytes()
    o = oracle.storage_application_build(tree, dedup, qt, caps[blk.STORAGE_APPLICATION])
    assert B.witness_get(blk.STORAGE_APPLICATION, nv.SAP_ROOTS, np.uint8).tobytes() == o["roots"].tobytes()
    assert B.num_instances(blk.STORAGE_APPLICATION) == o["instances"].size >= 2
    # closed forms of every non-VM type: the builder-by-builder path, the oracle on the same records, the sequencer
    assert B.linear_hasher_instance().tobytes() == a["linear_hasher_instance"].tobytes()
    a["public_inputs"][blk.STORAGE_APPLICATION] = ctx.closed_form_public_inputs(blk.STORAGE_APPLICATION, sap.get(nv.SAP_INSTANCES))[1]
    a["recursion_queues"][blk.STORAGE_APPLICATION] = ctx.recursion_queue_push(blk.STORAGE_APPLICATION, a["public_inputs"][blk.STORAGE_APPLICATION])
    assert set(a["public_inputs"]) == set(range(2, 14))
    for ctype, what, key in pairs:
        if ctype in nv.CLOSED_FORM_RECORD:
            assert np.array_equal(oracle.closed_form_public_inputs(ctype, w[key].get(what))[1], a["public_inputs"][ctype]), key
    assert np.array_equal(oracle.closed_form_public_inputs(blk.STORAGE_APPLICATION, o["instances"])[1], a["public_inputs"][blk.STORAGE_APPLICATION])
    assert np.array_equal(oracle.closed_form_public_inputs(13, a["linear_hasher_instance"])[1], a["public_inputs"][13])
    sap.free()
    # public inputs and recursion queues
    for ctype, pi in a["public_inputs"].items():
        assert np.array_equal(B.public_inputs(ctype), pi), ctype
        enc, states = B.recursion_queue(ctype)
        assert np.array_equal(enc, a["recursion_queues"][ctype][0]) and np.array_equal(states, a["recursion_queues"][ctype][1])
    # synthesis in emission order through the circuit callback
    seen = []

    def on_circuit(ctype, inst, trace, slot, pi):
        bad, first = B.check_satisfied(ctype, trace, slot)
        assert bad == 0, (ctype, inst, bad, first)
        assert pi == [int(x) for x in a["public_inputs"][ctype][inst]]
        seen.append((ctype, inst))

    n = B.synthesize(1 << 18, ring_slots=3, callback=on_circuit)
    order = [blk.LOG_DEMUXER, blk.RAM_PERMUTATION, blk.STORAGE_APPLICATION, blk.DECOMMITS_SORTER, blk.CODE_DECOMMITTER, blk.KECCAK256, blk.SHA256, blk.ECRECOVER, blk.STORAGE_SORTER, blk.EVENTS_SORTER, blk.L1_MESSAGES_SORTER,
             blk.L1_MESSAGES_HASHER]
    assert seen == [(t, i) for t in order for i in range(B.num_instances(t))] and n == len(seen) > 12
    spans = {name for name, _, _ in B.timings()}
    assert {"builders", "ram_permutation", "decommit_sorter.finish", "log_demuxer", "storage_application", "synthesis"} <= spans
    B.free()
    for wit in w.values():
        wit.free()


def test_block_sequencer_reports_builder_failures(ctx):
    """a failure on a worker thread (here: a decommit request whose bytecode is missing) surfaces as an error, not a crash"""
    from era_zkevm_test_harness_amd import native as nv

    b = synthetic.block_after_vm(seed=2)
    b["bytecodes"].pop(next(iter(b["bytecodes"])))
    with pytest.raises(nv.ZkwError) as ei:
        nv.Block(0, b, {2: 5, 3: 7, 4: 64, 8: 1000})
    assert ei.value.code == nv.ERR_INVALID and "bytecode" in str(ei.value)


def test_block_sharded_synthesis_and_gather(ctx):
    """the multi-GPU path on one GPU: the LPT plan splits the block's instances over `world` ranks (each rank's share
    synthesized here in turn, disjoint and complete), and the C-ABI gather (world 1: no transport) returns the records
    [type, instance, compact form, public input] in emission order."""
    from era_zkevm_test_harness_amd import block as blk, native as nv

    b = synthetic.block_after_vm(seed=5)
    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
    B = nv.Block(0, b, caps)
    order = [blk.LOG_DEMUXER, blk.RAM_PERMUTATION, blk.DECOMMITS_SORTER, blk.CODE_DECOMMITTER, blk.KECCAK256, blk.SHA256, blk.ECRECOVER, blk.STORAGE_SORTER, blk.EVENTS_SORTER, blk.L1_MESSAGES_SORTER,
             blk.L1_MESSAGES_HASHER]
    full = [(t, i) for t in order for i in range(B.num_instances(t))]
    owner = nv.shard_lpt([t for t, _ in full], 3)
    got = []
    for rank in range(3):
        seen = []
        n = B.synthesize(1 << 18, ring_slots=2, callback=lambda t, i, tr, s, pi: seen.append((t, i)), rank=rank, world=3)
        assert n == len(seen) and seen == [x for x, o in zip(full, owner) if o == rank]
        got += seen
    assert sorted(got) == sorted(full) and len(set(got)) == len(full)
    comm = nv.Comm(ctx, 0, 1)
    rec = B.gather_closed_form_inputs(comm)
    assert rec.shape == (len(full), 24)
    assert [(int(r[0]), int(r[1])) for r in rec] == full
    for r in rec:
        t, i = int(r[0]), int(r[1])
        assert np.array_equal(r[20:], B.public_inputs(t)[i])
    what = {blk.RAM_PERMUTATION: nv.RAM_COMPACT_FORMS, blk.LOG_DEMUXER: nv.DMX_COMPACT_FORMS}
    for t, w in what.items():
        cf = B.witness_get(t, w).reshape(-1, 18)
        for r in rec[rec[:, 0] == t]:
            assert np.array_equal(r[2:20], cf[int(r[1])])
    comm.destroy()
    B.free()


def test_block_with_main_vm_slicing(ctx, oracle):
    """zkw_block_run given the tracer's cycle-stamped vectors also returns the MainVM instance records: the entry states come
    from the queues the block itself hashed (VM prefix of the memory queue, unsorted decommit queue); compared with the
    oracle's slicing over the builder-by-builder path's states"""
    from era_zkevm_test_harness_amd import block as blk, native as nv

    b = synthetic.block_after_vm(seed=7)
    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
    n_vm, n_dec = b["vm_memory_queries"].size, b["decommit_queries"].size
    t = synthetic.vm_tracer_streams(n_cycles=2000, cycles_per_snapshot=250, seed=11, n_memory=n_vm, sparse=60)
    rng = np.random.default_rng(12)
    t["decommit_state_cycles"] = np.sort(rng.integers(0, 2000, n_dec)).astype(np.uint32)
    B = nv.Block(0, b, caps, vm_tracer=t)
    got = B.vm_instances()
    assert got is not None and got.size == t["snapshot_cycles"].size - 1 == B.num_instances(blk.MAIN_VM)
    a = blk.create_artifacts_after_vm(ctx, b, caps)
    t_ref = dict(t)
    t_ref["vm_memory_queries"] = b["vm_memory_queries"]
    t_ref["memory_queue_tails"] = a["witnesses"]["ram_permutation"].get(nv.RAM_UNSORTED_TAILS)[:n_vm]
    t_ref["decommit_queue_tails"] = a["witnesses"]["decommits_sorter"].get(nv.DEC_UNSORTED_TAILS)
    exp, _, _ = oracle.vm_slice_instances(t_ref)
    assert got.tobytes() == exp.tobytes()
    assert int(got[-1]["memory_queue_final_state"]["length"]) == n_vm and int(got[-1]["decommitment_queue_final_state"]["length"]) == n_dec
    B.free()
    for wit in a["witnesses"].values():
        wit.free()


def test_blocks_run_many_at_once(ctx, oracle):
    """zkw_blocks_run: several different blocks in flight together, every context on the chain service (their queue chains
    travel in shared launches): each block's records equal what zkw_block_run gives for it alone"""
    from era_zkevm_test_harness_amd import block as blk, native as nv

    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
    bs = [synthetic.block_after_vm(seed=20 + k, n_vm_memory=1500 + 400 * k, n_storage=60 + 30 * k) for k in range(5)]
    many = nv.Block.run_many(0, bs, caps)
    whats = ((blk.RAM_PERMUTATION, nv.RAM_INSTANCES), (blk.DECOMMITS_SORTER, nv.DEC_INSTANCES), (blk.LOG_DEMUXER, nv.DMX_INSTANCES),
             (blk.STORAGE_SORTER, nv.STO_INSTANCES), (blk.EVENTS_SORTER, nv.EVT_INSTANCES), (blk.CODE_DECOMMITTER, nv.DCM_INSTANCES),
             (blk.SHA256, nv.PRC_INSTANCES))
    for b, m in zip(bs, many):
        one = nv.Block(0, b, caps)
        assert m.memory_queue_state().tobytes() == one.memory_queue_state().tobytes()
        for t, w in whats:
            assert m.witness_get(t, w, np.uint8).tobytes() == one.witness_get(t, w, np.uint8).tobytes(), t
        for t in (2, 4, 8, 9, 11, 12):
            assert np.array_equal(m.public_inputs(t), one.public_inputs(t))
            assert np.array_equal(m.recursion_queue(t)[1], one.recursion_queue(t)[1])
        bad = []
        n = m.synthesize(1 << 18, ring_slots=2, callback=lambda t, i, tr, s, pi: bad.append(m.check_satisfied(t, tr, s)[0]))
        assert n == len(bad) > 10 and not any(bad)
        one.free()
    # the oracle agrees with one of them end to end (public inputs of the RAM permutation)
    a = blk.create_artifacts_after_vm(ctx, bs[2], caps)
    assert np.array_equal(many[2].public_inputs(blk.RAM_PERMUTATION), a["public_inputs"][blk.RAM_PERMUTATION])
    for wit in a["witnesses"].values():
        wit.free()
    for m in many:
        m.free()


def test_blocks_run_512_in_flight_heterogeneous(ctx):
    """zkw_blocks_run with 512 blocks in flight (fibers of one host thread, launches merged per kernel and stage: csrc/zkw_batch.h), the
    blocks of EIGHT different shapes interleaved — among them blocks with no events, no L1 messages, no storage accesses, no precompile
    calls (builders that take their early exits while the other blocks' equal stages go on: ADVICE r5, the chain service's stage keys) —
    half of the shapes with their queues resident on the device (zkw_block_inputs.queues_on_device): every block's records equal what
    zkw_block_run gives for its shape alone, and every instance zkw_blocks_synthesize hands out satisfies its circuit."""
    import threading

    from era_zkevm_test_harness_amd import block as blk, native as nv

    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
    shapes = [synthetic.block_after_vm(seed=60, n_vm_memory=900, n_storage=50),
              synthetic.block_after_vm(seed=61, n_vm_memory=1400, n_storage=70, n_events=0),
              synthetic.block_after_vm(seed=62, n_vm_memory=700, n_storage=40, n_l1_messages=0),
              synthetic.block_after_vm(seed=63, n_vm_memory=1100, n_storage=0, n_storage_cells=1),
              synthetic.block_after_vm(seed=64, n_vm_memory=800, n_storage=30, n_precompile_calls=(0, 0, 0)),
              synthetic.block_after_vm(seed=65, n_vm_memory=1000, n_storage=20, n_events=0, n_l1_messages=0),
              synthetic.block_after_vm(seed=66, n_vm_memory=1300, n_storage=90, n_decommits=40, n_bytecodes=9),
              synthetic.block_after_vm(seed=67, n_vm_memory=600, n_storage=10, n_events=3, n_l1_messages=1, n_precompile_calls=(1, 0, 2))]
    whats = ((blk.RAM_PERMUTATION, nv.RAM_INSTANCES), (blk.DECOMMITS_SORTER, nv.DEC_INSTANCES), (blk.LOG_DEMUXER, nv.DMX_INSTANCES),
             (blk.STORAGE_SORTER, nv.STO_INSTANCES), (blk.EVENTS_SORTER, nv.EVT_INSTANCES), (blk.L1_MESSAGES_SORTER, nv.EVT_INSTANCES),
             (blk.CODE_DECOMMITTER, nv.DCM_INSTANCES), (blk.KECCAK256, nv.PRC_INSTANCES), (blk.SHA256, nv.PRC_INSTANCES), (blk.ECRECOVER, nv.PRC_INSTANCES))

    def record(b):
        r = {"mem": b.memory_queue_state().tobytes()}
        for t, w in whats:
            a = b.witness_get(t, w, np.uint8)
            r[(t, w)] = None if a is None else a.tobytes()
        for t in range(2, 14):
            pi = b.public_inputs(t)
            r[("pi", t)] = None if pi is None else pi.tobytes()
            rq = b.recursion_queue(t)
            r[("rq", t)] = None if rq is None or rq[1] is None else rq[1].tobytes()
        return r

    ref = []
    for sh in shapes:
        one = nv.Block(0, sh, caps)
        ref.append(record(one))
        one.free()
    inputs = [nv.Block.queues_to_device(sh) if k % 2 else sh for k, sh in enumerate(shapes)]
    K = 512
    templates = nv.Block.prepare_many(0, [inputs[k % len(inputs)] for k in range(K)], caps)
    many = nv.Block.run_prepared(0, templates)
    assert len(many) == K
    for k, m in enumerate(many):
        got, exp = record(m), ref[k % len(shapes)]
        assert got.keys() == exp.keys()
        for key in exp:
            assert got[key] == exp[key], (k, key)
    # every instance of the first 64 blocks (eight of each shape) through zkw_blocks_synthesize, checked as it is handed out
    bad, lock, local, checkers = [], threading.Lock(), threading.local(), []

    def cb(bi, t, i, tr, s, pi):
        if not hasattr(local, "ctx"):  # a checker context per calling thread: the blocks' own contexts are busy (include/zkw.h, zkw_blocks_synthesize)
            local.ctx = nv.Context(0)
            with lock:
                checkers.append(local.ctx)
        v = many[bi].check_satisfied(t, tr, s, ctx=local.ctx)[0]
        with lock:
            bad.append((bi, t, i, v))

    n = nv.Block.synthesize_many(many[:64], 1 << 18, ring_slots=1, callback=cb)
    assert n == len(bad) and n >= 64 * 9 and not any(v for *_x, v in bad), [x for x in bad if x[3]][:5]
    for c in checkers:
        c.close()
    nv.Block.free_many(many)


def test_blocks_synthesize_many_equals_block_by_block(ctx):
    """zkw_blocks_synthesize (the ECRecover instances of all blocks in joint calls, the other types block by block on the library's
    threads): every trace it hands out is the trace zkw_block_synthesize hands out for the same (block, type, instance) — compared by a
    digest of all cells — and satisfies its circuit; a chunk smaller than the total splits the joint calls"""
    import hashlib
    import threading

    from era_zkevm_test_harness_amd import block as blk, native as nv

    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
    bs = [synthetic.block_after_vm(seed=40 + k, n_vm_memory=1200 + 300 * k, n_storage=50 + 20 * k) for k in range(4)]
    many = nv.Block.run_many(0, bs, caps)
    n_rows = 1 << 18
    lib = nv.load()

    def digest(trace, slot, ctype):
        # the circuit's own columns (a ring is as wide as the widest type: the columns beyond a type's width belong to nobody)
        cols, rows = int(nv.circuit_layout(ctype)["num_columns"]), lib.zkw_trace_num_rows(trace)
        a = np.zeros((cols, rows), np.uint64)
        nv._check(lib.zkw_trace_get(trace, slot, 0, cols, a.ctypes.data))
        return hashlib.sha256(a.tobytes()).hexdigest()

    ref = {}
    for bi, m in enumerate(many):
        m.synthesize(n_rows, ring_slots=2, callback=lambda t, i, tr, s, pi, bi=bi: ref.__setitem__((bi, t, i), (digest(tr, s, t), tuple(pi))))
    got, bad, lock, local, checkers = {}, [], threading.Lock(), threading.local(), []

    def cb(bi, t, i, tr, s, pi):
        if not hasattr(local, "ctx"):  # a checker context per calling thread: the blocks' own contexts are busy (include/zkw.h, zkw_blocks_synthesize)
            local.ctx = nv.Context(0)
            with lock:
                checkers.append(local.ctx)
        d, v = digest(tr, s, t), many[bi].check_satisfied(t, tr, s, ctx=local.ctx)[0]
        with lock:
            got[(bi, t, i)] = (d, tuple(pi))
            bad.append(v)
    for chunk in (32, 3):  # 3 < the blocks' ECRecover instances together: several joint calls
        got.clear()
        n = nv.Block.synthesize_many(many, n_rows, ring_slots=2, ec_chunk=chunk, callback=cb)
        assert n == len(ref) == len(got), (chunk, n, len(ref), len(got))
        assert got == ref, [k for k in ref if got.get(k) != ref[k]][:5]
    assert not any(bad)
    assert sum(1 for k in ref if k[1] == 7) >= 4  # ECRecover instances of several blocks went through the joint calls
    for m in many:
        m.free()


def test_blocks_run_sharded_and_gather_two_ranks_over_tcp(ctx):
    """zkw_blocks_run_sharded + zkw_blocks_gather_closed_form_inputs with world = 2 on one GPU: two host threads are the two
    ranks (the socket transport of zkw_comm between them), each builds only its round-robin share of five blocks, and the root
    receives every block's records in block order — equal to what each block gives alone."""
    import socket
    import threading

    from era_zkevm_test_harness_amd import block as blk, native as nv

    caps = {blk.DECOMMITS_SORTER: 5, blk.CODE_DECOMMITTER: 7, blk.LOG_DEMUXER: 64, blk.KECCAK256: 3, blk.SHA256: 4, blk.ECRECOVER: 2,
            blk.RAM_PERMUTATION: 1000, blk.STORAGE_SORTER: 40, blk.EVENTS_SORTER: 16, blk.L1_MESSAGES_SORTER: 9, blk.L1_MESSAGES_HASHER: 48}
    bs = [synthetic.block_after_vm(seed=50 + k, n_vm_memory=1200 + 500 * k, n_storage=40 + 25 * k) for k in range(5)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    res, err, failed = {}, [], {}

    def rank_main(rank):
        try:
            c = nv.Context(0)
            comm = nv.Comm.tcp(c, "127.0.0.1", port, rank, 2)
            mine = nv.Block.run_sharded(0, bs, rank, 2, caps)
            assert [m is not None for m in mine] == [k % 2 == rank for k in range(5)]
            res[rank] = nv.Block.gather_sharded(mine, comm, rank, 2, root=1, max_per_block=256)
            # a failure only ONE rank can see (rank 0 hands over a block it did not build) must stay collective: rank 0 still
            # takes part with an error record, both it and the root come back with an error instead of the root waiting forever
            broken = [None if (rank == 0 and k == 2) else m for k, m in enumerate(mine)]
            try:
                nv.Block.gather_sharded(broken, comm, rank, 2, root=1, max_per_block=256)
                failed[rank] = None
            except nv.ZkwError as e:
                failed[rank] = str(e)
            again = nv.Block.gather_sharded(mine, comm, rank, 2, root=1, max_per_block=256)  # the communicator is still in step
            assert (again is None) if rank == 0 else all(np.array_equal(a, b) for a, b in zip(again, res[rank]))
            comm.destroy()
            for m in mine:
                if m is not None:
                    m.free()
            c.close()
        except Exception as e:  # noqa: BLE001 — reported by the main thread
            err.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not err, err
    assert res[0] is None and len(res[1]) == 5
    assert failed[0] and "NULL" in failed[0] and failed[1] and "block 2" in failed[1], failed
    comm1 = nv.Comm(ctx, 0, 1)
    for b, got in zip(bs, res[1]):
        one = nv.Block(0, b, caps)
        assert np.array_equal(got, one.gather_closed_form_inputs(comm1))
        one.free()
    comm1.destroy()
