"""Host-side mirror of the post-VM half of `create_artifacts_from_tracer` (src/witness/oracle.rs:928-1130): the order in
which the reference runs its per-circuit witness builders over what the VM left behind, how the shared queues thread
through them, and the recursion queue every circuit type ends with (`CircuitMaker::process` / `into_results`,
src/witness/postprocessing/mod.rs:353-405). Everything heavy is a libzkw call on the GPU; this file is sequencing only.

Not covered: the MainVM instances themselves (they need the VM) and the storage application (it needs the
pre-block Merkle paths of the storage tree from a `BlockchainDataSource`; call
`Context.decompose_into_storage_application_witnesses` on `artifacts["storage_sorter"]` results with them).
"""
import numpy as np

from . import native as nv

# BaseLayerCircuitType, circuit_definitions/src/circuit_definitions/base_layer/mod.rs:55-71
(MAIN_VM, DECOMMITS_SORTER, CODE_DECOMMITTER, LOG_DEMUXER, KECCAK256, SHA256, ECRECOVER, RAM_PERMUTATION, STORAGE_SORTER,
 STORAGE_APPLICATION, EVENTS_SORTER, L1_MESSAGES_SORTER, L1_MESSAGES_HASHER) = range(1, 14)


def _queue_state12(tail, length):
    s = np.zeros(1, nv.QUEUE_STATE12)
    if tail is not None:
        s["tail"] = tail
    s["length"] = length
    return s


def create_artifacts_after_vm(ctx, block, capacities=None):
    """block: the dict of `synthetic.block_after_vm` (or the same arrays from a real VM run). capacities: circuit type ->
    instance capacity (default: geometry_config.rs:5-20 via zkw_circuit_geometry_of). Returns a dict of witnesses (the
    caller frees them), the memory queue as the RAM permutation sees it, and per circuit type the public inputs and the
    recursion queue states."""
    cap = {t: int(nv.circuit_geometry(t)["capacity"]) for t in range(1, 14)}
    cap.update(capacities or {})
    art, pis = {}, {}

    # 1. sort + deduplicate the decommit requests (oracle.rs:928-945)
    dec = ctx.compute_decommitts_sorter_circuit_snapshots(block["decommit_queries"], cap[DECOMMITS_SORTER])
    art["decommits_sorter"] = dec
    pis[DECOMMITS_SORTER] = dec.get(nv.DEC_PUBLIC_INPUTS)
    dedup_q, dedup_tails = dec.get(nv.DEC_DEDUP_QUERIES), dec.get(nv.DEC_DEDUP_TAILS)

    # the VM's part of the memory queue (oracle.rs:894-903)
    vm_mem = np.ascontiguousarray(block["vm_memory_queries"], dtype=nv.MEM_QUERY)
    vm_tails = ctx.queue_push_chain_full(ctx.encode_memory_queries(vm_mem))
    mem_state = _queue_state12(vm_tails[-1] if vm_mem.size else None, vm_mem.size)
    memory = [vm_mem]

    # 2. unpack the bytecodes into memory (oracle.rs:947-961): words in the order of the deduplicated queue
    codes = [block["bytecodes"][h.tobytes()] for h in dedup_q["hash"]]
    woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
    dcm = ctx.compute_decommitter_circuit_snapshots(dedup_q, dedup_tails, np.concatenate(codes), woff, cap[CODE_DECOMMITTER], mem_state)
    art["code_decommitter"] = dcm
    pis[CODE_DECOMMITTER] = ctx.closed_form_public_inputs(CODE_DECOMMITTER, dcm.get(nv.DCM_INSTANCES))[1]
    q = dcm.get(nv.DCM_MEM_QUERIES)
    memory.append(q)
    mem_state = _queue_state12(dcm.get(nv.DCM_MEM_TAILS)[-1], int(mem_state["length"][0]) + q.size)

    # 3. demultiplex the log queue (oracle.rs:963-989)
    dmx = ctx.compute_logs_demux(block["log_queries"], cap[LOG_DEMUXER])
    art["log_demuxer"] = dmx
    pis[LOG_DEMUXER] = dmx.get(nv.DMX_PUBLIC_INPUTS)
    off = dmx.get(nv.DMX_OUT_OFFSETS).astype(np.int64)
    out_q, out_tails = dmx.get(nv.DMX_OUT_QUERIES), dmx.get(nv.DMX_OUT_NEW_TAILS)
    queue = lambda k: (out_q[off[k]:off[k + 1]], out_tails[off[k]:off[k + 1]])  # noqa: E731

    # 4. the three precompiles replay their calls and extend the memory queue (oracle.rs:991-1033)
    for k, (name, ctype, fn) in enumerate((("keccak256", KECCAK256, ctx.keccak256_decompose_into_per_circuit_witness),
                                           ("sha256", SHA256, ctx.sha256_decompose_into_per_circuit_witness),
                                           ("ecrecover", ECRECOVER, ctx.ecrecover_decompose_into_per_circuit_witness))):
        req, req_tails = queue(3 + k)
        mq = np.ascontiguousarray(block["precompile_memory_queries"][k], dtype=nv.MEM_QUERY)
        w = fn(req, req_tails, mq, cap[ctype], mem_state)
        art[name] = w
        pis[ctype] = ctx.closed_form_public_inputs(ctype, w.get(nv.PRC_INSTANCES))[1]
        if mq.size:
            memory.append(mq)
            mem_state = _queue_state12(w.get(nv.PRC_MEM_TAILS)[-1], int(mem_state["length"][0]) + mq.size)

    # 5. the RAM permutation over the whole memory queue (oracle.rs:1035-1051)
    all_mem = np.concatenate(memory)
    ram = ctx.compute_ram_circuit_snapshots(all_mem, cap[RAM_PERMUTATION], 0)
    art["ram_permutation"] = ram
    pis[RAM_PERMUTATION] = ram.get(nv.RAM_PUBLIC_INPUTS)

    # 6. the three log sorters (oracle.rs:1053-1100) and the pubdata hash of the L1 messages (:1102-1112)
    sto = ctx.compute_storage_dedup_and_sort(queue(0)[0], cap[STORAGE_SORTER])
    art["storage_sorter"] = sto
    pis[STORAGE_SORTER] = sto.get(nv.STO_PUBLIC_INPUTS)
    evs = ctx.compute_events_dedup_and_sort(queue(1)[0], cap[EVENTS_SORTER])
    art["events_sorter"] = evs
    pis[EVENTS_SORTER] = evs.get(nv.EVT_PUBLIC_INPUTS)
    l1s = ctx.compute_events_dedup_and_sort(queue(2)[0], cap[L1_MESSAGES_SORTER])
    art["l1_messages_sorter"] = l1s
    pis[L1_MESSAGES_SORTER] = l1s.get(nv.EVT_PUBLIC_INPUTS)
    pubdata_hash = ctx.compute_linear_keccak256(l1s.get(nv.EVT_RESULT_QUERIES))
    hasher = np.zeros(1, nv.LINEAR_HASHER_INSTANCE)  # the LinearHasher instance, data_hasher_and_merklizer.rs:34-60
    hasher["start_flag"] = hasher["completion_flag"] = 1
    hasher["queue_state"] = l1s.get(nv.EVT_INSTANCES)["final_queue_state"][-1]
    hasher["keccak256_hash"] = np.frombuffer(pubdata_hash, np.uint8)
    art_hasher = hasher
    pis[L1_MESSAGES_HASHER] = ctx.closed_form_public_inputs(L1_MESSAGES_HASHER, hasher)[1]

    # 7. one recursion queue per circuit type (postprocessing/mod.rs:393-400)
    recursion = {t: ctx.recursion_queue_push(t, p) for t, p in pis.items()}
    return {"witnesses": art, "memory_queries": all_mem, "memory_queue_state": mem_state, "demuxed_offsets": off,
            "public_inputs": pis, "recursion_queues": recursion, "l1_messages_pubdata_hash": pubdata_hash,
            "linear_hasher_instance": art_hasher, "capacities": cap}


def synthesize_and_check(ctx, artifacts, n_rows):
    """ZkSyncBaseLayerCircuit::synthesis + check_if_satisfied for every instance of the six circuit types libzkw
    synthesizes; returns {circuit type: number of instances}, raises on the first unsatisfied trace."""
    w, cap = artifacts["witnesses"], artifacts["capacities"]
    # through the type-dispatching entry points (zkw_synthesize / zkw_check_satisfied = the reference's enum methods)
    plan = ((RAM_PERMUTATION, w["ram_permutation"], None), (DECOMMITS_SORTER, w["decommits_sorter"], None), (LOG_DEMUXER, w["log_demuxer"], 151),
            (STORAGE_SORTER, w["storage_sorter"], None), (EVENTS_SORTER, w["events_sorter"], None), (L1_MESSAGES_SORTER, w["l1_messages_sorter"], None))
    done = {}
    for ctype, wit, n_cols in plan:
        n_inst = wit.num_instances
        t = nv.Trace(ctx, n_rows, 1, n_cols=n_cols)
        for i in range(n_inst):
            ctx.synthesize(ctype, wit, t, i, 1, 0)
            bad, first = ctx.check_if_satisfied(ctype, t, 0, cap[ctype])
            if bad:
                t.free()
                raise AssertionError(f"circuit type {ctype}, instance {i}: {bad} violations, first {first}")
            assert t.get(0, 0, 4)[:, _pi_row(ctype, cap[ctype])].tolist() == artifacts["public_inputs"][ctype][i].tolist()
        t.free()
        done[ctype] = n_inst
    return done


def _pi_row(ctype, capacity):
    rows_per_cycle, off = {RAM_PERMUTATION: (6, 2), DECOMMITS_SORTER: (7, 3), LOG_DEMUXER: (12, 2), STORAGE_SORTER: (22, 5),
                           EVENTS_SORTER: (13, 5), L1_MESSAGES_SORTER: (13, 5)}[ctype]
    return rows_per_cycle * ((capacity + 63) // 64 * 64) + off
