"""TEST INFRASTRUCTURE — the post-VM half of `create_artifacts_from_tracer` (src/witness/oracle.rs:928-1130) on the CPU
oracle, in the REFERENCE's order (one builder after the other on the caller's thread, the memory queue threaded
VM -> code decommitter -> keccak256 -> sha256 -> ecrecover -> RAM permutation): the checker of `zkw_block_run`
(tests/test_gpu_block.py) and what bench.py's cpu_baseline leg times for the full-block wall time.
Only tests, smoke() and the cpu_baseline leg may import this.
"""
import time

import numpy as np

from . import pyoracle as o

(MAIN_VM, DECOMMITS_SORTER, CODE_DECOMMITTER, LOG_DEMUXER, KECCAK256, SHA256, ECRECOVER, RAM_PERMUTATION, STORAGE_SORTER,
 STORAGE_APPLICATION, EVENTS_SORTER, L1_MESSAGES_SORTER, L1_MESSAGES_HASHER) = range(1, 14)

# circuit_sequencer_api/src/geometry_config.rs:5-20
DEFAULT_CAPACITY = {MAIN_VM: 5585, DECOMMITS_SORTER: 117500, CODE_DECOMMITTER: 2845, LOG_DEMUXER: 58750, KECCAK256: 293,
                    SHA256: 2206, ECRECOVER: 7, RAM_PERMUTATION: 136714, STORAGE_SORTER: 46921, STORAGE_APPLICATION: 33,
                    EVENTS_SORTER: 31287, L1_MESSAGES_SORTER: 31287, L1_MESSAGES_HASHER: 774}


def _state12(tail, length):
    s = np.zeros(1, o.QUEUE_STATE12)
    if tail is not None:
        s["tail"] = tail
    s["length"] = length
    return s


def create_artifacts_after_vm_concurrent(block, capacities=None, threads=8):
    """The same builders as create_artifacts_after_vm on concurrent host threads, the way a CPU implementation with a thread
    pool would run them (the C oracle releases the GIL): decommit sorter || the VM's memory-queue chain || log demuxer first;
    the code decommitter and the precompile builders then extend the memory queue one after the other (its hash chain is
    serial); the RAM permutation, the storage / events / L1 sorters and the hasher run beside them as soon as their inputs
    exist. Used by bench.py's full-block leg so that the GPU's dependency-graph sequencer is compared with a CPU side that
    is also concurrent, not with a sequential one. Returns (artifacts-lite dict, wall seconds)."""
    from concurrent.futures import ThreadPoolExecutor

    cap = dict(DEFAULT_CAPACITY)
    cap.update(capacities or {})
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        f_dec = ex.submit(o.decommit_sorter_build, block["decommit_queries"], cap[DECOMMITS_SORTER])
        vm_mem = np.ascontiguousarray(block["vm_memory_queries"], dtype=o.MEM_QUERY)
        f_vm = ex.submit(lambda: o.queue_push_chain_full(o.encode_memory_queries(vm_mem)))
        f_dmx = ex.submit(o.log_demux_build, block["log_queries"], cap[LOG_DEMUXER])
        dec = f_dec.result()
        codes = [np.ascontiguousarray(block["bytecodes"][h.tobytes()], dtype=np.uint32).reshape(-1, 8) for h in dec["dedup_q"]["hash"]]
        woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
        dmx = f_dmx.result()
        off = dmx["out_offsets"].astype(np.int64)
        queue = lambda k: (dmx["out_q"][off[k]:off[k + 1]], dmx["out_new_tails"][off[k]:off[k + 1]])  # noqa: E731
        f_sto = ex.submit(o.storage_sorter_build, queue(0)[0], cap[STORAGE_SORTER])
        f_evs = ex.submit(o.events_sorter_build, queue(1)[0], cap[EVENTS_SORTER])
        f_l1 = ex.submit(lambda: (lambda l1s: (l1s, o.linear_keccak256(l1s["result_q"])))(o.events_sorter_build(queue(2)[0], cap[L1_MESSAGES_SORTER])))
        vm_tails = f_vm.result()
        mem_state = _state12(vm_tails[-1] if vm_mem.size else None, vm_mem.size)
        dcm = o.decommitter_build(dec["dedup_q"], dec["dedup_tails"], np.concatenate(codes), woff, cap[CODE_DECOMMITTER], mem_state)
        memory = [vm_mem, dcm["mem_q"]] + [np.ascontiguousarray(m, dtype=o.MEM_QUERY) for m in block["precompile_memory_queries"]]
        f_ram = ex.submit(o.ram_build_instances, np.concatenate(memory), cap[RAM_PERMUTATION], 0)  # contents only: no hash of the chain above is needed
        mem_state = _state12(dcm["mem_tails"][-1], int(mem_state["length"][0]) + dcm["mem_q"].size)
        pre = []
        for k, ctype in enumerate((KECCAK256, SHA256, ECRECOVER)):
            req, req_tails = queue(3 + k)
            mq = np.ascontiguousarray(block["precompile_memory_queries"][k], dtype=o.MEM_QUERY)
            w = o.precompile_build(k, req, req_tails, mq, cap[ctype], mem_state)
            pre.append(w)
            if mq.size:
                mem_state = _state12(w["mem_tails"][-1], int(mem_state["length"][0]) + mq.size)
        out = {"decommits_sorter": dec, "code_decommitter": dcm, "log_demuxer": dmx, "precompiles": pre, "ram_permutation": f_ram.result(),
               "storage_sorter": f_sto.result(), "events_sorter": f_evs.result(), "l1": f_l1.result()}
    return out, time.perf_counter() - t0


def create_artifacts_after_vm(block, capacities=None, storage_tree=None, timings=None):
    """Returns a dict of the builders' outputs, public inputs per circuit type (for the types that have an encoder) and
    one recursion queue per type. `storage_tree`: an oracle.Tree holding the pre-block state (mutated), or None.
    `timings`: optional dict, filled with seconds per builder."""
    cap = dict(DEFAULT_CAPACITY)
    cap.update(capacities or {})
    T = timings if timings is not None else {}

    def timed(name, fn, *a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t
        return r

    art, pis = {}, {}
    dec = timed("decommit_sorter", o.decommit_sorter_build, block["decommit_queries"], cap[DECOMMITS_SORTER])
    art["decommits_sorter"] = dec
    pis[DECOMMITS_SORTER] = o.decommit_sorter_public_inputs(dec["instances"])[1]
    # the VM's part of the memory queue (oracle.rs:894-903)
    vm_mem = np.ascontiguousarray(block["vm_memory_queries"], dtype=o.MEM_QUERY)
    vm_tails = timed("vm_memory_queue", lambda: o.queue_push_chain_full(o.encode_memory_queries(vm_mem)))
    mem_state = _state12(vm_tails[-1] if vm_mem.size else None, vm_mem.size)
    memory = [vm_mem]
    codes = [np.ascontiguousarray(block["bytecodes"][h.tobytes()], dtype=np.uint32).reshape(-1, 8) for h in dec["dedup_q"]["hash"]]
    woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
    dcm = timed("code_decommitter", o.decommitter_build, dec["dedup_q"], dec["dedup_tails"], np.concatenate(codes), woff,
                cap[CODE_DECOMMITTER], mem_state)
    art["code_decommitter"] = dcm
    pis[CODE_DECOMMITTER] = o.closed_form_public_inputs(CODE_DECOMMITTER, dcm["instances"])[1]
    memory.append(dcm["mem_q"])
    mem_state = _state12(dcm["mem_tails"][-1], int(mem_state["length"][0]) + dcm["mem_q"].size)
    dmx = timed("log_demuxer", o.log_demux_build, block["log_queries"], cap[LOG_DEMUXER])
    art["log_demuxer"] = dmx
    pis[LOG_DEMUXER] = o.log_demux_public_inputs(dmx["instances"])[1]
    off = dmx["out_offsets"].astype(np.int64)
    queue = lambda k: (dmx["out_q"][off[k]:off[k + 1]], dmx["out_new_tails"][off[k]:off[k + 1]])  # noqa: E731
    for k, (name, ctype) in enumerate((("keccak256", KECCAK256), ("sha256", SHA256), ("ecrecover", ECRECOVER))):
        req, req_tails = queue(3 + k)
        mq = np.ascontiguousarray(block["precompile_memory_queries"][k], dtype=o.MEM_QUERY)
        w = timed(name, o.precompile_build, k, req, req_tails, mq, cap[ctype], mem_state)
        art[name] = w
        pis[ctype] = o.closed_form_public_inputs(ctype, w["instances"])[1]
        if mq.size:
            memory.append(mq)
            mem_state = _state12(w["mem_tails"][-1], int(mem_state["length"][0]) + mq.size)
    all_mem = np.concatenate(memory)
    ram = timed("ram_permutation", o.ram_build_instances, all_mem, cap[RAM_PERMUTATION], 0)
    art["ram_permutation"] = ram
    pis[RAM_PERMUTATION] = o.ram_public_inputs(ram["instances"])[1]
    sto = timed("storage_sorter", o.storage_sorter_build, queue(0)[0], cap[STORAGE_SORTER])
    art["storage_sorter"] = sto
    pis[STORAGE_SORTER] = o.storage_sorter_public_inputs(sto["instances"])[1]
    evs = timed("events_sorter", o.events_sorter_build, queue(1)[0], cap[EVENTS_SORTER])
    art["events_sorter"] = evs
    pis[EVENTS_SORTER] = o.events_sorter_public_inputs(evs["instances"])[1]
    l1s = timed("l1_messages_sorter", o.events_sorter_build, queue(2)[0], cap[L1_MESSAGES_SORTER])
    art["l1_messages_sorter"] = l1s
    pis[L1_MESSAGES_SORTER] = o.events_sorter_public_inputs(l1s["instances"])[1]
    pubdata_hash = timed("l1_messages_hasher", o.linear_keccak256, l1s["result_q"])
    hasher = np.zeros(1, o.LINEAR_HASHER_INSTANCE)  # data_hasher_and_merklizer.rs:34-60
    hasher["start_flag"] = hasher["completion_flag"] = 1
    hasher["queue_state"] = l1s["instances"]["final_queue_state"][-1]
    hasher["keccak256_hash"] = np.frombuffer(pubdata_hash, np.uint8)
    art["l1_messages_hasher"] = {"instances": hasher, "messages": l1s["result_q"]}
    pis[L1_MESSAGES_HASHER] = o.closed_form_public_inputs(L1_MESSAGES_HASHER, hasher)[1]
    if storage_tree is not None:
        sap = timed("storage_application", o.storage_application_build, storage_tree, sto["result_q"],
                    sto["result_new_tails"], cap[STORAGE_APPLICATION])
        sap["queries"] = sto["result_q"]  # the instances' tree queries (a read = one Merkle walk, a write = two)
        art["storage_application"] = sap
        pis[STORAGE_APPLICATION] = o.closed_form_public_inputs(STORAGE_APPLICATION, sap["instances"])[1]
    recursion = {t: o.recursion_queue(t, p) for t, p in pis.items()}
    return {"witnesses": art, "memory_queries": all_mem, "memory_queue_state": mem_state, "demuxed_offsets": off,
            "public_inputs": pis, "recursion_queues": recursion, "l1_messages_pubdata_hash": pubdata_hash, "capacities": cap}


def _linear_hasher_synthesize(w, i, capacity, n_rows):
    return o.linear_hasher_synthesize(w["messages"], w["instances"]["queue_state"][0], capacity, n_rows)[0]


def _storage_application_synthesize(w, i, capacity, n_rows):
    return o.storage_application_synthesize(w, w["queries"], i, capacity, n_rows)


SYNTH = {STORAGE_APPLICATION: ("storage_application", _storage_application_synthesize), LOG_DEMUXER: ("log_demuxer", o.log_demux_synthesize), RAM_PERMUTATION: ("ram_permutation", o.ram_synthesize),
         DECOMMITS_SORTER: ("decommits_sorter", o.decommit_sorter_synthesize), STORAGE_SORTER: ("storage_sorter", o.storage_sorter_synthesize),
         EVENTS_SORTER: ("events_sorter", o.events_sorter_synthesize), L1_MESSAGES_SORTER: ("l1_messages_sorter", o.events_sorter_synthesize),
         CODE_DECOMMITTER: ("code_decommitter", o.code_decommitter_synthesize), KECCAK256: ("keccak256", o.keccak_round_synthesize), SHA256: ("sha256", o.sha256_round_synthesize), ECRECOVER: ("ecrecover", o.ecrecover_synthesize), L1_MESSAGES_HASHER: ("l1_messages_hasher", _linear_hasher_synthesize)}
# oracle.rs:975-984 demuxer, :1039-1049 RAM, :1115-1130 storage application (when the block has a storage tree), then CircuitMaker
# order :1494-1732 (the types that have a synthesis here)
EMISSION_ORDER = (LOG_DEMUXER, RAM_PERMUTATION, STORAGE_APPLICATION, DECOMMITS_SORTER, CODE_DECOMMITTER, KECCAK256, SHA256, ECRECOVER, STORAGE_SORTER, EVENTS_SORTER, L1_MESSAGES_SORTER,
                  L1_MESSAGES_HASHER)


def synthesize_all(artifacts, n_rows, on_trace=None):
    """ZkSyncBaseLayerCircuit::synthesis of every instance of the twelve synthesized types in emission order; returns the
    number of instances. on_trace(circuit_type, instance, trace) is called with each filled trace."""
    done = 0
    for ctype in EMISSION_ORDER:
        key, fn = SYNTH[ctype]
        if key not in artifacts["witnesses"]:
            continue  # (no storage tree: no storage application)
        w = artifacts["witnesses"][key]
        for i in range(w["instances"].size):
            t = fn(w, i, artifacts["capacities"][ctype], n_rows)
            if on_trace is not None:
                on_trace(ctype, i, t)
            done += 1
    return done
