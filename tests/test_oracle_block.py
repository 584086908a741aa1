"""CPU: the oracle's block sequencing (oracle/block.py, reference order) — the checker of zkw_block_run — is internally
consistent: shared queues thread through the builders, every synthesized instance satisfies its circuit."""
import numpy as np

from era_zkevm_test_harness_amd import synthetic


def test_oracle_block_threads_queues_and_satisfies_circuits(oracle):
    from oracle import block as ob

    b = synthetic.block_after_vm(seed=1)
    caps = {ob.DECOMMITS_SORTER: 5, ob.CODE_DECOMMITTER: 7, ob.LOG_DEMUXER: 64, ob.KECCAK256: 3, ob.SHA256: 4, ob.ECRECOVER: 2,
            ob.RAM_PERMUTATION: 1000, ob.STORAGE_SORTER: 40, ob.STORAGE_APPLICATION: 5, ob.EVENTS_SORTER: 16, ob.L1_MESSAGES_SORTER: 9}
    timings = {}
    a = ob.create_artifacts_after_vm(b, caps, timings=timings)
    w = a["witnesses"]
    total = a["memory_queries"].size
    assert total == b["vm_memory_queries"].size + w["code_decommitter"]["mem_q"].size + sum(m.size for m in b["precompile_memory_queries"])
    ram0 = w["ram_permutation"]["instances"][0]
    assert int(ram0["unsorted_queue_initial_state"]["length"]) == total == int(a["memory_queue_state"]["length"][0])
    assert np.array_equal(ram0["unsorted_queue_initial_state"]["tail"], a["memory_queue_state"]["tail"][0])
    assert {"ram_permutation", "decommit_sorter", "log_demuxer", "storage_sorter"} <= set(timings)
    checks = {ob.LOG_DEMUXER: oracle.log_demux_check, ob.RAM_PERMUTATION: oracle.ram_check, ob.DECOMMITS_SORTER: oracle.decommit_sorter_check,
              ob.STORAGE_SORTER: oracle.storage_sorter_check, ob.EVENTS_SORTER: oracle.events_sorter_check,
              ob.L1_MESSAGES_SORTER: oracle.events_sorter_check}
    seen = []

    def on_trace(ctype, i, t):
        bad, first = checks[ctype](t, caps[ctype])
        assert bad == 0, (ctype, i, first)
        seen.append(ctype)

    n = ob.synthesize_all(a, 1 << 15, on_trace)
    assert n == len(seen) > 12 and seen == sorted(seen, key=ob.EMISSION_ORDER.index)
    for ctype, (enc, states) in a["recursion_queues"].items():
        assert np.array_equal(enc[:, 1:5], a["public_inputs"][ctype]) and (enc[:, 0] == ctype).all()
