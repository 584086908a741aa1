"""The builders of K production blocks (zkw_blocks_run) twice, nothing else: the run to put under `rocprofv3 --kernel-trace` when the
question is what the batch's one stream does before the memory-queue chain can start. Usage: probe_blocks_builders_trace.py K"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from era_zkevm_test_harness_amd import native as nv, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
base = [nv.Block.queues_to_device(synthetic.block_production(seed=1 + k)) for k in range(4)]
templates = nv.Block.prepare_many(0, [base[k % 4] for k in range(K)])
for r in range(2):
    t = time.perf_counter()
    bs = nv.Block.run_prepared(0, templates)
    print(f"builders of {K} blocks: {(time.perf_counter() - t) * 1e3:.0f} ms", flush=True)
    nv.Block.free_many(bs)
