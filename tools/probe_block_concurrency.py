"""K production-capacity blocks in flight at once through zkw_blocks_run (the blocks' builders as fibers of one thread, their launches
merged per kernel and stage: csrc/zkw_batch.h): throughput of whole blocks. Usage: probe_block_concurrency.py K [reps]"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
base = [synthetic.block_production(seed=1 + k) for k in range(min(K, 4))]
blocks = [base[k % len(base)] for k in range(K)]
warm = nv.Block(0, base[0]); warm.synthesize(1 << 20, ring_slots=1); warm.free()
for r in range(reps):
    t0 = time.perf_counter()
    bs = nv.Block.run_many(0, blocks)
    t1 = time.perf_counter()
    import torch
    free_b, total_b = torch.cuda.mem_get_info(0)
    n = nv.Block.synthesize_many(bs, 1 << 20, ring_slots=1)
    t2 = time.perf_counter()
    spans = {}
    for name, s, e in bs[0].timings():
        spans[name] = round(e - s, 1)
    for b in bs: b.free()
    t3 = time.perf_counter()
    print(f"K={K:3d} round {r}: builders {1e3*(t1-t0):.0f} ms, synthesis of {n} instances {1e3*(t2-t1):.0f} ms, free {1e3*(t3-t2):.0f} ms -> "
          f"{K/(t2-t0):.2f} blocks/s, {n/(t2-t0):.1f} synthesized circuits/s; block 0 spans: ram {spans.get('ram_permutation')}, dec {spans.get('decommit_sorter.finish')}, dmx {spans.get('log_demuxer')}; HBM in use after the builders {(total_b - free_b) / 2**30:.1f} GiB")
