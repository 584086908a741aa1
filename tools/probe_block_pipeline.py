"""Whole blocks per second with the builders of batch k + 1 running next to the synthesis of batch k (two host threads, one driving
zkw_blocks_run, one zkw_blocks_synthesize): usage probe_block_pipeline.py K [batches]"""
import sys, time, threading
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
base = [synthetic.block_production(seed=1 + k) for k in range(4)]
blocks = [base[k % 4] for k in range(K)]
warm = nv.Block(0, base[0]); warm.synthesize(1 << 20, ring_slots=1); warm.free()
bs = nv.Block.run_many(0, blocks)  # fills the caches
nv.Block.synthesize_many(bs, 1 << 20, ring_slots=1)
for b in bs: b.free()

def synth(bs, out):
    t = time.perf_counter()
    out.append(nv.Block.synthesize_many(bs, 1 << 20, ring_slots=1))  # zkw_blocks_synthesize: ECRecover of all blocks in joint calls, the other types on library threads
    for b in bs: b.free()
    out.append(time.perf_counter() - t)

t0 = time.perf_counter()
prev, th, n_inst, tb, ts = None, None, 0, [], []
for r in range(batches):
    t = time.perf_counter()
    cur = nv.Block.run_many(0, blocks)
    tb.append(time.perf_counter() - t)
    if th is not None:
        th.join(); n_inst += res[0]; ts.append(res[1])
    res = []
    th = threading.Thread(target=synth, args=(cur, res)); th.start()
th.join(); n_inst += res[0]; ts.append(res[1])
dt = time.perf_counter() - t0
print(f"K={K} x {batches} batches pipelined: {K*batches/dt:.1f} blocks/s, {n_inst/dt:.0f} circuits/s; builders per batch {[round(x,2) for x in tb]} s, synthesis {[round(x,2) for x in ts]} s")
