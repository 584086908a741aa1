"""Whole blocks per second through zkw_blocks_run + zkw_blocks_synthesize, batch after batch or overlapped (the builders of batch k + 1 —
one host thread, a wave per SIMD at most — under the synthesis of batch k; the release of batch k - 1 on a third thread).
Usage: probe_blocks_pipeline.py K batches [seq|overlap] [host|device]   (ZKW_BATCH_LOG=1 / ZKW_BLOCK_MEM_LOG=1 for the library's own lines)"""
import sys, time, threading
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from era_zkevm_test_harness_amd import native as nv, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "seq"
where = sys.argv[4] if len(sys.argv) > 4 else "device"
base = [synthetic.block_production(seed=1 + k) for k in range(4)]
if where == "device":
    base = [nv.Block.queues_to_device(b) for b in base]
templates = nv.Block.prepare_many(0, [base[k % 4] for k in range(K)])
warm = nv.Block(0, base[0]); warm.synthesize(1 << 20, ring_slots=1); warm.free()


def build():
    t = time.perf_counter()
    bs = nv.Block.run_prepared(0, templates)
    return bs, time.perf_counter() - t


def synth(bs):
    t = time.perf_counter()
    n = nv.Block.synthesize_many(bs, 1 << 20, ring_slots=1, ec_chunk=int(os.environ.get("EC_CHUNK", "0")))
    return n, time.perf_counter() - t


def free(bs):
    t = time.perf_counter()
    nv.Block.free_many(bs)
    return time.perf_counter() - t


bs, tb = build(); n, ts = synth(bs); tf = free(bs)   # warm-up batch (fills the allocation caches)
print(f"warm-up: builders {tb*1e3:.0f} ms, synthesis {ts*1e3:.0f} ms, release {tf*1e3:.0f} ms", flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
total = 0
if mode in ("seq", "seqbg"):
    bg = None
    for r in range(batches):
        bs, tb = build()
        if bg is not None: bg.join()
        n, ts = synth(bs); total += n
        if mode == "seqbg":   # the release of a batch under the builders of the next one (host work both)
            tf = 0.0
            bg = threading.Thread(target=free, args=(bs,)); bg.start()
        else:
            tf = free(bs)
        fr, tot = torch.cuda.mem_get_info(0)
        print(f"K={K} batch {r}: builders {tb*1e3:.0f} ms, synthesis of {n} instances {ts*1e3:.0f} ms, release {tf*1e3:.0f} ms; HBM in use {(tot-fr)/2**30:.0f} GiB", flush=True)
    if bg is not None: bg.join()
else:
    res = {}
    def t_build(): res["b"] = build()
    def t_synth(x): res["s"] = synth(x)
    def t_free(x): res["f"] = free(x)
    cur, _ = build()
    old = None
    for r in range(batches):
        th = [threading.Thread(target=t_synth, args=(cur,))]
        if r + 1 < batches: th.append(threading.Thread(target=t_build))
        if old is not None: th.append(threading.Thread(target=t_free, args=(old,)))
        for t in th: t.start()
        for t in th: t.join()
        n, ts = res["s"]; total += n
        tb = res["b"][1] if r + 1 < batches else 0.0
        fr, tot = torch.cuda.mem_get_info(0)
        print(f"K={K} step {r}: synthesis of {n} instances {ts*1e3:.0f} ms | builders of the next batch {tb*1e3:.0f} ms | release {res.get('f', 0)*1e3:.0f} ms; HBM in use {(tot-fr)/2**30:.0f} GiB", flush=True)
        old = cur
        cur = res["b"][0] if r + 1 < batches else None
    free(old)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"K={K} x {batches} batches, {mode}, queues on {where}: {K*batches/wall:.1f} blocks/s, {total/wall:.0f} synthesized circuits/s, wall {wall*1e3:.0f} ms")
