"""Sha256RoundFunction synthesis at production geometry (2^20 rows, capacity 2206): instances per second"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv, synthetic
from oracle import pyoracle as o
o.build()
ctx = nv.Context(0)
cap, n_rows, slots = 2206, 1 << 20, 8
req, mq = synthetic.precompile_trace(1, 6000, seed=5, max_rounds=6)
tails = o.queue_push_chain_log(o.encode_log_queries(req))[1]
w = ctx._precompile(1, req, tails, mq, cap, np.zeros(1, nv.QUEUE_STATE12))
print(f"builder: {w.num_rounds} rounds, {w.num_instances} instances")
t = nv.Trace(ctx, n_rows, slots, n_cols=nv.SC_COLS)
ni = min(w.num_instances, slots)
ctx.profile_enable(True)
for rep in range(3):
    ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
    ctx.synthesize_sha256_round_function(w, t, 0, ni, 0)
    ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"synthesis of {ni} instances: {1e3*dt:.2f} ms = {ni/dt:.0f} circuits/s, {ni*nv.SC_COLS*n_rows*8/dt/1e9:.0f} GB/s of trace", {k: round(v[0], 2) for k, v in ctx.profile().items()})
t0 = time.perf_counter(); bad = ctx.check_if_satisfied_sha256_round_function(t, 0, cap); print("check", bad, f"{1e3*(time.perf_counter()-t0):.1f} ms")
