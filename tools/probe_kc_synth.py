"""Keccak256RoundFunction synthesis at production geometry (2^20 rows, capacity 293): instances per second, bytes per second"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from era_zkevm_test_harness_amd import native as nv, synthetic
from oracle import pyoracle as o
o.build()
ctx = nv.Context(0)
cap, n_rows, slots = 293, 1 << 20, 8
req, mq = synthetic.precompile_trace(0, 1400, seed=5, max_rounds=6)
tails = o.queue_push_chain_log(o.encode_log_queries(req))[1]
mem_in = np.zeros(1, nv.QUEUE_STATE12)
t0 = time.perf_counter()
w = ctx._precompile(0, req, tails, mq, cap, mem_in)
ctx.synchronize()
print(f"builder: {w.num_rounds} rounds, {w.num_instances} instances in {1e3*(time.perf_counter()-t0):.1f} ms")
t = nv.Trace(ctx, n_rows, slots, n_cols=nv.KC_COLS)
ni = min(w.num_instances, slots)
ctx.profile_enable(True)
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    ctx.synthesize_keccak_round_function(w, t, 0, ni, 0)
    ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"synthesis of {ni} instances: {1e3*dt:.2f} ms = {ni/dt:.0f} circuits/s, {ni*nv.KC_COLS*n_rows*8/dt/1e9:.0f} GB/s of trace")
print(ctx.profile() if hasattr(ctx, "profile") else "")
bad = ctx.check_if_satisfied_keccak_round_function(t, ni - 1, cap)
t0 = time.perf_counter(); bad = ctx.check_if_satisfied_keccak_round_function(t, 0, cap); print("check", bad, f"{1e3*(time.perf_counter()-t0):.1f} ms")
