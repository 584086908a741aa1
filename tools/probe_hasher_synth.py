"""LinearHasher / single-instance Keccak256RoundFunction synthesis at production geometry: where the milliseconds go"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv, synthetic
from oracle import pyoracle as o
o.build()
ctx = nv.Context(0)
n_rows = 1 << 20
q = synthetic.mixed_log_queue(4000, seed=3)[:560]
qs = np.zeros(1, nv.QUEUE_STATE4)
t = nv.Trace(ctx, n_rows, 1, n_cols=151)
ctx.set_pointer_mode(nv.PTR_HOST)
ctx.profile_enable(True)
for rep in range(3):
    ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
    ctx.synthesize_linear_hasher(q, qs, 774, t, 0)
    ctx.synchronize(); print(f"linear hasher: {1e3*(time.perf_counter()-t0):.2f} ms", {k: round(v[0], 3) for k, v in ctx.profile().items()})
req, mq = synthetic.precompile_trace(0, 60, seed=5, max_rounds=6)
tails = o.queue_push_chain_log(o.encode_log_queries(req))[1]
w = ctx._precompile(0, req, tails, mq, 293, np.zeros(1, nv.QUEUE_STATE12))
for rep in range(3):
    ctx.profile_reset(); ctx.synchronize(); t0 = time.perf_counter()
    ctx.synthesize_keccak_round_function(w, t, 0, 1, 0)
    ctx.synchronize(); print(f"keccak x1: {1e3*(time.perf_counter()-t0):.2f} ms", {k: round(v[0], 3) for k, v in ctx.profile().items()})
