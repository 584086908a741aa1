"""ECRecover synthesis at production geometry for several numbers of instances per call (default 8 16 24 27 28 32 36 64): call time,
circuits/s and the per-kernel HIP-event times of the main stream (the side stream's kernels are in a rocprofv3 trace only).
Usage: python tools/probe_ecrecover_sizes.py [instances per call ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from era_zkevm_test_harness_amd import native, synthetic
ctx = native.Context(0)
for n_inst in [int(x) for x in sys.argv[1:]] or (8, 16, 24, 27, 28, 32, 36, 64):
    req, mq = synthetic.precompile_trace(2, 7 * n_inst, seed=5)
    tails = ctx.queue_push_chain_log(ctx.encode_log_queries(req))[1]
    w = ctx._precompile(2, req, tails, mq, 7, np.zeros(1, native.QUEUE_STATE12))
    t = native.Trace(ctx, 1 << 20, n_inst, n_cols=native.EK_COLS)
    ctx.synthesize_ecrecover(w, t, 0, n_inst, 0); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ctx.synthesize_ecrecover(w, t, 0, n_inst, 0); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"ecrecover: {n_inst} instances {best*1e3:.2f} ms = {n_inst/best:.0f} circuits/s", {k: round(v[0] / 3, 3) for k, v in ctx.profile().items() if v[0] / 3 > 0.2}, flush=True)
    ctx.profile_enable(False)
    t.free(); w.free()
