"""per-kernel times of the ECRecover circuit's synthesis at production geometry (2^20 rows, 7 requests per instance)"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native, synthetic
ctx = native.Context(0)
for n_inst in (8, 32):
    req, mq = synthetic.precompile_trace(2, 7 * n_inst, seed=5)
    tails = ctx.queue_push_chain_log(ctx.encode_log_queries(req))[1]
    w = ctx._precompile(2, req, tails, mq, 7, np.zeros(1, native.QUEUE_STATE12))
    t = native.Trace(ctx, 1 << 20, n_inst, n_cols=native.EK_COLS)
    ctx.synthesize_ecrecover(w, t, 0, n_inst, 0); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ctx.synthesize_ecrecover(w, t, 0, n_inst, 0); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"ecrecover: {n_inst} instances {best*1e3:.2f} ms = {n_inst/best:.0f} circuits/s", {k: round(v[0] / 3, 3) for k, v in ctx.profile().items()})
    ctx.profile_enable(False)
    assert all(ctx.check_if_satisfied_ecrecover(t, i, 7) == (0, (0, 0, 0)) for i in (0, n_inst - 1))
    t.free(); w.free()
