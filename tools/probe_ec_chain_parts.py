import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from era_zkevm_test_harness_amd import native, synthetic
ctx = native.Context(0)
n_inst = 8
req, mq = synthetic.precompile_trace(2, 7 * n_inst, seed=5)
tails = ctx.queue_push_chain_log(ctx.encode_log_queries(req))[1]
w = ctx._precompile(2, req, tails, mq, 7, np.zeros(1, native.QUEUE_STATE12))
t = native.Trace(ctx, 1 << 20, n_inst, n_cols=native.EK_COLS)
try:
    ctx.synthesize_ecrecover(w, t, 0, n_inst, 0); ctx.synchronize()
except Exception as e: print("err", str(e)[:80])
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(3):
    try: ctx.synthesize_ecrecover(w, t, 0, n_inst, 0); ctx.synchronize()
    except Exception: pass
print({k: round(v[0] / 3, 3) for k, v in ctx.profile().items() if k.startswith("k_ec_")})
