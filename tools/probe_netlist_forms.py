import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from era_zkevm_test_harness_amd import native, synthetic
ctx = native.Context(0)
N=8; n_rows=1<<20; mem_in=np.zeros(1, native.QUEUE_STATE12)
for form in (0,1):
    ctx.set_netlist_fill_form(form)
    for name, kind, n_req, cap, cols, synth in (("keccak", 0, 1400, 293, native.KC_COLS, ctx.synthesize_keccak_round_function), ("sha256", 1, 6000, 2206, native.SC_COLS, ctx.synthesize_sha256_round_function)):
        req, mq = synthetic.precompile_trace(kind, n_req, seed=5, max_rounds=6)
        tails = ctx.queue_push_chain_log(ctx.encode_log_queries(req))[1]
        w = ctx._precompile(kind, req, tails, mq, cap, mem_in)
        n = min(N, w.num_instances)
        t = native.Trace(ctx, n_rows, n, n_cols=cols)
        synth(w, t, 0, n, 0); ctx.synchronize()
        ctx.profile_enable(True); ctx.profile_reset()
        best=1e9
        for _ in range(3):
            t0=time.perf_counter(); synth(w, t, 0, n, 0); ctx.synchronize(); best=min(best,time.perf_counter()-t0)
        print("form",form,name, f"{best*1e3:.2f} ms = {n/best:.0f} circuits/s", {k: round(v[0]/3,3) for k,v in ctx.profile().items()})
        ctx.profile_enable(False); t.free(); w.free()
