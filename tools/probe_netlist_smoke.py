import numpy as np, sys, time
sys.path.insert(0,'.')
from era_zkevm_test_harness_amd import native, synthetic
from oracle import pyoracle as oracle
oracle.build()
ctx = native.Context(0)
n_rows = 1 << 18
for kind, cap, cols, name, synth, check, osynth, ocheck in (
    (1, 7, native.SC_COLS, "sha256_rounds", ctx.synthesize_sha256_round_function, ctx.check_if_satisfied_sha256_round_function, oracle.sha256_round_synthesize, oracle.sha256_round_check),
    (0, 6, native.KC_COLS, "keccak_rounds", ctx.synthesize_keccak_round_function, ctx.check_if_satisfied_keccak_round_function, oracle.keccak_round_synthesize, oracle.keccak_round_check)):
    req, mq = synthetic.precompile_trace(kind, 9, seed=3, max_rounds=4)
    tails = ctx.queue_push_chain_log(ctx.encode_log_queries(req))[1]
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    w = ctx._precompile(kind, req, tails, mq, cap, mem_in)
    ow = oracle.precompile_build(kind, req, oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1], mq, cap, np.zeros(1, oracle.QUEUE_STATE12))
    ni = w.num_instances
    t = native.Trace(ctx, n_rows, ni, n_cols=cols)
    t0=time.time(); synth(w, t, 0, ni, 0); ctx.synchronize(); print(name, "synth", time.time()-t0)
    for i in range(ni):
        got = t.get(i)
        want = osynth(ow, i, cap, n_rows)
        eq = np.array_equal(got, want)
        print(name, i, "equal", eq, "check", check(t, i, cap))
        if not eq:
            d = np.argwhere(got != want)
            print(" first diffs", d[:8].tolist(), [ (int(got[c,r]), int(want[c,r])) for c,r in d[:8]])
    t.free(); w.free()
ctx.close()
