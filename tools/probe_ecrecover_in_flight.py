import sys, time
import numpy as np
sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
from era_zkevm_test_harness_amd import native, synthetic
n_rows = 1 << 20
req, mq = synthetic.precompile_trace(2, 7 * 32, seed=5, max_rounds=6)
mem_in = np.zeros(1, native.QUEUE_STATE12)
ctxs, ws, ts = [], [], []
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for k in range(NT):
    c = native.Context(0)
    w = c._precompile(2, req, c.queue_push_chain_log(c.encode_log_queries(req))[1], mq, 7, mem_in)
    t = native.Trace(c, n_rows, 32, n_cols=native.EK_COLS)
    c.synthesize_ecrecover(w, t, 0, 32, 0); c.synchronize()
    ctxs.append(c); ws.append(w); ts.append(t)
def calls(k, reps=6):
    for _ in range(reps):
        ctxs[k].synthesize_ecrecover(ws[k], ts[k], 0, 32, 0)
    ctxs[k].synchronize()
for nt in range(1, NT + 1):
    best = None
    with ThreadPoolExecutor(nt) as ex:
        for _ in range(3):
            t0 = time.perf_counter()
            for f in [ex.submit(calls, k) for k in range(nt)]: f.result()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
    print(f"{nt} calls in flight: {nt * 6 * 32 / best:.0f} circuits/s")
