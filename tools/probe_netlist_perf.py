"""per-kernel times of the netlist circuits' synthesis at production geometry (2^20 rows, reference capacities)"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native, synthetic
ctx = native.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8  # instances per synthesis call (8 = the bench's hash_circuits leg)
n_rows = 1 << 20
mem_in = np.zeros(1, native.QUEUE_STATE12)
for name, kind, n_req, cap, cols, synth in (("keccak", 0, 1400 * N // 8, 293, native.KC_COLS, ctx.synthesize_keccak_round_function),
                                            ("sha256", 1, 6000 * N // 8, 2206, native.SC_COLS, ctx.synthesize_sha256_round_function)):
    req, mq = synthetic.precompile_trace(kind, n_req, seed=5, max_rounds=6)
    tails = ctx.queue_push_chain_log(ctx.encode_log_queries(req))[1]
    w = ctx._precompile(kind, req, tails, mq, cap, mem_in)
    n = min(N, w.num_instances)
    t = native.Trace(ctx, n_rows, n, n_cols=cols)
    synth(w, t, 0, n, 0); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); synth(w, t, 0, n, 0); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    prof = ctx.profile()
    print(name, f"{n} instances {best*1e3:.2f} ms = {n/best:.0f} circuits/s", {k: round(v[0] / 3, 3) for k, v in prof.items()})
    ctx.profile_enable(False)
    t.free(); w.free()
if N != 8:
    sys.exit(0)
q = synthetic.mixed_log_queue(4000, seed=3)[:700]
t = native.Trace(ctx, n_rows, 1, n_cols=native.LH_COLS)
ctx.synthesize_linear_hasher(q, np.zeros(1, native.QUEUE_STATE4), 774, t, 0); ctx.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
t0 = time.perf_counter(); ctx.synthesize_linear_hasher(q, np.zeros(1, native.QUEUE_STATE4), 774, t, 0); ctx.synchronize(); dt = time.perf_counter() - t0
print("linear hasher", f"{dt*1e3:.2f} ms", {k: round(v[0], 3) for k, v in ctx.profile().items()})
queues = [synthetic.mixed_log_queue(4000, seed=3 + k)[:700] for k in range(8)]
t8 = native.Trace(ctx, n_rows, 8, n_cols=native.LH_COLS)
st = np.zeros(8, native.QUEUE_STATE4)
qtails = [ctx.queue_push_chain_log(ctx.encode_log_queries(q))[1] for q in queues]  # the queues' states, as the sorter that builds a queue holds them
ctx.synthesize_linear_hasher_batch(queues, st, 774, t8, 0, tails=qtails); ctx.synchronize()
ctx.profile_reset()
t0 = time.perf_counter(); ctx.synthesize_linear_hasher_batch(queues, st, 774, t8, 0, tails=qtails); ctx.synchronize(); dt = time.perf_counter() - t0
print("linear hasher, 8 queues per call", f"{dt*1e3:.2f} ms = {8/dt:.0f} circuits/s", {k: round(v[0], 3) for k, v in ctx.profile().items()})
# StorageApplication (type 10): 8 instances of 33 tree queries each (Blake2s Merkle walks, 8 481 cycles per instance)
ctx.profile_enable(False)
sq, _existing = synthetic.storage_application_trace(200, seed=4, write_fraction=0.6)
stails = ctx.queue_push_chain_log(ctx.encode_log_queries(sq))[1]
tree, answers = synthetic.storage_tree_for(sq, seed=1)
idx, paths = answers(sq)
w = ctx.decompose_into_storage_application_witnesses(sq, stails, idx, paths, tree.root, tree.next_enumeration_index, 33)
n = min(8, w.num_instances)
t = native.Trace(ctx, n_rows, n, n_cols=native.SA_COLS)
ctx.synthesize_storage_application(w, t, 0, n, 0); ctx.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); ctx.synthesize_storage_application(w, t, 0, n, 0); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
print("storage application", f"{n} instances {best*1e3:.2f} ms = {n/best:.0f} circuits/s", {k: round(v[0] / 3, 3) for k, v in ctx.profile().items()})
assert all(ctx.check_if_satisfied_storage_application(t, i, 33)[0] == 0 for i in range(n))
