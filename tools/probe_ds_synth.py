"""Throughput of CodeDecommittmentsSorter synthesis at production geometry (capacity 117 500, 2^20 rows)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native, synthetic
ctx = native.Context(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)
capacity, n_rows, n_inst = 117500, 1 << 20, 16
q = synthetic.decommit_trace(capacity * n_inst, 40000, seed=5)
t0 = time.perf_counter()
w = ctx.compute_decommitts_sorter_circuit_snapshots(q, capacity)
torch.cuda.synchronize()
print(f"witness build ({q.size} requests, {n_inst} instances): {1e3*(time.perf_counter()-t0):.1f} ms (host pointers, first call)")
t = native.Trace(ctx, n_rows, n_inst)
ctx.synthesize_decommit_sorter(w, t)
torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    ctx.synthesize_decommit_sorter(w, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
prof = ctx.profile()
bytes_per_inst = native.circuit_fill_bytes(2, capacity, n_rows)[0]  # bytes a synthesis writes into a slot that already holds this layout (slot layout tags)
print(f"synthesis: {dt*1e3:.2f} ms per {n_inst} instances = {n_inst/dt:.0f} circuits/s, {n_inst*bytes_per_inst/dt/1e12:.2f} TB/s of bytes written")
for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:24s} {ms/reps:8.3f} ms per pass ({cnt//reps} launches)")
bad, first = ctx.check_if_satisfied_decommit_sorter(t, 3, capacity)
print("check:", bad, first)
