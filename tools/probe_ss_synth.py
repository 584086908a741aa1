"""Throughput of StorageSorter synthesis at production geometry (capacity 46 921, 2^20 rows)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native, synthetic
ctx = native.Context(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)
capacity, n_rows, n_inst = 46921, 1 << 20, 16
q = synthetic.storage_trace(capacity * n_inst, capacity * n_inst // 6, seed=5)
t0 = time.perf_counter()
w = ctx.compute_storage_dedup_and_sort(q, capacity)
torch.cuda.synchronize()
n_inst = w.num_instances
print(f"witness build ({q.size} records, {n_inst} instances): {1e3*(time.perf_counter()-t0):.1f} ms (host pointers, first call)")
t0 = time.perf_counter()
w2 = ctx.compute_storage_dedup_and_sort(q, capacity)
torch.cuda.synchronize()
print(f"witness build, second call: {1e3*(time.perf_counter()-t0):.1f} ms")
w2.free()
t = native.Trace(ctx, n_rows, n_inst)
ctx.synthesize_storage_sorter(w, t)
torch.cuda.synchronize()
ctx.profile_enable(True); ctx.profile_reset()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    ctx.synthesize_storage_sorter(w, t)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
prof = ctx.profile()
bytes_per_inst = native.circuit_fill_bytes(9, capacity, n_rows)[0]  # bytes a synthesis writes into a slot that already holds this layout (slot layout tags)
print(f"synthesis: {dt*1e3:.2f} ms per {n_inst} instances = {n_inst/dt:.0f} circuits/s, {n_inst*bytes_per_inst/dt/1e12:.2f} TB/s of bytes written")
for k, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:24s} {ms/reps:8.3f} ms per pass ({cnt//reps} launches)")
bad, first = ctx.check_if_satisfied_storage_sorter(t, 1, capacity)
print("check:", bad, first)
