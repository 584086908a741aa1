"""per-kernel times of the setup-as-field-elements path at production size (2^20 rows, LDE x 2, cap 16): NTT passes as GB/s over their
algorithmic traffic (read + write of the array per pass), leaf hashing as permutations/s, and the whole zkw_setup_commit of a layout"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native
ctx = native.Context(0)
ctx.set_pointer_mode(native.PTR_DEVICE)
import torch
log_n, n_cols, lde = 20, 131, 2
n = 1 << log_n
P = 0xFFFFFFFF00000001
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.randint(0, 2**62, (n_cols, n), dtype=torch.int64, device="cuda", generator=g)
out = torch.empty((lde, n_cols, n), dtype=torch.int64, device="cuda")
lib = native.load()
import ctypes as C
def run(fn, reps=3):
    fn(); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = {k: (v[0] / reps, v[1] // reps) for k, v in ctx.profile().items()}
    ctx.profile_enable(False)
    return dt, prof
bytes_arr = n_cols * n * 8
dt, prof = run(lambda: native._check(lib.zkw_ntt(ctx.handle, vals.data_ptr(), out.data_ptr(), log_n, n_cols, 0)))
print(f"zkw_ntt 2^{log_n} x {n_cols} columns: {dt*1e3:.2f} ms", {k: f"{ms:.2f} ms = {2*bytes_arr/ms/1e6:.0f} GB/s (read + write)" for k, (ms, c) in prof.items()})
dt, prof = run(lambda: native._check(lib.zkw_lde(ctx.handle, vals.data_ptr(), log_n, n_cols, lde, out.data_ptr())))
print(f"zkw_lde x{lde}: {dt*1e3:.2f} ms", {k: f"{ms:.2f} ms / {c} launches = {c*2*bytes_arr/ms/1e6:.0f} GB/s" for k, (ms, c) in prof.items()})
cap = torch.empty((16, 4), dtype=torch.int64, device="cuda")
dt, prof = run(lambda: native._check(lib.zkw_merkle_tree_with_cap(ctx.handle, out.data_ptr(), lde, n_cols, n, 16, cap.data_ptr(), None)))
perms = lde * n * ((n_cols + 7) // 8)
print(f"zkw_merkle_tree_with_cap {lde * n} leaves of {n_cols}: {dt*1e3:.2f} ms", {k: f"{ms:.2f} ms" for k, (ms, c) in prof.items()},
      f"leaf hashing {perms/prof['k_merkle_leaves'][0]/1e6:.2f} G permutations/s, reads {lde*bytes_arr/prof['k_merkle_leaves'][0]/1e6:.0f} GB/s")
for ctype, name in ((8, "RAMPermutation"), (11, "EventsSorter"), (4, "LogDemuxer")):
    t0 = time.perf_counter()
    native._check(lib.zkw_setup_commit(ctx.handle, ctype, 0, log_n, lde, 16, cap.data_ptr())); ctx.synchronize()
    print(f"zkw_setup_commit {name}: {(time.perf_counter()-t0)*1e3:.0f} ms wall (host union-find of the copy permutation included)")
