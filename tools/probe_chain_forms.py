import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native
ctx = native.Context(0); lib = native.load()
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream); ctx.set_pointer_mode(native.PTR_DEVICE)
def run(form, nc, L):
    ctx.set_chain_form(form)
    enc = torch.randint(0, 2**62, (nc * L, 8), dtype=torch.int64, device='cuda')
    tails = torch.empty((nc * L, 12), dtype=torch.int64, device='cuda')
    offs = np.arange(nc + 1, dtype=np.uint64) * L
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, enc.data_ptr(), offs.ctypes.data, nc, None, tails.data_ptr()))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    dt = min(ts)
    print(f"form {form:2d}: {nc:5d} chains x {L}: {dt*1e3:8.2f} ms  {dt*1e6/L:6.2f} us/step  {nc*L/dt/1e6:8.1f} Mperm/s", flush=True)
for nc, L in ((16, 20000), (4096, 5000), (6144, 4000), (8192, 2500), (11416, 2500), (16384, 2000), (11416, 20000)):
    for form in (16, 4):
        run(form, nc, L)
