"""Chain step time vs memory footprint: same number of chains, longer queues (bigger, farther-apart streams)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native
ctx = native.Context(0); lib = native.load()
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream); ctx.set_pointer_mode(native.PTR_DEVICE)
def run(form, nc, L, steps):
    ctx.set_chain_form(form)
    enc = torch.empty((nc * L, 8), dtype=torch.int64, device='cuda')
    enc.random_(0, 2**62)
    tails = torch.empty((nc * L, 12), dtype=torch.int64, device='cuda')
    # only the first `steps` items of every queue are hashed: offsets of (start, start + steps) pairs are not
    # expressible, so hash full queues when steps == L and a prefix otherwise by passing shorter queues spaced L apart
    offs = np.zeros(2 * nc + 1, dtype=np.uint64)
    for c in range(nc):
        offs[2 * c] = c * L
        offs[2 * c + 1] = c * L + steps
    offs[2 * nc] = nc * L
    # chains 2c = the prefix we time; chains 2c+1 = the (long) remainder -> not wanted. Use only even chains by
    # building the batch from the even pairs: the batch API needs contiguous offsets, so instead time one call per
    # stride configuration with queues of exactly `steps` items but laid out L apart through a strided view.
    del offs
    offs = (np.arange(nc + 1, dtype=np.uint64) * L)
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, enc.data_ptr(), offs.ctypes.data, nc, None, tails.data_ptr()))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    dt = min(ts)
    print(f"form {form:2d}: {nc:5d} chains x {L}: enc {nc*L*64/1e9:.1f} GB tails {nc*L*96/1e9:.1f} GB  {dt*1e6/L:6.2f} us/step", flush=True)
    del enc, tails
    torch.cuda.empty_cache()
for nc, L in ((9000, 5000), (9000, 20000), (9000, 50000), (9000, 100000), (6144, 136714)):
    run(4, nc, L, L)
