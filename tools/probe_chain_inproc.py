"""Same process: RAM builder's chains (compact outputs) vs the plain batch chain API, same number of chains."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native, synthetic
import bench
B, n = int(sys.argv[1]), 20000
dev = torch.device('cuda', 0)
ctx = native.Context(0); lib = native.load()
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream); ctx.set_pointer_mode(native.PTR_DEVICE)
def chain_api(nc, L):
    enc = torch.empty((nc * L, 8), dtype=torch.int64, device='cuda'); enc.random_(0, 2**62)
    tails = torch.empty((nc * L, 12), dtype=torch.int64, device='cuda')
    offs = (np.arange(nc + 1, dtype=np.uint64) * L)
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, enc.data_ptr(), offs.ctypes.data, nc, None, tails.data_ptr()))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"chain API  {nc} chains x {L}: {min(ts)*1e6/L:.2f} us/step", flush=True)
def ram(B):
    base, q = bench.make_inputs(B, n, 0, dev)
    offs = np.arange(B + 1, dtype=np.uint64) * n
    w = native.RamWitness(ctx)
    for _ in range(2):
        ctx.profile_enable(True); ctx.profile_reset()
        ctx.compute_ram_circuit_snapshots((q.data_ptr(), B * n), bench.CAPACITY, 0, block_offsets=offs, witness=w)
        torch.cuda.synchronize()
        prof = ctx.profile(); ctx.profile_enable(False)
        for name, (ms, calls) in prof.items():
            if 'chain' in name:
                print(f"RAM build  {2*B} chains x {n}: {name} {ms*1e3/n:.2f} us/step", flush=True)
    return w
chain_api(2 * B, 2000)
w = ram(B)
chain_api(2 * B, 2000)
del w
chain_api(2 * B, 2000)

# the plain chain API on the builder's own arrays
B2 = 2 * B
base, q = bench.make_inputs(B2, n, 0, dev)
offs2 = np.arange(B2 + 1, dtype=np.uint64) * n
w2 = native.RamWitness(ctx)
ctx.compute_ram_circuit_snapshots((q.data_ptr(), B2 * n), bench.CAPACITY, 0, block_offsets=offs2, witness=w2)
torch.cuda.synchronize()
for what, name in ((native.RAM_UNSORTED_ENC, "unsorted_enc"), (native.RAM_SORTED_ENC, "sorted_enc")):
    ptr = w2.device_ptr(what)
    tails = torch.empty((B2 * n, 12), dtype=torch.int64, device='cuda')
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, ptr, offs2.ctypes.data, B2, None, tails.data_ptr()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"chain API on the builder's {name}: {B2} chains x {n}: {dt*1e6/n:.2f} us/step", flush=True)
    # the same bytes copied into a torch tensor
    enc_copy = torch.empty((B2 * n, 8), dtype=torch.int64, device='cuda')
    import ctypes
    native._check(0)
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(enc_copy.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(B2 * n * 64), 3)
    for _ in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, enc_copy.data_ptr(), offs2.ctypes.data, B2, None, tails.data_ptr()))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"chain API on a torch copy of {name}: {dt*1e6/n:.2f} us/step", flush=True)
    del tails, enc_copy
