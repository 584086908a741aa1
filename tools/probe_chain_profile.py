"""Does the in-library HIP-event profiling change the chain kernel's speed? Wall-clock, profile on/off, both paths."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native
import bench
B, n = int(sys.argv[1]), 20000
dev = torch.device('cuda', 0)
ctx = native.Context(0); lib = native.load()
s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream); ctx.set_pointer_mode(native.PTR_DEVICE)
base, q = bench.make_inputs(B, n, 0, dev)
offs = np.arange(B + 1, dtype=np.uint64) * n
w = native.RamWitness(ctx)
def build(tag):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.compute_ram_circuit_snapshots((q.data_ptr(), B * n), bench.CAPACITY, 0, block_offsets=offs, witness=w)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"RAM build {2*B} chains x {n}, {tag}: wall {min(ts)*1e3:.1f} ms", flush=True)
enc = torch.empty((2 * B * 2000, 8), dtype=torch.int64, device='cuda'); enc.random_(0, 2**62)
tails = torch.empty((2 * B * 2000, 12), dtype=torch.int64, device='cuda')
o2 = (np.arange(2 * B + 1, dtype=np.uint64) * 2000)
def api(tag):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        native._check(lib.zkw_queue_push_chain_full_batch(ctx.handle, enc.data_ptr(), o2.ctypes.data, 2 * B, None, tails.data_ptr()))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"chain API {2*B} chains x 2000, {tag}: {min(ts)*1e6/2000:.2f} us/step", flush=True)
build("profile off"); api("profile off")
ctx.profile_enable(True)
build("profile on"); api("profile on")
ctx.profile_enable(False)
build("profile off again")
