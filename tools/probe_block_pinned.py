"""zkw_blocks_run with the blocks' inputs in PINNED host memory (hipHostRegister on the arrays; a host would take them from
zkw_buffer_alloc(pinned_host = 1)): the uploads of 96 blocks' inputs stop serialising through the runtime's staging copies.
Usage: probe_block_pinned.py K [reps] [pin=1]"""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 96
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pin = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
base = [synthetic.block_production(seed=1 + k) for k in range(4)]
keep = []


def pinned(a):
    a = np.ascontiguousarray(a)
    if pin and a.nbytes:
        rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0)
        assert rc == 0, rc
    keep.append(a)
    return a


for b in base:
    for k in ("vm_memory_queries", "decommit_queries", "log_queries"):
        b[k] = pinned(b[k])
    b["precompile_memory_queries"] = [pinned(x) for x in b["precompile_memory_queries"]]
blocks = [base[k % 4] for k in range(K)]
warm = nv.Block(0, base[0]); warm.synthesize(1 << 20, ring_slots=1); warm.free()
for r in range(reps):
    t0 = time.perf_counter()
    bs = nv.Block.run_many(0, blocks)
    t1 = time.perf_counter()
    n = nv.Block.synthesize_many(bs, 1 << 20, ring_slots=1)
    t2 = time.perf_counter()
    for b in bs: b.free()
    print(f"K={K:3d} pinned={pin} round {r}: builders {1e3*(t1-t0):.0f} ms, synthesis of {n} instances {1e3*(t2-t1):.0f} ms -> {K/(t2-t0):.2f} blocks/s")
