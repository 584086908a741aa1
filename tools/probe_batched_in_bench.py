"""reproduces bench.py's batched full-block leg in isolation (with / without torch's HIP context alive)"""
import sys, os
sys.path.insert(0, '.')
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
import torch
if mode == "torch":
    torch.cuda.init(); x = torch.zeros(16, device="cuda:0"); torch.cuda.synchronize()
import bench
from era_zkevm_test_harness_amd import synthetic, native
blk = synthetic.block_production(seed=1)
if mode != "nowarm":
    first = native.Block(0, blk); first.synthesize(1 << 20, ring_slots=2); first.free()
print(bench.full_blocks_batched(0, blk, K=K))
