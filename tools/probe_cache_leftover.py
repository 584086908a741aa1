"""free HBM before / after the full-block legs and after zkw_trim_caches (what does the batched leg leave behind?)"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from era_zkevm_test_harness_amd import synthetic, native
dev = torch.device("cuda", 0)
def free(tag):
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(dev); print(f"{tag}: free {f/1e9:.2f} GB of {t/1e9:.2f}", flush=True)
free("start")
blk = synthetic.block_production(seed=1)
first = native.Block(0, blk); first.synthesize(1 << 20, ring_slots=2); first.free()
free("after one block")
native.trim_caches(); free("after trim")
print(bench.full_blocks_batched(0, blk, K=int(sys.argv[1]) if len(sys.argv) > 1 else 48, rounds=2))
free("after batched")
native.trim_caches(); free("after trim")
torch.cuda.empty_cache(); free("after empty_cache")
