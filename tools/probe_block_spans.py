import sys, time
sys.path.insert(0, '.')
from era_zkevm_test_harness_amd import native as nv, synthetic
K = int(sys.argv[1])
base = [synthetic.block_production(seed=1 + k) for k in range(4)]
blocks = [base[k % 4] for k in range(K)]
warm = nv.Block(0, base[0]); warm.free()
for r in range(2):
    t0 = time.perf_counter()
    bs = nv.Block.run_many(0, blocks)
    t1 = time.perf_counter()
    if r == 1:
        for idx in (0, K // 2, K - 1):
            print(f"--- block {idx} (builders {1e3*(t1-t0):.0f} ms)")
            for name, s, e in sorted(bs[idx].timings(), key=lambda x: x[1]):
                print(f"  {name:34s} {s:8.1f} -> {e:8.1f}  ({e-s:7.1f} ms)")
    for b in bs: b.free()
