#!/bin/bash
# does the full-block leg (which runs before the timed region) slow the throughput leg down? (hardware queues it leaves mapped)
pp='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]; print(round(d["value"]), round(d["ms_per_step"]), {a:round(k[a]) for a in ("k_ram_fill_tail","k_ram_fill_A","k_gp_apply")}, d.get("full_block",{}).get("batched",{}).get("blocks_per_s"))'
echo "== default order (full-block leg first)"; timeout 900 python bench.py --steps 5 --no-cpu-baseline --no-sensitivity --no-validate --no-h2d 2>/dev/null | python3 -c "$pp"
echo "== ZKW_CHAIN_WORKERS=4"; ZKW_CHAIN_WORKERS=4 timeout 900 python bench.py --steps 5 --no-cpu-baseline --no-sensitivity --no-validate --no-h2d 2>/dev/null | python3 -c "$pp"
echo "== ZKW_BATCHED_BLOCKS=16"; ZKW_BATCHED_BLOCKS=16 timeout 900 python bench.py --steps 5 --no-cpu-baseline --no-sensitivity --no-validate --no-h2d 2>/dev/null | python3 -c "$pp"
