#!/bin/bash
pp='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]; print(round(d["value"]), round(d["ms_per_step"]), d["config"].get("blocks"), d["config"].get("pipelines"), {a:round(k[a]) for a in ("k_chain_full_q4","k_ram_fill_poseidon") if a in k}, "hbm", round(d["hbm_used_GB"]))'
for P in 3 4; do echo "== --pipelines $P"; timeout 900 python bench.py --steps 6 --pipelines $P --no-full-block --no-cpu-baseline --no-sensitivity --no-validate --no-h2d 2>/tmp/e.txt | python3 -c "$pp" || tail -3 /tmp/e.txt; done
