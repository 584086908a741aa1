#!/bin/bash
pp='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels_ms_per_step"]; print(round(d["value"]), round(d["ms_per_step"]), {a:round(k[a]) for a in ("k_ram_fill_tail","k_ram_fill_A","k_gp_apply")}, "blocks/s", d.get("full_block",{}).get("batched",{}).get("blocks_per_s"), "block ms", d.get("full_block",{}).get("wall_ms"), {a:round(b["circuits_per_s"]) for a,b in d.get("hash_circuits",{}).items()}, d["hash_circuits"]["ecrecover"].get("two_calls_in_flight",{}).get("circuits_per_s"), "hbm", d["hbm_used_GB"], "valid", d["validation"]["ok"] if d.get("validation") else None)'
for i in 1 2; do timeout 1200 python bench.py --steps 5 --no-cpu-baseline 2>/tmp/err_$i.txt | python3 -c "$pp"; tail -2 /tmp/err_$i.txt | grep -v amdgpu; done
