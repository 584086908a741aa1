#!/bin/bash
# headline experiments: pipelines x chain form (bench.py's throughput leg only). Usage: bash tools/probe_pipelines.sh OUT
OUT=${1:-gpurun_out/pipes}; mkdir -p "$OUT"
run() { # name, env...
    name=$1; shift
    env "$@" timeout -s KILL 400 python bench.py --no-cpu-baseline --no-full-block --steps 3 --warmup 1 > "$OUT/$name.json" 2> "$OUT/$name.err"
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels_ms_per_step"]
    print(sys.argv[2], round(d["value"], 1), "circuits/s", round(d["ms_per_step"]), "ms/step", "P", d["config"]["pipelines_per_gpu"], "B", d["config"]["blocks_per_gpu"],
          {n: round(v) for n, v in k.items() if v > 300})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for spec in "$@"; do :; done
run p2_quad ZKW_PIPELINES=2
run p3_quad ZKW_PIPELINES=3
run p4_quad ZKW_PIPELINES=4
run p3_lane ZKW_PIPELINES=3 ZKW_CHAIN_FORM=1
run p4_lane ZKW_PIPELINES=4 ZKW_CHAIN_FORM=1
