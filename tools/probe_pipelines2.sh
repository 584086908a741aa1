#!/bin/bash
# throughput-leg packing experiments (3 timed steps each, no side legs): pipelines x chain form
set -u
OUT=${1:-gpurun_out/pipes}
mkdir -p $OUT
run() { name=$1; shift; env "$@" timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-block > $OUT/$name.json 2> $OUT/$name.err; python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get("kernels_ms_per_step", {})
    print(sys.argv[2], round(d["value"], 1), "circuits/s", round(d["ms_per_step"]), "ms/step", "blocks", d["config"]["blocks_per_gpu"], {x: round(k[x]) for x in list(k)[:4]})
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run p2_quad ZKW_PIPELINES=2
run p3_pair ZKW_PIPELINES=3 ZKW_CHAIN_FORM=2
run p2_pair ZKW_PIPELINES=2 ZKW_CHAIN_FORM=2
run p4_pair ZKW_PIPELINES=4 ZKW_CHAIN_FORM=2
