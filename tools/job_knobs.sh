#!/bin/bash
# chain-service knobs at 96 / 192 blocks in flight; the netlist synthesis probes (cost of the closed-form section); rocprofv3 at 48 blocks in flight
OUT=${1:-gpurun_out/r05/knobs}
mkdir -p "$OUT"
export TMPDIR=/tmp
{
for P in "4 4000" "6 4000" "8 4000" "4 10000" "6 10000"; do
    set -- $P
    echo "== workers $1 window $2 us"
    ZKW_CHAIN_WORKERS=$1 ZKW_CHAIN_WINDOW_US=$2 timeout -s KILL 300 python tools/probe_block_concurrency.py 96 3 2>&1 | grep "^K=" | tail -2
done
echo "== K=192 default"
timeout -s KILL 400 python tools/probe_block_concurrency.py 192 2 2>&1 | grep "^K=" | tail -1
} > "$OUT/blocks_knobs.txt" 2>&1
{
echo "== tools/probe_ecrecover_synth.py"; timeout -s KILL 300 python tools/probe_ecrecover_synth.py 2>&1 | grep -v amdgpu.ids
echo "== tools/probe_netlist_perf.py"; timeout -s KILL 300 python tools/probe_netlist_perf.py 2>&1 | grep -v amdgpu.ids
} > "$OUT/synthesis_probes.txt" 2>&1
timeout -s KILL 900 python bench.py --steps 2 --no-cpu-baseline --no-full-block --no-sensitivity --no-validate > "$OUT/bench_hash.json" 2> "$OUT/bench_hash.err"
cat "$OUT/blocks_knobs.txt"; cat "$OUT/synthesis_probes.txt" | cut -c1-400
python3 - "$OUT/bench_hash.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"])
for k, v in d["hash_circuits"].items():
    print(k, round(v["circuits_per_s"]), {a: round(b["circuits_per_s"]) for a, b in v.items() if isinstance(b, dict) and "circuits_per_s" in b})
PY
