#!/bin/bash
# the rocprofv3 --kernel-trace --stats pass of the default bench command alone (step 2 of tools/run_round_profiles.sh)
set -u
OUT=${1:-gpurun_out/profiles}
ROOT=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
for K in ${2:-16}; do
cd /tmp && rm -rf /tmp/prof_stats && timeout -s KILL 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
    env ZKW_BATCHED_BLOCKS=$K python "$ROOT/bench.py" --no-cpu-baseline --no-sensitivity > "$OUT/bench_default_under_rocprofv3.json" 2> "$OUT/rocprof_stats.err"
echo "K=$K rc=$?"
f=$(ls /tmp/prof_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$OUT/bench_default_kernel_stats.csv" && break
done
ls -la "$OUT"
