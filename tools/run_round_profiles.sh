#!/bin/bash
# Reproduces the artefacts under profiles/rNN/ on a GPU box (run from the repo root):
#   bash tools/run_round_profiles.sh gpurun_out/r01
# then copy the directory's contents into profiles/r01/ and run `python tools/make_profile_readme.py`.
set -u
OUT=${1:-gpurun_out/profiles}
ROOT=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
# 1. the default bench, without a profiler
timeout -s KILL 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout -s KILL 900 python bench.py --pipelines 1 --no-cpu-baseline > "$OUT/bench_sequential.json" 2> "$OUT/bench_sequential.err"
# 2. the same command under rocprofv3 --kernel-trace --stats
cd /tmp && rm -rf /tmp/prof_stats && timeout -s KILL 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
    python "$ROOT/bench.py" > "$OUT/bench_default_under_rocprofv3.json" 2> "$OUT/rocprof_stats.err"
cp "$(ls /tmp/prof_stats/*/*kernel_stats.csv | head -1)" "$OUT/bench_default_kernel_stats.csv"
# 3. HBM counters, one pass each (never combined with other trace domains)
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_pmc && timeout -s KILL 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -- \
        python "$ROOT/bench.py" --pipelines 1 --blocks 32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/rocprof_pmc_$C.err"
    cp "$(ls /tmp/prof_pmc/*/*counter_collection.csv | head -1)" "$OUT/bench_b32_pmc_$C.csv"
done
cd "$ROOT"
# 4. synthesis throughput of the other circuit types at production geometry
for P in ds es ld ss; do
    echo "== tools/probe_${P}_synth.py" >> "$OUT/synthesis_probes.txt"
    timeout -s KILL 300 python tools/probe_${P}_synth.py 2>&1 | grep -v amdgpu.ids >> "$OUT/synthesis_probes.txt"
done
ls -la "$OUT"
