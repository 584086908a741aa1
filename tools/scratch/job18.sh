#!/bin/bash
mkdir -p gpurun_out
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_s && timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $ROOT/bench.py --pipelines 1 --steps 1 --warmup 0 --no-cpu-baseline --no-full-block --no-h2d --no-sensitivity --no-validate > $ROOT/gpurun_out/j18_bench.json 2> $ROOT/gpurun_out/j18_bench.err
cd $ROOT
f=$(ls /tmp/prof_s/*/*kernel_stats.csv | head -1); cp $f gpurun_out/j18_kernel_stats.csv; head -30 gpurun_out/j18_kernel_stats.csv | cut -c1-230
