#!/bin/bash
mkdir -p gpurun_out
ROOT=$(pwd)
export TMPDIR=/tmp
export ZKW_BATCH_LOG=1
cd /tmp && rm -rf /tmp/prof_b && timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $ROOT/tools/probe_blocks_pipeline.py 128 2 seq device > $ROOT/gpurun_out/j5_prof.txt 2>&1
cd $ROOT; grep -v rocprofv3 gpurun_out/j5_prof.txt | tail -5
f=$(ls /tmp/prof_b/*/*kernel_stats.csv | head -1); cp $f gpurun_out/j5_kernel_stats.csv; head -60 gpurun_out/j5_kernel_stats.csv | cut -c1-160
