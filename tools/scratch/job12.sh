#!/bin/bash
mkdir -p gpurun_out
ROOT=$(pwd)
export TMPDIR=/tmp
ZKW_BLOCK_MEM_LOG=1 timeout 600 python tools/probe_blocks_pipeline.py 512 3 seq device > gpurun_out/j12_512_seq.txt 2>&1; grep -v "zkw blocks" gpurun_out/j12_512_seq.txt | tail -4; grep "zkw blocks" gpurun_out/j12_512_seq.txt | tail -2
cd /tmp && rm -rf /tmp/prof_b && timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python $ROOT/tools/probe_blocks_pipeline.py 256 2 seq device > $ROOT/gpurun_out/j12_prof.txt 2>&1
cd $ROOT; grep -v rocprofv3 gpurun_out/j12_prof.txt | tail -3
f=$(ls /tmp/prof_b/*/*kernel_stats.csv | head -1); cp $f gpurun_out/j12_kernel_stats.csv
