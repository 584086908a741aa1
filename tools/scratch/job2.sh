#!/bin/bash
mkdir -p gpurun_out
export ZKW_BATCH_LOG=1
for K in 128 256 512; do timeout 600 python tools/probe_block_concurrency.py $K 2 > gpurun_out/j2_probe_$K.txt 2>&1; tail -4 gpurun_out/j2_probe_$K.txt; done
