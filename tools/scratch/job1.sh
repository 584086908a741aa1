#!/bin/bash
# round 6, first contact: the converted kernels / own radix sort under the existing parity tests, then blocks in flight on the batch
mkdir -p gpurun_out
export ZKW_BATCH_LOG=1
timeout 900 python -m pytest tests/test_gpu_ram_path.py tests/test_gpu_block.py -x -q -m gpu > gpurun_out/j1_tests.txt 2>&1
tail -5 gpurun_out/j1_tests.txt
for K in 1 8 64; do timeout 300 python tools/probe_block_concurrency.py $K 2 > gpurun_out/j1_probe_$K.txt 2>&1; tail -4 gpurun_out/j1_probe_$K.txt; done
