#!/bin/bash
mkdir -p gpurun_out
export ZKW_BATCH_LOG=1 ZKW_BLOCK_MEM_LOG=1
timeout 600 python tools/probe_blocks_pipeline.py 256 2 seq device > gpurun_out/j4_256_seq.txt 2>&1; tail -8 gpurun_out/j4_256_seq.txt
unset ZKW_BATCH_LOG ZKW_BLOCK_MEM_LOG
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/j4_tests.txt 2>&1; tail -5 gpurun_out/j4_tests.txt
