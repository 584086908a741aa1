#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -8
export ZKW_BATCH_LOG=1
timeout 600 python tools/probe_blocks_pipeline.py 256 2 seq device > gpurun_out/j8_256_seq.txt 2>&1; grep -v "zkw batch" gpurun_out/j8_256_seq.txt | tail -3; grep "zkw batch" gpurun_out/j8_256_seq.txt | tail -4
