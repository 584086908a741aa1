#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -4
export ZKW_BATCH_LOG=1
timeout 600 python tools/probe_blocks_pipeline.py 256 2 seq device > gpurun_out/j9_256_seq.txt 2>&1; grep -v "zkw batch" gpurun_out/j9_256_seq.txt | tail -3; grep "zkw batch" gpurun_out/j9_256_seq.txt | tail -4
unset ZKW_BATCH_LOG
timeout 600 python tools/probe_blocks_pipeline.py 512 2 seq device > gpurun_out/j9_512_seq.txt 2>&1; tail -3 gpurun_out/j9_512_seq.txt
for T in 2 4; do ZKW_SYNTH_THREADS=$T timeout 600 python tools/probe_blocks_pipeline.py 256 2 seq device 2>&1 | tail -2; done
ZKW_SYNTH_GROUP=32 timeout 600 python tools/probe_blocks_pipeline.py 256 2 seq device 2>&1 | tail -2
