#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/probe_blocks_pipeline.py 512 2 seq device > gpurun_out/j7_512_seq.txt 2>&1; tail -3 gpurun_out/j7_512_seq.txt
for T in 4 6; do
ZKW_SYNTH_THREADS=$T timeout 600 python tools/probe_blocks_pipeline.py 256 2 seq device > gpurun_out/j7_256_seq_t$T.txt 2>&1; echo "threads $T"; tail -2 gpurun_out/j7_256_seq_t$T.txt
done
timeout 600 python tools/probe_blocks_pipeline.py 256 5 overlap device > gpurun_out/j7_256_ovl.txt 2>&1; tail -6 gpurun_out/j7_256_ovl.txt
timeout 900 python -m pytest tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -3
