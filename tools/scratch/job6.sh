#!/bin/bash
mkdir -p gpurun_out
for T in 16 32; do
ZKW_SYNTH_THREADS=$T timeout 600 python tools/probe_blocks_pipeline.py 256 2 seq device > gpurun_out/j6_256_seq_t$T.txt 2>&1; echo "threads $T"; tail -3 gpurun_out/j6_256_seq_t$T.txt
done
ZKW_SYNTH_THREADS=16 timeout 600 python tools/probe_blocks_pipeline.py 256 5 overlap device > gpurun_out/j6_256_ovl_t16.txt 2>&1; tail -6 gpurun_out/j6_256_ovl_t16.txt
