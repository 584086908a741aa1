#!/bin/bash
mkdir -p gpurun_out
export ZKW_SYNTH_LOG=1
for cfg in "3 16 2" "3 32 2" "3 16 1"; do set -- $cfg
ZKW_SYNTH_THREADS=$1 EC_CHUNK=$2 ZKW_EC_THREADS=$3 timeout 600 python tools/probe_blocks_pipeline.py 512 2 seq device > gpurun_out/j16_$1_$2_$3.txt 2>&1; echo "threads $1 ec_chunk $2 ec_threads $3"; tail -7 gpurun_out/j16_$1_$2_$3.txt
done
timeout 900 python -m pytest tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -3
