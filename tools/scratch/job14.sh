#!/bin/bash
mkdir -p gpurun_out
for cfg in "2 16" "3 16" "3 32" "4 16"; do set -- $cfg
ZKW_SYNTH_THREADS=$1 EC_CHUNK=$2 timeout 600 python tools/probe_blocks_pipeline.py 512 2 seq device > gpurun_out/j14_$1_$2.txt 2>&1; echo "threads $1 ec_chunk $2"; tail -3 gpurun_out/j14_$1_$2.txt
done
