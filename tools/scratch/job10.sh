#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/probe_blocks_pipeline.py 256 6 overlap device > gpurun_out/j10_256_ovl.txt 2>&1; tail -8 gpurun_out/j10_256_ovl.txt
timeout 600 python tools/probe_blocks_pipeline.py 320 5 overlap device > gpurun_out/j10_320_ovl.txt 2>&1; tail -3 gpurun_out/j10_320_ovl.txt
