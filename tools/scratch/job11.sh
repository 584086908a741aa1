#!/bin/bash
mkdir -p gpurun_out
ZKW_CHAIN_WG4=0 timeout 600 python tools/probe_blocks_pipeline.py 256 5 overlap device > gpurun_out/j11_256_ovl_nowg4.txt 2>&1; tail -7 gpurun_out/j11_256_ovl_nowg4.txt
ZKW_CHAIN_WG4=0 timeout 600 python tools/probe_blocks_pipeline.py 512 2 seq device > gpurun_out/j11_512_seq_nowg4.txt 2>&1; tail -3 gpurun_out/j11_512_seq_nowg4.txt
