#!/bin/bash
mkdir -p gpurun_out
export ZKW_BATCH_LOG=1 ZKW_BLOCK_MEM_LOG=1
timeout 600 python tools/probe_blocks_pipeline.py 512 2 seq device > gpurun_out/j3_512_seq_dev.txt 2>&1; tail -6 gpurun_out/j3_512_seq_dev.txt
timeout 600 python tools/probe_blocks_pipeline.py 512 2 seq host > gpurun_out/j3_512_seq_host.txt 2>&1; tail -3 gpurun_out/j3_512_seq_host.txt
timeout 600 python tools/probe_blocks_pipeline.py 256 4 overlap device > gpurun_out/j3_256_ovl_dev.txt 2>&1; tail -6 gpurun_out/j3_256_ovl_dev.txt
