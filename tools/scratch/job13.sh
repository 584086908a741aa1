#!/bin/bash
mkdir -p gpurun_out
ZKW_BLOCK_MEM_LOG=1 timeout 600 python tools/probe_blocks_pipeline.py 512 3 seq device > gpurun_out/j13_512_seq.txt 2>&1; grep -v "zkw blocks" gpurun_out/j13_512_seq.txt | tail -4; grep "zkw blocks" gpurun_out/j13_512_seq.txt | tail -2
timeout 600 python tools/probe_blocks_pipeline.py 512 4 seqbg device > gpurun_out/j13_512_seqbg.txt 2>&1; tail -5 gpurun_out/j13_512_seqbg.txt
