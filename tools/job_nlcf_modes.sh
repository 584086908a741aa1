#!/bin/bash
for m in 0 1; do echo "== ZKW_NLCF_MODE=$m"; ZKW_NLCF_MODE=$m timeout -s KILL 300 python tools/probe_netlist_perf.py 2>&1 | grep -v amdgpu.ids | cut -c1-330; done
