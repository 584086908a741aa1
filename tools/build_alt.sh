#!/bin/bash
# A/B builds of libzkw.so with extra -D flags: tools/build_alt.sh <name> <flags...>  ->  era_zkevm_test_harness_amd/_alt/libzkw_<name>.so
# (bench.py / the tests load it with ZKW_LIB=<path>)
set -e
name=$1; shift
cd "$(dirname "$0")/../era_zkevm_test_harness_amd"
mkdir -p _alt/_obj_$name
objs=""
for f in zkw_api zkw_sorters zkw_precompiles zkw_setup zkw_block zkw_recursion zkw_comm zkw_vm_trace zkw_dispatch zkw_commit sort; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c csrc/$f.hip -o _alt/_obj_$name/$f.o &
  objs="$objs _alt/_obj_$name/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _alt/libzkw_$name.so $objs
echo _alt/libzkw_$name.so
