# A/B of two libzkw builds on one box: bash tools/job_ab.sh <alt name> [steps]
alt=$1; steps=${2:-3}
mkdir -p gpurun_out/r05
for lib in base $alt base $alt; do
  if [ $lib = base ]; then unset ZKW_LIB; else export ZKW_LIB=$PWD/era_zkevm_test_harness_amd/_alt/libzkw_$lib.so; fi
  python bench.py --steps $steps --warmup 1 --no-full-block --no-cpu-baseline --no-h2d > gpurun_out/r05/ab_$lib.json 2> gpurun_out/r05/ab_$lib.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r05/ab_$lib.json"))
k=d["kernels_ms_per_step"]
print("$lib", round(d["value"],1), round(d["ms_per_step"]), d["validation"]["ok"], {x:round(k[x]) for x in list(k)[:8]})
PY
done
