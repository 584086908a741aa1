# usage: bash tools/job_bench.sh <tag> [bench args...]   -> gpurun_out/r05/bench_<tag>.json
tag=$1; shift
mkdir -p gpurun_out/r05
python -X faulthandler bench.py "$@" > gpurun_out/r05/bench_$tag.json 2> gpurun_out/r05/bench_$tag.err
echo "bench rc=$?"
grep -v "amdgpu.ids" gpurun_out/r05/bench_$tag.err | tail -25
python - <<PY
import json
d=json.load(open("gpurun_out/r05/bench_$tag.json"))
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("validation")))
r=d["roofline_valu"]; print(json.dumps(r["step"])[:400])
print({k:(round(v["frac"],3),round(v["ms_per_step"])) for k,v in r["per_kernel"].items()})
fb=d.get("full_block") or {}
print("full_block wall", fb.get("wall_ms"), json.dumps(fb.get("batched"))[:600])
print(json.dumps({k:round(v["circuits_per_s"]) for k,v in (d.get("hash_circuits") or {}).items()}))
PY
