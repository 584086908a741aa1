mkdir -p gpurun_out/r05
timeout 300 tools/ubench_glmul > gpurun_out/r05/glmul.json 2> gpurun_out/r05/glmul.err; cat gpurun_out/r05/glmul.err gpurun_out/r05/glmul.json
timeout 600 tools/ubench_valu_ceiling > gpurun_out/r05/valu_ceiling.json 2> gpurun_out/r05/valu_ceiling.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/valu_ceiling.json'))
for c in d['classes']:
    print(c['class'], {w:(round(v.get('cycles_per_wave_inst_per_simd_at_2.4GHz',0),2) or '%.3e'%v['units_per_s']) for w,v in c['by_waves_per_simd'].items()})
PY
timeout 900 python -m pytest tests -m gpu -x -q -k "slot_reuse or ram_synthesis or closed_forms" 2>&1 | tail -5
