python -m pytest tests/test_gpu_ram_path.py tests/test_gpu_reference_kats.py -m gpu -x -q 2>&1 | tail -2
A="--no-cpu-baseline --no-full-block --no-h2d --no-hash-circuits"
for i in 1 2; do python bench.py $A > gpurun_out/q4p_$i.json 2>/dev/null; done
python - <<'PY'
import json
for i in ['1','2']:
    d=json.loads(open('gpurun_out/q4p_%s.json'%i).read().strip().split('\n')[-1])
    sp=d['pass_spans_ms']
    print(i, round(d['value']), round(d['ms_per_step']), 'B', [round(s[2]-s[1]) for s in sp][:10], 'S', [round(s[-2]-s[-3]) for s in sp][:10], {k:round(v) for k,v in list(d['kernels_ms_per_step'].items())[:3]})
PY
