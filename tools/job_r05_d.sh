mkdir -p gpurun_out/r05
for n in a b c; do for m in 33 34; do echo "variant $n mode $m"; tools/ubench_valu_ceiling_$n $m | python -c "
import sys,json
d=json.load(sys.stdin)
for c in d['classes']: print(c['class'], {w:'%.3e'%v['units_per_s'] for w,v in c['by_waves_per_simd'].items()})"; done; done
tools/ubench_glmul > gpurun_out/r05/glmul2.json 2> gpurun_out/r05/glmul2.err; cat gpurun_out/r05/glmul2.err
