for n in c d; do for m in 34; do echo "variant $n mode $m"; tools/ubench_valu_ceiling_$n $m | python -c "
import sys,json
d=json.load(sys.stdin)
for c in d['classes']: print(c['class'], {w:'%.3e'%v['units_per_s'] for w,v in c['by_waves_per_simd'].items()})"; done; done
