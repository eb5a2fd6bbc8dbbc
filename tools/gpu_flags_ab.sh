#!/bin/bash
# Same-box A/B of dims flag sets (bench.py --kernel-flags): round-robin, REPS repetitions, selected regions.
# usage: FLAGS="0 33554432" [KEYS="embed_bwd_small embed_bwd_dw1 ..."] bash tools/gpu_flags_ab.sh <tag>
OUT=gpurun_out/${1:-flab}; mkdir -p $OUT; export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
for f in $FLAGS; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $f > $OUT/bench_${f}_$rep.json 2> $OUT/bench_${f}_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${f}_$rep.json') if l.startswith('{')][0])
    ks = {k['kernel'].split('(')[0]: k['avg_us'] for k in j['roofline']['kernels']}
    print('flags %-10s rep $rep: %.1f env-steps/s %.3f ms/step  ' % ('$f', j['value'], j['ms_per_step']) + ' '.join('%s=%.1f' % (k, ks.get(k, -1)) for k in '${KEYS:-embed_bwd_small embed_scatter_bwd embed_bwd_dw1 embed_bwd_dw2 embed_bwd_pool16m}'.split()))
except Exception as e:
    print('flags $f rep $rep failed', e); print(open('$OUT/bench_${f}_$rep.err').read()[-800:])
PY
done
done
