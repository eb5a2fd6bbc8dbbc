#!/bin/bash
# Round 4, visit 14: the reworked loss kernel (row leaders accumulate in LDS, f64 divisions hoisted) and Adam update (1024-element blocks):
# full GPU suite, then the default bench twice; the regions' times against visit 10's (ppo_loss 65-90 us, gradnorm_clip_adam 45-55 us).
TAG=${1:-r4v14}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_$rep.json') if l.startswith('{')][0])
    print('rep $rep: %.1f env-steps/s %.3f ms/step' % (j['value'], j['ms_per_step']))
    for k in j['roofline']['kernels']:
        if k['kernel'].startswith(('ppo_loss', 'gradnorm', 'embed_bwd_pool16', 'lstm')):
            print('   %-34s n=%3d avg %8.1f us' % (k['kernel'], k['launches_per_step'], k['avg_us']))
except Exception as e:
    print('bench failed', e); print(open('$OUT/bench_$rep.err').read()[-800:])
PY
done
