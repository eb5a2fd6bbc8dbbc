#!/bin/bash
# Short GPU-box visit: the GPU suite, then a short bench line (no CPU baseline) with its per-region table.
# Usage: bash tools/gpu_quick.sh <tag> [extra bench args]
TAG=${1:-q}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
try:
    j = json.load(open('$OUT/bench.json'))
    print(j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step')
    tot = 0
    for k in j['roofline']['kernels']:
        print('%-28s n=%3d avg=%8.1f us  %6.3f ms  %6.1f TF' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k.get('achieved_tflops') or 0.0)); tot += k['ms_per_step']
    print('profiled regions: %.3f ms' % tot)
    print('secondary', json.dumps(j.get('secondary')))
    print('epoch_graph', json.dumps(j.get('epoch_graph')))
except Exception as e:
    print('bench failed', e); print(open('$OUT/bench.err').read()[-2000:])
PY
