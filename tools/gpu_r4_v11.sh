#!/bin/bash
# Round 4, visit 11: team forward with the member's own units first (part 1 of the product under the peers' hand-off): parity, soak, timing.
TAG=${1:-r4v11}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_soak.py -m gpu -q -x > $OUT/pytest_targeted.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_targeted.log; tail -4 $OUT/pytest_targeted.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_$rep.json') if l.startswith('{')][0])
    ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
    print('run $rep: %.1f env-steps/s %.3f ms/step  lstm_fwd_team %.1f us  lstm_bwd_team %.1f us  ratio %.3f' % (j['value'], j['ms_per_step'], ks['lstm_fwd_team'], ks['lstm_bwd_team'], ks['lstm_fwd_team'] / ks['lstm_bwd_team']))
except Exception as e:
    print('bench failed', e); print(open('$OUT/bench_$rep.err').read()[-1200:])
PY
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --cell gru --hidden 256 > $OUT/bench_gru.json 2> $OUT/bench_gru.err
python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_gru.json') if l.startswith('{')][0])
    ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
    print('gru-256 256x256: %.1f env-steps/s %.3f ms/step  gru_fwd_team %.1f us  gru_bwd_team %.1f us' % (j['value'], j['ms_per_step'], ks.get('gru_fwd_team', 0), ks.get('gru_bwd_team', 0)))
except Exception as e:
    print('gru bench failed', e)
PY
ls $OUT | head -3
