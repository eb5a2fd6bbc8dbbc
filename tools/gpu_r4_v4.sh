#!/bin/bash
# Round 4, visit 4: fence-free fused loss / Adam chains, f16 pieces also in the fused embedding kernels: targeted tests, bench A/B, full suite.
TAG=${1:-r4v4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_targeted.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_targeted.log; tail -5 $OUT/pytest_targeted.log
for fl in 0 131072 0 131072; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $fl > $OUT/bench_flags_$fl.json 2> $OUT/bench_flags_$fl.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_flags_$fl.json') if l.startswith('{')][0])
    print('flags $fl:', j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step')
    for k in j['roofline']['kernels']:
        print('   %-32s n=%3d avg=%8.1f us %6.3f ms  %s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k.get('achieved_tflops') or k.get('achieved_gbs')))
except Exception as e:
    print('bench failed', e); print(open('$OUT/bench_flags_$fl.err').read()[-1500:])
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --kernel-flags 131072 > $OUT/bench_f16x2_full.json 2> $OUT/bench_f16x2_full.err
python - <<PY
import json
j = json.loads([l for l in open('$OUT/bench_f16x2_full.json') if l.startswith('{')][0])
print('f16x2 full:', j['value'], j['ms_per_step'], 'parity', j['parity']['ok'], j['parity']['parity_rel_err'], j['parity']['per_quantity'])
PY
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
ls $OUT
