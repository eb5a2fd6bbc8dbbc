#!/bin/bash
# Round 4, visit 7: the sixteen-wave sparse max-pool backward: targeted tests (parity incl. sparse vs dense vs eight-wave, guard / soak), bench A/B
# (default vs DC_DIMS_POOL16_8W), the three-product f16 variant (A/B library) for the power-limit question.
TAG=${1:-r4v7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_targeted.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_targeted.log; tail -5 $OUT/pytest_targeted.log
for fl in 0 262144 0 262144; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $fl > $OUT/bench_flags_$fl.json 2> $OUT/bench_flags_$fl.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_flags_$fl.json') if l.startswith('{')][0])
    print('flags $fl:', j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step', [(k['kernel'], k['avg_us']) for k in j['roofline']['kernels'][:4]])
except Exception as e:
    print('bench failed', e); print(open('$OUT/bench_flags_$fl.err').read()[-1500:])
PY
done
DC_LIB=$(pwd)/dotaclient_amd/libdotaclient_hip_x2h3.so timeout 300 python bench.py --steps 20 --warmup 5 --no-weak-unit > $OUT/bench_x2h3.json 2> $OUT/bench_x2h3.err
python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_x2h3.json') if l.startswith('{')][0])
    print('three-product variant:', j['value'], j['ms_per_step'], 'parity', j['parity']['ok'], j['parity']['parity_rel_err'])
    for k in j['roofline']['kernels'][:9]:
        print('   %-32s n=%3d avg=%8.1f us %6.3f ms  %s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k.get('achieved_tflops') or k.get('achieved_gbs')))
except Exception as e:
    print('x2h3 failed', e); print(open('$OUT/bench_x2h3.err').read()[-1500:])
PY
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log
ls $OUT
