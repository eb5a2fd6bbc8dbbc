#!/bin/bash
# Round 4, visit 3: fused loss / Adam chains + the two-f16-piece products (DC_DIMS_F16X2): targeted tests, GEMM A/B, bench A/B on one box.
TAG=${1:-r4v3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_targeted.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_targeted.log; tail -5 $OUT/pytest_targeted.log
timeout 300 python tools/gemm_bench.py 65536 256 > $OUT/gemm_bench_f16x2.txt 2>&1; cat $OUT/gemm_bench_f16x2.txt | cut -c1-330
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
timeout 600 python bench.py --steps 20 --warmup 5 --kernel-flags 131072 > $OUT/bench_f16x2_full.json 2> $OUT/bench_f16x2_full.err; tail -c 1500 $OUT/bench_f16x2_full.json
ls $OUT
