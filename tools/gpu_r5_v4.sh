#!/bin/bash
# Round 5, visit 4: embed_pool16m.hip as two kernels (dW2 with k-quarter waves; d(basic) -> dW1 with whole pairs per wave): gradients vs the dense kernels, timing, tests
OUT=gpurun_out/r5v4; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/pool16_debug.py 2>&1 | grep "scaled err" | cut -c1-120 | head -6
for flags in 0 2097152; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $flags > $OUT/bench_${flags}.json 2> $OUT/bench_${flags}.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${flags}.json') if l.startswith('{')][0])
    ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
    print('flags %-8s: %.1f env-steps/s %.3f ms/step  pool16 %.1f us  fwd %.1f us' % ('$flags', j['value'], j['ms_per_step'], ks.get('embed_bwd_pool16', -1), ks.get('embed_fwd_fused', -1)))
except Exception as e:
    print('flags $flags failed', e); print(open('$OUT/bench_${flags}.err').read()[-1500:])
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sparse_pool" > $OUT/pytest_sel.log 2>&1; tail -5 $OUT/pytest_sel.log
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-weak-unit > $OLDPWD/$OUT/prof.log 2>&1; cd $OLDPWD
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/kernel_stats.csv > /dev/null 2>&1; grep -i "pool16\|gemm_fast\|gemm_kernel" $OUT/kernel_stats.csv | head -8
find $OUT/prof -name '*.db' -delete
