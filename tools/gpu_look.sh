#!/bin/bash
# Quick look - the pool16 variants' correctness test, the bench without the CPU legs, rocprofv3 kernel stats of the interesting kernels
OUT=gpurun_out/${1:-r5q}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sparse_pool" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
j = json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0])
print('%.1f env-steps/s %.3f ms/step' % (j['value'], j['ms_per_step']))
print(' '.join('%s=%.0f' % (k['kernel'].split('(')[0], k['avg_us']) for k in j['roofline']['kernels'][:12]))
PY
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-weak-unit ${BENCH_ARGS} > $OLDPWD/$OUT/prof.log 2>&1; cd $OLDPWD
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/kernel_stats.csv > /dev/null 2>&1; grep -i "${KGREP:-pool16m}" $OUT/kernel_stats.csv | cut -c1-150 | head -8
find $OUT/prof -name '*.db' -delete
