#!/bin/bash
# Round 5, visit 2: the matrix-core max-pool backward (embed_pool16m.hip) - correctness against the dense kernels, then A/B timing against
# the sparse VALU kernel (DC_DIMS_POOL16_VALU = 2097152); the range-edge tests after the NaN-propagating relu / max-pool.
OUT=gpurun_out/r5v2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sparse_pool or range_edge or out_of_range" > $OUT/pytest_pool.log 2>&1; tail -25 $OUT/pytest_pool.log
for rep in 1 2; do
for flags in 0 2097152; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $flags > $OUT/bench_${flags}_$rep.json 2> $OUT/bench_${flags}_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${flags}_$rep.json') if l.startswith('{')][0])
    ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
    print('flags %-8s rep $rep: %.1f env-steps/s %.3f ms/step  pool16 %.1f us  fwd %.1f us' % ('$flags', j['value'], j['ms_per_step'], ks.get('embed_bwd_pool16', -1), ks.get('embed_fwd_fused', -1)))
except Exception as e:
    print('flags $flags rep $rep failed', e); print(open('$OUT/bench_${flags}_$rep.err').read()[-1500:])
PY
done
done
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -8 $OUT/pytest_gpu.log
