#!/bin/bash
# Round 5, visit 3: embed_pool16m.hip with wave-private LDS staging - where do its gradients differ from the dense kernels', and how fast is it
OUT=gpurun_out/r5v3; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/pool16_debug.py > $OUT/pool16_debug.txt 2>&1; cat $OUT/pool16_debug.txt | cut -c1-330
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
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -k "sparse_pool or range_edge or out_of_range or nan_recovery or publish" > $OUT/pytest_sel.log 2>&1; tail -12 $OUT/pytest_sel.log
