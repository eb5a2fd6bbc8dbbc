#!/bin/bash
# Round 5, visit 4: where a pair's cycles go in embed_pool16m.hip (A/B build -DDC_PM_TIMING: s_memtime sums per phase), then timing + tests
OUT=gpurun_out/r5v4; mkdir -p $OUT; export TMPDIR=/tmp
DC_LIB=$(pwd)/dotaclient_amd/libdotaclient_hip_pmtime.so timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-weak-unit --no-secondary > $OUT/timing.json 2> $OUT/timing.err
grep -h "pool16m wave" $OUT/timing.json $OUT/timing.err | tail -8 | head -3
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
