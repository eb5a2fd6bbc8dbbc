#!/bin/bash
# Same-box A/B of two builds of the library (DC_LIB): bench without the CPU legs, alternating, two repetitions
# usage: ALT=dotaclient_amd/libdotaclient_hip_<variant>.so [KEYS="embed_bwd_pool16m ..."] bash tools/gpu_ab.sh <tag>
OUT=gpurun_out/${1:-r5ab}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "${TESTS:-sparse_pool}" 2>&1 | tail -2
for rep in 1 2; do
for lib in "" alt; do
  L=""; [ -n "$lib" ] && L=$(pwd)/$ALT
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit ${BENCH_ARGS} > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${lib}_$rep.json') if l.startswith('{')][0])
    ks = {k['kernel'].split('(')[0]: k['avg_us'] for k in j['roofline']['kernels']}
    print('lib %-4s rep $rep: %.1f env-steps/s %.3f ms/step  ' % ('$lib' or 'dflt', j['value'], j['ms_per_step']) + ' '.join('%s=%.1f' % (k, ks.get(k, -1)) for k in '${KEYS:-embed_bwd_pool16m embed_fwd_fused lstm_fwd_team}'.split()))
except Exception as e:
    print('lib $lib rep $rep failed', e); print(open('$OUT/bench_${lib}_$rep.err').read()[-800:])
PY
done
done
