#!/bin/bash
# Same-box comparison of several builds of the library (DC_LIB=libdotaclient_hip_<variant>.so; "dflt" = the regular one): bench without the
# CPU legs, round-robin, REPS repetitions.  usage: VARIANTS="dflt a b" [KEYS="lstm_fwd_team ..."] [TESTS="-k expr"] bash tools/gpu_abn.sh <tag>
OUT=gpurun_out/${1:-abn}; mkdir -p $OUT; export TMPDIR=/tmp
[ -n "$TESTS" ] && { timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "$TESTS" 2>&1 | tail -2; }
for rep in $(seq 1 ${REPS:-2}); do
for v in $VARIANTS; do
  L=""; [ "$v" != dflt ] && L=$(pwd)/dotaclient_amd/libdotaclient_hip_$v.so
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit ${BENCH_ARGS} > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${v}_$rep.json') if l.startswith('{')][0])
    ks = {k['kernel'].split('(')[0]: k['avg_us'] for k in j['roofline']['kernels']}
    print('lib %-8s rep $rep: %.1f env-steps/s %.3f ms/step  ' % ('$v', j['value'], j['ms_per_step']) + ' '.join('%s=%.1f' % (k, ks.get(k, -1)) for k in '${KEYS:-lstm_fwd_team lstm_bwd_team}'.split()))
except Exception as e:
    print('lib $v rep $rep failed', e); print(open('$OUT/bench_${v}_$rep.err').read()[-800:])
PY
done
done
