#!/bin/bash
# Round 4, visit 8: variants of the sixteen-wave sparse backward (A/B libraries): fold at the end of the iteration, pair fold - correctness
# (sparse vs dense vs eight-wave test, degenerate arg-max test) and time per launch next to the eight-wave default, one box.
TAG=${1:-r4v8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for lib in "" p16e p16p; do
  L=""; [ -n "$lib" ] && L=$(pwd)/dotaclient_amd/libdotaclient_hip_$lib.so
  DC_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sparse_pool or golden" > $OUT/pytest_$lib.log 2>&1
  echo "lib '$lib' pytest exit $?"; tail -2 $OUT/pytest_$lib.log
done
for rep in 1 2; do
for cfg in ":0" ":262144" "p16e:262144" "p16p:262144"; do
  lib=${cfg%%:*}; fl=${cfg##*:}
  L=""; [ -n "$lib" ] && L=$(pwd)/dotaclient_amd/libdotaclient_hip_$lib.so
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $fl > $OUT/bench_${lib}_$fl.json 2> $OUT/bench_${lib}_$fl.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${lib}_$fl.json') if l.startswith('{')][0])
    k = [x for x in j['roofline']['kernels'] if x['kernel'] == 'embed_bwd_pool16'][0]
    print('lib %-5s flags %-7s: %.1f env-steps/s %.3f ms/step  embed_bwd_pool16 %.1f us' % ('$lib', '$fl', j['value'], j['ms_per_step'], k['avg_us']))
except Exception as e:
    print('bench failed', '$lib', '$fl', e); print(open('$OUT/bench_${lib}_$fl.err').read()[-800:])
PY
done
done
ls $OUT
