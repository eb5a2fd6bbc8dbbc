#!/bin/bash
# Round 4, visit 12: the f16-piece products with THREE MFMAs (default build: hh, hm, mh) against the four-MFMA build (-DDC_X2H_KEEP_MM):
# parity tests on both, error against f64 of both on the network's GEMM shapes, and the default bench alternating between the two.
TAG=${1:-r4v12}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
MM4=$(pwd)/dotaclient_amd/libdotaclient_hip_mm4.so
for lib in "" mm4; do
  L=""; [ -n "$lib" ] && L=$MM4
  DC_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest_$lib.log 2>&1
  echo "lib '$lib' pytest exit $?"; tail -2 $OUT/pytest_$lib.log
  cp gpurun_out/elementwise_parity.json $OUT/elementwise_parity_$lib.json 2>/dev/null
  DC_LIB=$L timeout 300 python tools/gemm_bench.py 65536 256 > $OUT/gemm_bench_$lib.txt 2>&1
  grep -c . $OUT/gemm_bench_$lib.txt
done
for rep in 1 2 3; do
for lib in "" mm4; do
  L=""; [ -n "$lib" ] && L=$MM4
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${lib}_$rep.json') if l.startswith('{')][0])
    print('lib %-4s rep $rep: %.1f env-steps/s %.3f ms/step parity %s' % ('$lib', j['value'], j['ms_per_step'], j.get('parity')))
except Exception as e:
    print('bench failed', '$lib', e); print(open('$OUT/bench_${lib}_$rep.err').read()[-800:])
PY
done
done
ls $OUT | head -3
