#!/bin/bash
# Round 4, visit 17: the f16x2 split-on-load GEMM with THREE workgroups per CU (48 KB of stage buffers, half-size epilogue image; default
# build) against two (-DDC_X3_OCC=2): parity tests on the default, gemm_bench and the default bench alternating between the two.
TAG=${1:-r4v17}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ALT=$(pwd)/dotaclient_amd/libdotaclient_hip_${2:-occ2}.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -2 $OUT/pytest.log
for lib in "" alt; do
  L=""; [ -n "$lib" ] && L=$ALT
  DC_LIB=$L timeout 300 python tools/gemm_bench.py 65536 256 > $OUT/gemm_bench_$lib.txt 2>&1
done
python - <<PY
import re
a = open('$OUT/gemm_bench_.txt').read().splitlines(); b = open('$OUT/gemm_bench_alt.txt').read().splitlines()
for x, y in zip(a, b):
    mx = re.search(r'f16x2\s+([\d.]+) us', x); my = re.search(r'f16x2\s+([\d.]+) us', y)
    if mx and my: print('%-24s default %8s us   alt %8s us' % (x[:24], mx.group(1), my.group(1)))
PY
for rep in 1 2 3; do
for lib in "" alt; do
  L=""; [ -n "$lib" ] && L=$ALT
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${lib}_$rep.json') if l.startswith('{')][0])
    ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
    print('lib %-4s rep $rep: %.1f env-steps/s %.3f ms/step  dW %.1f fwd %.1f dX %.1f us' % ('$lib', j['value'], j['ms_per_step'], ks.get('gemm_f32_dW(TN,split-K)', 0), ks.get('gemm_f32_fwd(NT)', 0), ks.get('gemm_f32_dX(NN)', 0)))
except Exception as e:
    print('bench failed', '$lib', e); print(open('$OUT/bench_${lib}_$rep.err').read()[-800:])
PY
done
done
