#!/bin/bash
# Short GPU-box visit: the LSTM variant tests first, an A/B bench of the two persistent-LSTM variants, then
# the whole GPU suite.  Usage: bash tools/gpu_ab.sh <tag>
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "persist_variants" > $OUT/t_variants.log 2>&1
echo "variants exit $?"; tail -15 $OUT/t_variants.log
for v in mfma valu; do
  DC_LSTM_PERSIST=$v timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
try:
    j = json.load(open('$OUT/bench_$v.json'))
    print('$v', j['value'], j['ms_per_step'], [(k['kernel'], k['avg_us']) for k in j['roofline']['kernels'][:6]])
except Exception as e:
    print('$v failed', e); print(open('$OUT/bench_$v.err').read()[-1500:])
PY
done
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/pytest_gpu.log
