#!/bin/bash
# Round 5, visit 1: the new kernel-level tests of the default arithmetic (prec 4, range edges), the whole GPU suite, and the driver's
# bench command with the new `secondary` / `products_fallback` blocks.
OUT=gpurun_out/r5v1; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "x3" > $OUT/pytest_x3.log 2>&1; tail -15 $OUT/pytest_x3.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "range_edge or out_of_range" > $OUT/pytest_edge.log 2>&1; tail -30 $OUT/pytest_edge.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
/usr/bin/time -v timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -5 $OUT/bench.err
python - <<PY
import json
j = json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['parity']['ok'], j['parity']['parity_rel_err'])
print(json.dumps(j.get('secondary'), indent=0)[:3000])
print(json.dumps(j.get('products_fallback'), indent=0)[:1500])
PY
