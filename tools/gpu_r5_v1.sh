#!/bin/bash
# Round 5, visit 1b: the range-edge test of the fused embedding first layer and the driver's bench command with the new blocks.
OUT=gpurun_out/r5v1; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "range_edge or out_of_range" > $OUT/pytest_edge.log 2>&1; tail -30 $OUT/pytest_edge.log
T0=$(date +%s)
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? wall $(( $(date +%s) - T0 )) s"; tail -5 $OUT/bench.err
python - <<PY
import json
j = json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['parity']['ok'], j['parity']['parity_rel_err'])
print(json.dumps(j.get('secondary'), indent=0)[:6000])
print(json.dumps(j.get('products_fallback'), indent=0)[:1500])
PY
