#!/bin/bash
# Whole GPU suite + the driver's bench command (what the driver runs at round end)
OUT=gpurun_out/${1:-r5full}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<PY
import json
j = json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], 'parity', j['parity']['ok'], j['parity']['parity_rel_err'], j['parity']['per_quantity'])
for k,v in (j.get('secondary') or {}).items():
    print(k, v.get('value'), v.get('ms_per_step'), v.get('parity',{}).get('ok'), v.get('parity',{}).get('parity_rel_err'), v.get('error'))
fb = j.get('products_fallback') or {}
print('fallback', fb.get('value'), fb.get('ms_per_step'), (fb.get('parity') or {}).get('ok'))
PY
