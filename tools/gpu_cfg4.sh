#!/bin/bash
# BASELINE.json configs[4] per-GPU shard on one MI355X: 2-layer LSTM-512, 256 trajectories x 512 steps, bf16 MFMA path:
# persistent team kernel (flags 4096) vs launch-per-step bf16 kernels (4096 + 65536)
OUT=gpurun_out/${1:-cfg4}
mkdir -p $OUT
for MODE in 4096 69632; do
  timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --kernel-flags $MODE > $OUT/bench_flags$MODE.json 2> $OUT/bench_flags$MODE.err
  python - <<PY
import json
try:
    j = json.load(open('$OUT/bench_flags$MODE.json'))
    print('flags $MODE:', j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step', j['dtype'], 'status', j['nan_status'])
    for k in j['roofline']['kernels'][:6]:
        print('   ', k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'])
except Exception as e:
    print('flags $MODE failed:', e); print(open('$OUT/bench_flags$MODE.err').read()[-800:])
PY
done
