#!/bin/bash
# BASELINE.json configs[4] per-GPU shard on one MI355X: 2-layer LSTM-512, 256 trajectories x 512 steps, bf16 MFMA path
# (DC_DIMS_BF16 = 4096) and, beside it, the same shard f32-grade.  Usage: bash tools/gpu_cfg4.sh <tag>
TAG=${1:-cfg4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for MODE in 4096 0; do
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --kernel-flags $MODE > $OUT/bench_flags$MODE.json 2> $OUT/bench_flags$MODE.err
  python - <<PY
import json
try:
    j = json.load(open('$OUT/bench_flags$MODE.json'))
    print('flags $MODE:', j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step', j['dtype'])
    for k in j['roofline']['kernels'][:10]:
        print('   %-30s n=%4d avg=%9.1f us  %8.3f ms' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step']))
except Exception as e:
    print('bench failed', e); print(open('$OUT/bench_flags$MODE.err').read()[-1500:])
PY
done
