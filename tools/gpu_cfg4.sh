#!/bin/bash
# configs[4]'s per-GPU shard (2-layer LSTM-512, 256 x 512, DC_DIMS_BF16): the bf16-path tests, then a same-box A/B of bf16 storage of
# the gate buffers (default) against f32 storage (DC_DIMS_BF16_F32_STORE), alternating, two repetitions
# usage: bash tools/gpu_cfg4.sh <tag>      (TESTS=0 skips the tests)
OUT=gpurun_out/${1:-r5c4}; mkdir -p $OUT; export TMPDIR=/tmp
if [ "${TESTS:-1}" != "0" ]; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16.py -m gpu -q -x -s -k "bf16" 2>&1 | grep -v "^$" | tail -${TAIL:-25} | cut -c1-400
fi
CFG4="--cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --no-cpu-baseline --no-weak-unit --no-secondary"
for rep in 1 2; do
for mode in bf16 f32; do
  F=4096; [ "$mode" = "f32" ] && F=$((4096 + 4194304))
  timeout 300 python bench.py --steps 10 --warmup 3 $CFG4 --kernel-flags $F > $OUT/cfg4_${mode}_$rep.json 2> $OUT/cfg4_${mode}_$rep.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/cfg4_${mode}_$rep.json') if l.startswith('{')][0])
    ks = {k['kernel'].split('(')[0]: (k['avg_us'], k['ms_per_step']) for k in j['roofline']['kernels']}
    print('storage %-4s rep $rep: %.1f env-steps/s %.2f ms/step  ' % ('$mode', j['value'], j['ms_per_step']) +
          ' '.join('%s=%.0f(%.1fms)' % (k, ks[k][0], ks[k][1]) for k in ('lstm_fwd_team', 'lstm_bwd_team', 'gemm_f32_fwd', 'gemm_f32_dX', 'gemm_f32_dW') if k in ks))
except Exception as e:
    print('storage $mode rep $rep failed', e); print(open('$OUT/cfg4_${mode}_$rep.err').read()[-1500:])
PY
done
done
