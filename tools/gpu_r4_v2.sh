#!/bin/bash
# Round 4, visit 2: full GPU suite, the f16-piece GEMM probe, configs[4]'s shard measured the driver's way (bf16 path, parity on a 64-trajectory sub-batch).
TAG=${1:-r4v2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
cp gpurun_out/elementwise_parity.json $OUT/ 2>/dev/null
timeout 300 tools/ubench/gemm_x3 > $OUT/ubench_gemm_f16_pieces.txt 2>&1; cat $OUT/ubench_gemm_f16_pieces.txt
timeout 900 python3 bench.py --cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --kernel-flags 4096 --steps 20 --warmup 5 > $OUT/cfg4_bench.json 2> $OUT/cfg4_bench.err
echo "cfg4 bench exit $?"; tail -3 $OUT/cfg4_bench.err
python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/cfg4_bench.json') if l.startswith('{')][0])
    print(j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step', 'parity', j['parity'] and (j['parity']['ok'], j['parity']['parity_rel_err'], j['parity']['argmax_equal_fraction']), 'cpu', j['cpu_baseline'] and j['cpu_baseline']['value'])
    for k in j['roofline']['kernels']:
        print('%-32s n=%3d avg=%9.1f us  %7.3f ms  %s  frac %s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k.get('achieved_tflops') or k.get('achieved_gbs'), k.get('frac')))
except Exception as e:
    print('cfg4 bench failed', e)
PY
ls $OUT
