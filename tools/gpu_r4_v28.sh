#!/bin/bash
# Round 4, visit 28: three GEMM workgroups per CU in bf16 mode too (configs[4]'s shard): the bf16 tests, then the shard's bench line.
OUT=gpurun_out/${1:-r4v28}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x > $OUT/pytest_bf16.log 2>&1
echo "pytest exit $?"; tail -2 $OUT/pytest_bf16.log
timeout 900 python3 bench.py --cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --kernel-flags 4096 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/cfg4_bench.json 2> $OUT/cfg4_bench.err
python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/cfg4_bench.json') if l.startswith('{')][0])
    print('cfg4', j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step')
    for k in j['roofline']['kernels'][:8]:
        print('   %-32s n=%3d avg=%9.1f us  %7.3f ms' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step']))
except Exception as e:
    print('cfg4 bench failed', e); print(open('$OUT/cfg4_bench.err').read()[-800:])
PY
