#!/bin/bash
# GPU-box visit for the H = 256 cells (the reference's GRU-256, LSTM-256 of BASELINE.json configs[2..3]): the team-kernel
# tests, the phase timing of the forward kernel, short bench lines with the team kernels on and off.
# Usage: bash tools/gpu_h256.sh [tag]
TAG=${1:-h256}; OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team" > $OUT/pytest.log 2>&1
rc=$?; echo "pytest exit $rc"; tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit 0
for cfg in "lstm 256 64" "lstm 256 256"; do set -- $cfg
    echo "== $1-$2 batch $3 timing"
    DC_TEAM_TIMING=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --cell $1 --hidden $2 --batch $3 2>&1 | grep "timing" | tail -1
done
for cfg in "gru 256 64" "gru 256 256" "lstm 256 64" "lstm 256 256"; do set -- $cfg
  for m in 1 0; do
    echo "== $1-$2 batch $3 DC_RNN_TEAM=$m"
    DC_RNN_TEAM=$m timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cell $1 --hidden $2 --batch $3 2>$OUT/bench.err | tee $OUT/bench_$1_$2_$3_team$m.json | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step')
        for k in j['roofline']['kernels'][:4]: print('   %-28s n=%4d avg=%9.1f us %7.3f ms' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step']))
"
  done
done
