#!/bin/bash
# Round 4, visit 15: the eight-member team kernels (DC_DIMS_TEAM8, rnn_team8.hip): agreement with the per-step kernels on the ragged /
# many-round shapes, then the default bench against --kernel-flags 524288, at 256 and at 128 trajectories.
TAG=${1:-r4v15}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rnn_team_kernels_agree" > $OUT/pytest_team.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/pytest_team.log
for B in 128; do
for fl in 0 512 0 512; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $fl --batch $B > $OUT/bench_${B}_$fl.json 2> $OUT/bench_${B}_$fl.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${B}_$fl.json') if l.startswith('{')][0])
    ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
    print('batch %-3s flags %-7s: %.1f env-steps/s %.3f ms/step  lstm_fwd_team %.1f us lstm_bwd_team %.1f us' % ('$B', '$fl', j['value'], j['ms_per_step'], ks.get('lstm_fwd_team', 0), ks.get('lstm_bwd_team', 0)))
except Exception as e:
    print('bench failed', '$B', '$fl', e); print(open('$OUT/bench_${B}_$fl.err').read()[-1200:])
PY
done
done
