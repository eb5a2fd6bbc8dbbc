#!/bin/bash
# Round 4, visit 9: sixteen-wave sparse backward with the branch-free FULL iteration body (HEAD) and two A/B builds that ask the scheduler to
# spread the block's seven f32 MFMAs (sched_group_barrier, 24 / 40 vector-or-LDS instructions between them); eight-wave kernel for reference.
TAG=${1:-r4v9}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for lib in "" p16s24 p16s40; do
  L=""; [ -n "$lib" ] && L=$(pwd)/dotaclient_amd/libdotaclient_hip_$lib.so
  DC_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sparse_pool or golden" > $OUT/pytest_$lib.log 2>&1
  echo "lib '$lib' pytest exit $?"; tail -2 $OUT/pytest_$lib.log
done
for rep in 1 2; do
for cfg in ":262144" ":0" "p16s24:0" "p16s40:0"; do
  lib=${cfg%%:*}; fl=${cfg##*:}
  L=""; [ -n "$lib" ] && L=$(pwd)/dotaclient_amd/libdotaclient_hip_$lib.so
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --kernel-flags $fl > $OUT/bench_${lib}_$fl.json 2> $OUT/bench_${lib}_$fl.err
  python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/bench_${lib}_$fl.json') if l.startswith('{')][0])
    k = [x for x in j['roofline']['kernels'] if x['kernel'] == 'embed_bwd_pool16'][0]
    print('lib %-7s flags %-7s: %.1f env-steps/s %.3f ms/step  embed_bwd_pool16 %.1f us' % ('$lib', '$fl', j['value'], j['ms_per_step'], k['avg_us']))
except Exception as e:
    print('bench failed', '$lib', '$fl', e); print(open('$OUT/bench_${lib}_$fl.err').read()[-800:])
PY
done
done
ls $OUT | head -3
