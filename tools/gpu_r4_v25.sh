#!/bin/bash
OUT=gpurun_out/r4v25; mkdir -p $OUT; export TMPDIR=/tmp
ALT=$(pwd)/dotaclient_amd/libdotaclient_hip_efold.so
for rep in 1 2 3; do
for lib in "" alt; do
  L=""; [ -n "$lib" ] && L=$ALT
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
  python - <<PY
import json
j = json.loads([l for l in open('$OUT/bench_${lib}_$rep.json') if l.startswith('{')][0])
ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
print('lib %-4s rep $rep: %.1f env-steps/s %.3f ms/step  embed_fwd_fused %.1f pool_env %.1f attn_logits %.1f us' % ('$lib', j['value'], j['ms_per_step'], ks['embed_fwd_fused'], ks['pool_env_fwd'], ks['attn_logits']))
PY
done
done
