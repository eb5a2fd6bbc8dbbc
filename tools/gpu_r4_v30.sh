#!/bin/bash
# Round 4, visit 30: the mask-aware attention kernels with one wave (logits) / one block half (d query) per env-step instead of a capped
# grid walking four / sixteen steps per wave (A/B build -DATTN_GRID_CAP=65536 against the default 4096 blocks).
OUT=gpurun_out/r4v30; mkdir -p $OUT; export TMPDIR=/tmp
ALT=$(pwd)/dotaclient_amd/libdotaclient_hip_attncap.so
for rep in 1 2; do
for lib in "" alt; do
  L=""; [ -n "$lib" ] && L=$ALT
  DC_LIB=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
  python - <<PY
import json
j = json.loads([l for l in open('$OUT/bench_${lib}_$rep.json') if l.startswith('{')][0])
ks = {k['kernel']: k['avg_us'] for k in j['roofline']['kernels']}
print('lib %-4s rep $rep: %.1f env-steps/s %.3f ms/step  attn_logits %.1f attn_bwd_q %.1f us' % ('$lib', j['value'], j['ms_per_step'], ks['attn_logits'], ks['attn_bwd_q']))
PY
done
done
