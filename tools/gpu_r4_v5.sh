#!/bin/bash
# Round 4, visit 5: the full measurement visit at the new defaults (Engine.products f16x2): tools/gpu_round.sh (suite, driver bench + extras,
# rocprofv3 kernel stats, FETCH / WRITE / SQ PMC passes), then the bf16x3 line on the same box, configs[4]'s shard the driver's way, the
# self-launched two-rank flow on one device.
TAG=${1:-r4v5}
OUT=gpurun_out/$TAG
bash tools/gpu_round.sh $TAG
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit --products bf16x3 > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err
python - <<PY
import json
for f in ('bench', 'bench_bf16x3'):
    try:
        j = json.loads([l for l in open('$OUT/%s.json' % f) if l.startswith('{')][0])
        print(f, j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step', j['config'].get('products'), 'parity', j['parity'] and (j['parity']['ok'], j['parity']['parity_rel_err']),
              'weak unit', j.get('weak_scaling_unit') and j['weak_scaling_unit']['value'], 'cpu', j['cpu_baseline'] and j['cpu_baseline']['value'])
        for k in j['roofline']['kernels']:
            print('   %-32s n=%3d avg=%8.1f us %6.3f ms  %s frac %s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k.get('achieved_tflops') or k.get('achieved_gbs'), k.get('frac') or k.get('frac_of_hbm_peak')))
    except Exception as e:
        print(f, 'failed', e)
PY
timeout 900 python3 bench.py --cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --kernel-flags 4096 --steps 20 --warmup 5 > $OUT/cfg4_bench.json 2> $OUT/cfg4_bench.err
python - <<PY
import json
try:
    j = json.loads([l for l in open('$OUT/cfg4_bench.json') if l.startswith('{')][0])
    print('cfg4', j['value'], 'env-steps/s', j['ms_per_step'], 'ms/step', 'parity', j['parity'] and (j['parity']['ok'], j['parity']['parity_rel_err'], j['parity']['argmax_equal_fraction']), 'cpu', j['cpu_baseline'] and j['cpu_baseline']['value'])
    for k in j['roofline']['kernels'][:8]:
        print('   %-32s n=%3d avg=%9.1f us  %7.3f ms  %s  frac %s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k.get('achieved_tflops') or k.get('achieved_gbs'), k.get('frac')))
except Exception as e:
    print('cfg4 bench failed', e)
PY
DC_BENCH_ONE_DEVICE=1 timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_selflaunch2.json 2> $OUT/bench_selflaunch2.err
echo "self-launch exit $?"; grep -o '"n_gpus": [0-9]*' $OUT/bench_selflaunch2.json
ls $OUT
