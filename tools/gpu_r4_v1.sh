#!/bin/bash
# Round 4, visit 1: suite + driver bench line (now with weak_scaling_unit) + the self-launched two-rank flow on one device (gloo) +
# publish probe + power-limit evidence for the dense products (random vs zero operands: sysfs clock / power samples, GRBM_GUI_ACTIVE).
TAG=${1:-r4v1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
cp gpurun_out/elementwise_parity.json $OUT/ 2>/dev/null
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
# plain `python bench.py --gpus 2`: must launch two ranks itself (here both on cuda:0 over gloo: DC_BENCH_ONE_DEVICE=1) and print n_gpus 2
DC_BENCH_ONE_DEVICE=1 timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_selflaunch2.json 2> $OUT/bench_selflaunch2.err
echo "self-launch exit $?"; python - <<PY
import json
ls = [l for l in open('$OUT/bench_selflaunch2.json') if l.startswith('{')]
print('json lines', len(ls))
if ls:
    j = json.loads(ls[0]); print({k: j[k] for k in ('value', 'n_gpus', 'ms_per_step', 'nan_status')}, j['config']['parallelism'])
PY
# without the test aid on this one-GPU box: must refuse (exit 2, no line)
timeout 120 python3 bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_refuse.out 2> $OUT/bench_refuse.err; echo "refuse exit $? (want 2), stdout bytes $(wc -c < $OUT/bench_refuse.out)"
timeout 300 python tools/publish_probe.py $OUT/publish_probe.json > $OUT/publish_probe.log 2>&1; tail -16 $OUT/publish_probe.log
timeout 600 python tools/gemm_power_evidence.py $OUT/gemm_power_evidence.json > $OUT/gemm_power_evidence.log 2>&1; tail -6 $OUT/gemm_power_evidence.log | cut -c1-400
cd /tmp
for fill in random zero; do
  Z=""; [ $fill = zero ] && Z=1
  GEMM_ZERO=$Z timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $REPO/$OUT/pmc_clock_$fill -o pmc -- python $REPO/tools/gemm_bench.py 65536 256 > $REPO/$OUT/pmc_clock_$fill.log 2>&1
  python $REPO/tools/pmc_clock.py $REPO/$OUT/pmc_clock_$fill $REPO/$OUT/pmc_clock_$fill.json > $REPO/$OUT/pmc_clock_$fill.txt 2>&1
  head -8 $REPO/$OUT/pmc_clock_$fill.txt
done
cd $REPO
find $OUT -name '*.csv' -size +5M -delete
find $OUT -name '*.db' -delete
ls $OUT
