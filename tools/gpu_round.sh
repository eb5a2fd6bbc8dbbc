#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats, HBM-traffic PMC passes.
# Usage (from the repo root on the GPU box; default workload = bench.py's default, BASELINE.json configs[2]): [BENCH_ARGS='--cell gru --hidden 256 --batch 64' WORKLOAD=gru-256-64x256] bash tools/gpu_round.sh <tag> [skip_tests]
TAG=${1:-vX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
if [ -z "$2" ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
WORKLOAD=${WORKLOAD:-lstm-256-256x256}
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extras --extras-out $OUT/bench_extras.json $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err   # the driver's command (+ the side measurements, after the line is out)
tail -c 1500 $OUT/bench.json
# kernel-trace + stats of the same command (no CPU baseline and no 128-trajectory side run inside the traced runs: one workload per trace)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-weak-unit $BENCH_ARGS > $REPO/$OUT/prof_bench.log 2>&1
# PMC passes, each on its own (no trace domains besides kernel-trace)
timeout 180 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT/pmc_fetch -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-weak-unit $BENCH_ARGS > $REPO/$OUT/pmc_fetch.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $REPO/$OUT/pmc_write -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-weak-unit $BENCH_ARGS > $REPO/$OUT/pmc_write.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $REPO/$OUT/pmc_sq -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-weak-unit $BENCH_ARGS > $REPO/$OUT/pmc_sq.log 2>&1
cd $REPO
python tools/pmc_sq.py $OUT/pmc_sq $OUT/pmc_sq.json > $OUT/pmc_sq.txt 2>&1
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
[ -n "$DB" ] && python tools/rocpd_dispatches.py $DB gemm_x3 39 > $OUT/gemm_dispatches.txt 2>&1    # one step's products, in launch order
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json $WORKLOAD 2>&1 | tail -15
# keep the merge-back small: drop raw traces
find $OUT/prof -name '*.db' -size +20M -delete
find $OUT -name '*kernel_trace.csv' -size +20M -delete
find $OUT -name '*counter_collection.csv' -size +20M -delete
ls -la $OUT
