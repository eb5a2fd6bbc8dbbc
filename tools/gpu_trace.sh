#!/bin/bash
# Kernel-trace only: per-kernel totals of a short bench run (every kernel, also those outside bench.py's regions).
# Usage: bash tools/gpu_trace.sh <tag> [bench args]
TAG=${1:-t}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/prof -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $REPO/$OUT/prof_bench.log 2>&1
cd $REPO
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.csv
find $OUT/prof -name '*.db' -size +20M -delete
head -45 $OUT/kernel_stats.csv
