#!/bin/bash
# One rocprofv3 PMC pass with SQ counters over a 1-step bench (kernel-trace only, no other trace domains).
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $REPO/$OUT/pmc -o pmc -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/$OUT/pmc.log 2>&1
echo "rocprof exit $?"; tail -3 $REPO/$OUT/pmc.log
cd $REPO
python tools/pmc_sq.py $OUT/pmc $OUT/sq.json
find $OUT -name '*kernel_trace.csv' -size +20M -delete
find $OUT -name '*counter_collection.csv' -size +30M -delete
