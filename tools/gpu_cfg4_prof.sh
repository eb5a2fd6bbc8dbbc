#!/bin/bash
# rocprofv3 kernel stats of configs[4]'s shard (bf16 path)
OUT=gpurun_out/${1:-r5c4p}; mkdir -p $OUT; export TMPDIR=/tmp
CFG4="--cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --no-cpu-baseline --no-weak-unit --no-secondary --kernel-flags ${FLAGS:-4096}"
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 3 --warmup 1 $CFG4 > $OLDPWD/$OUT/prof.log 2>&1; cd $OLDPWD
python tools/rocpd_stats.py $(find $OUT/prof -name '*.db' | head -1) $OUT/cfg4_kernel_stats.csv > /dev/null 2>&1; head -${ROWS:-30} $OUT/cfg4_kernel_stats.csv | cut -c1-200
python tools/rocpd_dispatches.py $(find $OUT/prof -name '*.db' | head -1) "${DPAT:-gemm_x3}" ${DN:-20} | tee $OUT/cfg4_dispatches.txt
find $OUT/prof -name '*.db' -delete
