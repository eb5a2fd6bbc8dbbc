#!/bin/bash
# Phase clocks of the H = 256 team forward (-DTM_TIMING builds of rnn_team_mfma.hip print s_memtime ticks per step and phase):
# usage: VARIANTS="tmt tmt32" bash tools/gpu_team_timing.sh <tag>
OUT=gpurun_out/${1:-teamt}; mkdir -p $OUT
for v in $VARIANTS; do
  DC_LIB=$(pwd)/dotaclient_amd/libdotaclient_hip_$v.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-weak-unit > $OUT/bench_$v.txt 2>&1
  echo "--- $v"; grep "^team_" $OUT/bench_$v.txt | tail -${TAIL:-12} | cut -c1-320
done
