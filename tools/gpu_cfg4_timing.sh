#!/bin/bash
# Phase clocks of the LSTM-512 team kernels (developer build: DC_BUILD_VARIANT=timing DC_BUILD_FLAGS=-DDC_DEV_TIMING=1) on configs[4]'s shard,
# bf16 against f32 storage of the gate buffers
OUT=gpurun_out/${1:-r5c4t}; mkdir -p $OUT; export TMPDIR=/tmp
CFG4="--cell lstm --hidden 512 --layers 2 --batch 256 --seq-len 512 --no-cpu-baseline --no-weak-unit --no-secondary"
for mode in bf16 f32; do
  F=4096; [ "$mode" = "f32" ] && F=$((4096 + 4194304))
  DC_LIB=$(pwd)/dotaclient_amd/libdotaclient_hip_timing.so timeout 300 python bench.py --steps 2 --warmup 1 $CFG4 --kernel-flags $F > $OUT/t_${mode}.json 2> $OUT/t_${mode}.err
  echo "== storage $mode"; grep "lstm512_team_fwd timing" $OUT/t_${mode}.err | tail -4 | cut -c1-330; grep "lstm512_team_bwd timing" $OUT/t_${mode}.err | tail -4 | cut -c1-330
done
