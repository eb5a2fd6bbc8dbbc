#!/bin/bash
# phase clocks of gemm_x3 (developer timing build) on the headline's and configs[4]'s product shapes
export TMPDIR=/tmp
T=$(pwd)/dotaclient_amd/libdotaclient_hip_timing.so
for args in "4 65536 1024 256 0 0" "4 65536 256 896 0 0" "4 65536 256 1024 0 1" "4 1024 512 65536 1 1" "16 131072 2048 512 0 0" "16 131072 512 2048 0 1" "16 2048 1024 131072 1 1"; do
  python tools/gemm_time_one.py $args 2>/dev/null | tail -1
  DC_LIB=$T python tools/gemm_time_one.py $args 2>&1 | grep "gemm_x3<" | tail -1 | cut -c1-330
done
