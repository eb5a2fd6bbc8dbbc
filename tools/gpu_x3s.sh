#!/bin/bash
# GPU visit for the row-streaming product kernel: its tests, then the timing table (and the ablation builds named in VARIANTS).
# Usage: [VARIANTS="nostore nodma nomfma timing"] bash tools/gpu_x3s.sh <tag>
OUT=gpurun_out/${1:-x3s}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "x3s" > $OUT/pytest_x3s.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_x3s.log
timeout 300 python tools/x3s_bench.py > $OUT/x3s_bench.txt 2>&1; echo "bench exit $?"; grep -v amdgpu.ids $OUT/x3s_bench.txt | tail -12
for v in $VARIANTS; do
  echo "--- variant $v"
  DC_LIB=$(pwd)/dotaclient_amd/libdotaclient_hip_$v.so timeout 300 python tools/x3s_bench.py > $OUT/x3s_bench_$v.txt 2>&1
  if [ "$v" = timing ]; then grep "^gemm_x3s" $OUT/x3s_bench_$v.txt | awk 'NR%24==1'; else grep -v amdgpu.ids $OUT/x3s_bench_$v.txt | tail -12 | cut -c1-60,95-140; fi
done
