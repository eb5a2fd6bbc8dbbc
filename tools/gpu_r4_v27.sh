#!/bin/bash
# Round 4, visit 27: are the three-MFMA f16x2 products still power-limited?  tools/gemm_bench.py on random and on zero-filled operands
# (the same instruction stream at a lower power draw; profiles/r04/gemm_power_evidence.json did this for the six- and four-MFMA forms).
OUT=gpurun_out/r4v27; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/gemm_bench.py 65536 256 > $OUT/gemm_bench_random.txt 2>&1
GEMM_ZERO=1 timeout 300 python tools/gemm_bench.py 65536 256 > $OUT/gemm_bench_zero.txt 2>&1
python - <<PY
import re
a = open('$OUT/gemm_bench_random.txt').read().splitlines(); b = open('$OUT/gemm_bench_zero.txt').read().splitlines()
for x, y in zip(a, b):
    mx = re.search(r'f16x2\s+([\d.]+) us\s+([\d.]+) TF', x); my = re.search(r'f16x2\s+([\d.]+) us\s+([\d.]+) TF', y)
    if mx and my: print('%-24s random %8s us %6s TF   zeros %8s us %6s TF   ratio %.2f' % (x[:24], mx.group(1), mx.group(2), my.group(1), my.group(2), float(mx.group(1)) / float(my.group(1))))
PY
