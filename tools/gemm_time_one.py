"""A few launches of one gemm_x3 product (developer timing build: DC_LIB=.../libdotaclient_hip_timing.so prints the phase clocks).
usage: python tools/gemm_time_one.py <prec 4|6|1|16 (= 1 with bf16-stored operands)> M N K akm bkm [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import ops  # noqa: E402

prec, M, N, K, akm, bkm = [int(x) for x in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 3
dev = torch.device('cuda:0')
scratch = torch.empty(64 << 20, device=dev)
A = torch.randn((K, M) if akm else (M, K), device=dev)
B = torch.randn((K, N) if bkm else (N, K), device=dev) / 16
C = torch.empty(M, N, device=dev)
x3 = prec
if prec == 16:
    A = A.bfloat16()
    if akm:
        B = B.bfloat16()
    x3 = ops.prec_bf16_store(a=True, b=bool(akm))
elif prec == 4:
    x3 = ops.prec_f16x2(4, 8) if not akm else ops.prec_f16x2(18, 4)
    if akm:
        A = A / 65536
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(iters):
    if i == iters - 1:
        s.record()
    ops.gemm(A, B, C, M, N, K, M if akm else K, N if bkm else K, N, bool(akm), bool(bkm), scratch=scratch, x3=x3)
e.record()
torch.cuda.synchronize()
print('prec %d M %d N %d K %d akm %d bkm %d: last launch %.1f us (%.1f TF)' % (prec, M, N, K, akm, bkm, s.elapsed_time(e) * 1e3, 2.0 * M * N * K / s.elapsed_time(e) / 1e9))
