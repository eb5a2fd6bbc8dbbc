"""One bf16-stored product of gemm_x3.hip, a few launches (for a rocprofv3 --pmc pass).  usage: python tools/gemm_bf16_one.py fwd|dx|dw [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import ops  # noqa: E402

form = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
NR, H = 131072, 512
scratch = torch.empty(64 << 20, device=dev)
if form == 'fwd':      # L1: h W_ih^T
    M, N, K, akm, bkm = NR, 4 * H, H, False, False
elif form == 'dx':     # L1: dg W_ih
    M, N, K, akm, bkm = NR, H, 4 * H, False, True
else:                  # L1: dg^T [h | hprev]
    M, N, K, akm, bkm = 4 * H, 2 * H, NR, True, True
A = torch.randn((K, M) if akm else (M, K), device=dev).bfloat16()
B = torch.randn((K, N) if bkm else (N, K), device=dev) / 16
if akm:
    B = B.bfloat16()
c16 = form == 'fwd'
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
for _ in range(iters):
    ops.gemm(A, B, C, M, N, K, M if akm else K, N if bkm else K, N, akm, bkm, scratch=scratch, x3=ops.prec_bf16_store(a=True, b=akm, c=c16))
torch.cuda.synchronize()
