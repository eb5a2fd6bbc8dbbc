"""Split-K sweep of the weight-gradient (TN) products of one PPO epoch (64x256 LSTM-128): time per product and split
count, against the automatic choice of dc_gemm_f32.  Scratch tool.  Usage: python tools/gemm_split_sweep.py [rows] [H]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import ops  # noqa: E402

NR = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda:0')
SCRATCH = torch.empty(16 << 20, device=dev)


def t_us(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, M, N, lda in (('dW_heads', 154, H, 160), ('dW_ih', 4 * H, 256, None), ('dW_hh', 4 * H, H, None), ('dW_pre', 256, 896, None)):
    A = torch.randn(NR, lda or M, device=dev)
    B = torch.randn(NR, N, device=dev)
    C = torch.empty(M, N, device=dev)
    row = []
    for sp in (0, 4, 8, 16, 32, 64):
        if sp and sp * M * N > SCRATCH.numel():
            continue
        us = t_us(lambda: ops.gemm(A, B, C, M, N, NR, lda or M, N, N, True, True, splits=sp, scratch=SCRATCH))
        row.append('%s:%6.1f' % ('auto' if sp == 0 else sp, us))
    print('%-9s M=%4d N=%4d K=%6d  %s   (%.1f TF at the best)' % (name, M, N, NR, '  '.join(row), 2.0 * M * N * NR / min(float(r.split(':')[1]) for r in row) / 1e6))
