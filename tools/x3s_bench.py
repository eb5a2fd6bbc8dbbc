"""Times the row-streaming f16x2 kernel (gemm_x3s.hip) against the 128 x 128 split-on-load kernel (gemm_x3.hip, PREC 4) on the
network's x W^T / dy W shapes (configs[2]: 65 536 rows) and checks both against f64.  Scratch tool.  Usage: python tools/x3s_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import ops  # noqa: E402

NR = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device('cuda:0')


def t_us(fn, iters=20):
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


def case(name, M, N, K, bkm, relu=False, aux=False, la=4):
    A = torch.randn(M, K, device=dev)
    B = torch.randn((K, N) if bkm else (N, K), device=dev) / 16
    bias = torch.randn(N, device=dev)
    ax = torch.randn(M, N, device=dev) if aux else None
    C = torch.empty(M, N, device=dev)
    scratch = torch.empty(2 * N * K + 1024, device=dev)
    ref = A.double() @ (B if bkm else B.t()).double() + bias.double()
    if relu:
        ref = ref.clamp_min(0)
    if aux:
        ref = ref * (ax > 0)
    out = []
    for tile128 in (True, False):
        prec = ops.prec_f16x2(la, 8, tile128=tile128)
        f = lambda: ops.gemm(A, B, C, M, N, K, K, N if bkm else K, N, False, bkm, bias=bias, relu=relu, aux=ax, ldaux=N, scratch=scratch, x3=prec)
        us = t_us(f)
        f()
        err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
        out.append((us, err))
    fl = 2.0 * M * N * K
    print('%-26s M=%6d N=%4d K=%4d  tile128 %7.1f us %6.1f TF err %.1e | streaming %7.1f us %6.1f TF (%.3f of 833) err %.1e | x%.2f'
          % (name, M, N, K, out[0][0], fl / out[0][0] / 1e6, out[0][1], out[1][0], fl / out[1][0] / 1e6, fl / out[1][0] / 1e6 / 833.0, out[1][1],
             out[0][0] / out[1][0]), flush=True)


case('pre-rnn x W^T (relu)', NR, 256, 896, False, relu=True)
case('gates x W^T', NR, 1024, 256, False)
case('heads x W^T', NR, 160, 256, False)
case('dH = dheadout W', NR, 256, 160, True)
case('dpre = dgates W (mask)', NR, 256, 1024, True, aux=True)
case('dxcat = dpre W', NR, 896, 256, True)
case('GRU gates x W^T', NR, 768, 256, False)
