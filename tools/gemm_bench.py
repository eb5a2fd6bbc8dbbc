"""Times dc_gemm_f32 on the exact GEMM shapes of one PPO epoch (64x256 LSTM-128 by default) and, as a
ceiling reference only, the same products through torch.matmul (rocBLAS fp32).  Scratch tool, not part
of the bench contract.  Usage: python tools/gemm_bench.py [rows] [H]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import ops  # noqa: E402

NR = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
G = 4
dev = torch.device('cuda:0')


def t_ms(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def case(name, M, N, K, a_km, b_km, relu=False, aux=False, bias=False, count=1, lda=None):
    # storage shapes (lda > M: k-major A with padded rows, as the [rows][160] head-gradient buffer)
    A = torch.randn((K, lda or M) if a_km else (M, K), device=dev)
    B = torch.randn((K, N) if b_km else (N, K), device=dev)
    if os.environ.get('GEMM_ZERO'):   # zero-filled operands: the same instruction stream at a lower power draw (DVFS check)
        A.zero_(); B.zero_()
    C = torch.empty(M, N, device=dev)
    bz = torch.randn(N, device=dev) if bias else None
    ax = torch.randn(M, N, device=dev) if aux else None
    lda = (lda or M) if a_km else K
    ldb = N if b_km else K
    f = lambda: ops.gemm(A, B, C, M, N, K, lda, ldb, N, a_km, b_km, bias=bz, relu=relu, aux=ax, ldaux=N, scratch=SCRATCH)
    ms = t_ms(f)
    ms3 = err3 = ms4 = err4 = float('nan')
    if K % 16 == 0 and N % 4 == 0 and (a_km == b_km or not a_km) and not (a_km and (relu or aux)):
        f3 = lambda: ops.gemm(A, B, C, M, N, K, lda, ldb, N, a_km, b_km, bias=bz, relu=relu, aux=ax, ldaux=N, scratch=SCRATCH, x3=6)
        ms3 = t_ms(f3)
        f3()
        C3 = C.clone()
        f4 = lambda: ops.gemm(A, B, C, M, N, K, lda, ldb, N, a_km, b_km, bias=bz, relu=relu, aux=ax, ldaux=N, scratch=SCRATCH, x3=4)
        ms4 = t_ms(f4)
        f4()
        C4 = C.clone()
    Am = A[:, :M].t() if a_km else A
    Bm = B if b_km else B.t()
    ref = Am @ Bm
    if bias:
        ref = ref + bz
    if relu:
        ref = ref.clamp_min(0)
    if aux:
        ref = torch.where(ax > 0, ref, torch.zeros_like(ref))
    f()
    err = ((C - ref).abs().max() / ref.abs().max()).item()
    ms_ref = t_ms(lambda: torch.matmul(Am, Bm))
    fl = 2.0 * M * N * K
    if ms3 == ms3:
        err3 = ((C3 - ref).abs().max() / ref.abs().max()).item()
        ref64 = (Am.double() @ Bm.double()) + (bz.double() if bias else 0)
        if relu: ref64 = ref64.clamp_min(0)
        if aux: ref64 = torch.where(ax > 0, ref64, torch.zeros_like(ref64))
        err3 = ((C3.double() - ref64).abs().max() / ref64.abs().max()).item()
        err4 = ((C4.double() - ref64).abs().max() / ref64.abs().max()).item()
    print('%-22s M=%7d N=%4d K=%7d %s%s  fasttile %8.1f us %6.1f TF | x3 %8.1f us %6.1f TF | f16x2 %8.1f us %6.1f TF | rocblas %8.1f us %6.1f TF | x%d  err fasttile(vs f32) %.1e, vs f64: x3 %.1e f16x2 %.1e'
          % (name, M, N, K, 'T' if a_km else 'N', 'N' if b_km else 'T', ms * 1e3, fl / ms / 1e9, ms3 * 1e3, fl / ms3 / 1e9, ms4 * 1e3, fl / ms4 / 1e9, ms_ref * 1e3,
             fl / ms_ref / 1e9, count, err, err3, err4))
    return (ms3 if ms3 == ms3 else ms) * count


SCRATCH = torch.empty(24 << 20, device=dev)
tot = 0.0
print('--- forward')
for U in (1, 5, 16):
    tot += case('F1 unit emb U=%d' % U, NR * U, 128, 128, False, False, bias=True, count=3 if U == 1 else (2 if U == 16 else 1))
tot += case('F2 pre_rnn', NR, 256, 896, False, False, relu=True, bias=True)
tot += case('F3 W_ih', NR, G * H, 256, False, False, bias=True)
tot += case('F4 heads', NR, 154, H, False, False, bias=True)
fwd = tot
print('forward GEMMs: %.3f ms' % fwd)
print('--- backward')
tot = 0.0
tot += case('B1 dH (K pad 160)', NR, H, 160, False, True)
tot += case('B2 dW_heads', 154, H, NR, True, True, lda=160)
tot += case('B4 dW_ih', G * H, 256, NR, True, True)
tot += case('B5 dW_hh', G * H, H, NR, True, True)
tot += case('B6 dpre', NR, 256, G * H, False, True, aux=True)
tot += case('B7 dW_pre', 256, 896, NR, True, True)
tot += case('B8 dxcat', NR, 896, 256, False, True)
for U in (1, 5, 16):
    c = 3 if U == 1 else (2 if U == 16 else 1)
    tot += case('B9 dW_unit U=%d' % U, 128, 128, NR * U, True, True, count=c)
    tot += case('B10 dbasic U=%d' % U, NR * U, 128, 128, False, True, aux=True, count=c)
print('backward GEMMs: %.3f ms' % tot)
print('per bench step (5 fwd + 4 bwd): %.3f ms' % (5 * fwd + 4 * tot))
