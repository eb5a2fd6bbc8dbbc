"""Times the prec-1 (bf16) products of gemm_x3.hip on configs[4]'s shapes with the operand / output storage variants, and the same
products through torch.matmul on bf16 tensors (hipBLASLt) as a ceiling reference.  Scratch tool, not part of the bench contract.
usage: python tools/gemm_bf16_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import ops  # noqa: E402

NR = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dev = torch.device('cuda:0')
SCRATCH = torch.empty(64 << 20, device=dev)


def t_us(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def case(name, M, N, K, akm, bkm, a16, b16, c16):
    A = torch.randn((K, M) if akm else (M, K), device=dev)
    B = torch.randn((K, N) if bkm else (N, K), device=dev) / 16
    Ad = A.bfloat16() if a16 else A
    Bd = B.bfloat16() if b16 else B
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
    lda, ldb = (M if akm else K), (N if bkm else K)
    x3 = ops.prec_bf16_store(a=a16, b=b16, c=c16)
    f = lambda: ops.gemm(Ad, Bd, C, M, N, K, lda, ldb, N, akm, bkm, scratch=SCRATCH, x3=x3)
    us = t_us(f)
    Am = (A.t() if akm else A).bfloat16()
    Bm = (B if bkm else B.t()).bfloat16()
    us_ref = t_us(lambda: torch.matmul(Am, Bm))
    f()
    ref = (Am.float() @ Bm.float())
    err = ((C.float() - ref).abs().max() / ref.abs().max()).item()
    fl = 2.0 * M * N * K
    by = (2 if a16 else 4) * M * K + (2 if (b16 or not akm) else 4) * N * K + (2 if c16 else 4) * M * N
    print('%-26s M=%7d N=%5d K=%7d A %s B %s C %s  %8.1f us %7.1f TF %5.2f TB/s | torch bf16 matmul %8.1f us %7.1f TF | err %.1e'
          % (name, M, N, K, 'bf16' if a16 else 'f32 ', 'bf16' if (b16 or not akm) else 'f32 ', 'bf16' if c16 else 'f32 ', us, fl / us / 1e6, by / us / 1e6,
             us_ref, fl / us_ref / 1e6, err))


H = 512
for a16, c16 in ((False, False), (False, True), (True, True)):
    case('fwd L0 x W_ih^T', NR, 4 * H, 256, False, False, a16, False, c16)
    case('fwd L1 h W_ih^T', NR, 4 * H, H, False, False, a16, False, c16)
case('fwd pre', NR, 256, 896, False, False, False, False, False)
case('fwd L1, N = 512 only', NR, 512, H, False, False, True, False, True)
for a16 in (False, True):
    case('dX L1 dg W_ih', NR, H, 4 * H, False, True, a16, False, False)
    case('dX L0 dg W_ih', NR, 256, 4 * H, False, True, a16, False, False)
for a16, b16 in ((False, False), (True, False), (True, True)):
    case('dW L1 dg^T [h | hprev]', 4 * H, 2 * H, NR, True, True, a16, b16, False)
    case('dW L0 dg^T [x | hprev]', 4 * H, 256 + H, NR, True, True, a16, b16, False)
