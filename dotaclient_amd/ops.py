"""Thin tensor-level wrappers over the C ABI (argument checking + pointer plumbing only).

torch is used here for device memory and streams; all arithmetic happens in libdotaclient_hip.so.
"""
import torch

from . import _lib


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise _lib.DotaHipError('%s must live on the GPU (there is no CPU path)' % name)
    if t.dtype != dtype:
        raise TypeError('%s: expected %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    return t


def gae_scan(rewards, values, seq_off, seq_len, max_len, gamma=0.98, lam=0.97, adv=None, ret=None):
    """rewards [rows,10] f32, values [rows] f32, seq_off i64 [n], seq_len i32 [n] -> (adv, ret) [rows] f32.
    Replaces advantage_returns + sub-reward sum (optimizer.py:53-64,397,417-421)."""
    lib = _lib.load()
    _chk(rewards, torch.float32, 'rewards'); _chk(values, torch.float32, 'values')
    _chk(seq_off, torch.int64, 'seq_off'); _chk(seq_len, torch.int32, 'seq_len')
    rows = values.numel()
    assert rewards.shape == (rows, 10)
    if adv is None:
        adv = torch.empty(rows, dtype=torch.float32, device=values.device)
    if ret is None:
        ret = torch.empty(rows, dtype=torch.float32, device=values.device)
    _lib.check(lib.dc_gae_scan(_lib.ptr(rewards), _lib.ptr(values), _lib.ptr(seq_off), _lib.ptr(seq_len),
                               seq_off.numel(), int(max_len), float(gamma), float(lam), _lib.ptr(adv),
                               _lib.ptr(ret), _lib.stream_ptr()), 'dc_gae_scan')
    return adv, ret


def discount(x, gamma, y=None):
    """x [n] f32 (GPU) -> y[t] = x[t] + gamma * y[t+1] (optimizer.py:53-54)."""
    lib = _lib.load()
    _chk(x, torch.float32, 'x')
    if y is None:
        y = torch.empty_like(x)
    _lib.check(lib.dc_discount(_lib.ptr(x), x.numel(), float(gamma), _lib.ptr(y), _lib.stream_ptr()), 'dc_discount')
    return y


def advantage_returns(rewards, values, gamma, lam):
    """rewards, values [L+1] f32 (GPU; any terminal entries) -> (adv, ret) [L] f32 (optimizer.py:57-64)."""
    lib = _lib.load()
    _chk(rewards, torch.float32, 'rewards'); _chk(values, torch.float32, 'values')
    n = rewards.numel() - 1
    if values.numel() != n + 1:
        raise ValueError('advantage_returns: rewards and values must have the same length')
    adv = torch.empty(max(n, 0), dtype=torch.float32, device=values.device)
    ret = torch.empty_like(adv)
    _lib.check(lib.dc_advantage_returns(_lib.ptr(rewards), _lib.ptr(values), n, float(gamma), float(lam), _lib.ptr(adv),
                                        _lib.ptr(ret), _lib.stream_ptr()), 'dc_advantage_returns')
    return adv, ret


def prec_f16x2(log2_sa=0, log2_sb=0, tile128=False):
    """`x3` value for the two-f16-piece products with power-of-two pre-scales 2^log2_sa / 2^log2_sb of the A / B operand
    (DC_GEMM_PREC_F16X2, include/dotaclient_hip.h): what the network passes for activations (4), weights (8), gradients (ceil(log2 rows) + 2).
    tile128 (DC_GEMM_PREC_TILE128): keep x W^T / dy W on the 128 x 128 split-on-load kernel where the row-streaming kernel would run."""
    return 4 | ((log2_sa & 0xff) << 8) | ((log2_sb & 0xff) << 16) | (int(tile128) << 24)


def prec_bf16_store(a=False, b=False, c=False, aux=False):
    """`x3` value for the plain-bf16 products (prec 1) with A / B / C / aux STORED as bf16 tensors of the same shape and leading dimension
    (DC_GEMM_PREC_BF16_STORE, include/dotaclient_hip.h): configs[4]'s gate buffers."""
    return 1 | (int(a) << 8) | (int(b) << 9) | (int(c) << 10) | (int(aux) << 11)


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, bias=None, relu=False, aux=None,
         ldaux=0, accumulate=False, splits=0, scratch=None, x3=0):
    """x3 = 6 / 4 / 1: the split-on-load matrix-core kernel (dc_gemm_x3: three bf16 pieces / two f16 pieces (prec_f16x2 for pre-scales) /
    plain bf16); 0: dc_gemm_f32."""
    lib = _lib.load()
    stored = (x3 >> 8) & 0xf if (x3 & 0xff) == 1 else 0         # bf16-stored operands (prec_bf16_store)
    for i, (t, n) in enumerate(((A, 'A'), (B, 'B'), (C, 'C'))):
        _chk(t, torch.bfloat16 if (stored >> i) & 1 else torch.float32, n)
    if aux is not None:
        _chk(aux, torch.bfloat16 if stored & 8 else torch.float32, 'aux')
    if x3:
        _lib.check(lib.dc_gemm_x3(_lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, lda, ldb, ldc, int(a_kmajor), int(b_kmajor),
                                  _lib.ptr(bias), int(relu), _lib.ptr(aux), ldaux, int(accumulate), int(x3), _lib.ptr(scratch),
                                  0 if scratch is None else scratch.numel(), _lib.stream_ptr()), 'dc_gemm_x3')
        return C
    _lib.check(lib.dc_gemm_f32(_lib.ptr(A), _lib.ptr(B), _lib.ptr(C), M, N, K, lda, ldb, ldc, int(a_kmajor),
                               int(b_kmajor), _lib.ptr(bias), int(relu), _lib.ptr(aux), ldaux, int(accumulate),
                               splits, _lib.ptr(scratch), 0 if scratch is None else scratch.numel(), _lib.stream_ptr()),
               'dc_gemm_f32')
    return C
