"""GPU: unit parity of the individual HIP kernels (called through the C ABI) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_optimizer as RO
from tests import util

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def _gae_oracle(rew10, values, lens):
    adv, ret = [], []
    o = 0
    for L in lens:
        r = rew10[o:o + L].sum(axis=1)
        a, t = RO.advantage_returns(np.append(r, np.float32(0)), np.append(values[o:o + L], np.float32(0)), 0.98, 0.97)
        adv.append(a); ret.append(t); o += L
    return np.concatenate(adv), np.concatenate(ret)


@pytest.mark.parametrize('lens', [[16], [1], [48, 64, 32], [256] * 64, [7, 300, 64, 1, 2049, 640], [20000], [20480], [20481, 5],
                                  [50000, 300, 41000], [61441]])
def test_gae_scan_bit_exact(lens):
    from dotaclient_amd import ops
    dev = _dev()
    rng = np.random.Generator(np.random.PCG64(len(lens) * 131 + lens[0]))
    rows = sum(lens)
    rew = (0.05 * rng.standard_normal((rows, 10))).astype(np.float32)
    val = rng.standard_normal(rows).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    adv, ret = ops.gae_scan(torch.from_numpy(rew).to(dev), torch.from_numpy(val).to(dev),
                            torch.from_numpy(off).to(dev), torch.tensor(lens, dtype=torch.int32, device=dev), max(lens))
    ea, er = _gae_oracle(rew, val, lens)
    adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
    # integer-like bar: the float64 scan is re-associated across lanes (<= a few 1e-16 relative before
    # the cast), so results are expected bit-identical to the reference's sequential lfilter
    assert np.array_equal(ret, er), np.abs(ret - er).max()
    assert np.array_equal(adv, ea), np.abs(adv - ea).max()


def test_gae_scan_golden_vector():
    from dotaclient_amd import ops
    dev = _dev()
    g = np.load(util.GOLDEN + '/gae_kat.npz')
    r, v = g['r2'][:-1], g['v2'][:-1]
    rew = np.zeros((r.size, 10), np.float32); rew[:, 3] = r
    adv, ret = ops.gae_scan(torch.from_numpy(rew).to(dev), torch.from_numpy(v.copy()).to(dev),
                            torch.zeros(1, dtype=torch.int64, device=dev),
                            torch.tensor([r.size], dtype=torch.int32, device=dev), r.size)
    assert np.array_equal(adv.cpu().numpy(), g['adv2']) and np.array_equal(ret.cpu().numpy(), g['ret2'])


def test_gae_rollouts_longer_than_one_lds_block_golden():
    # the reference's lfilter has no length limit (optimizer.py:53-64); rollouts / vectors longer than one LDS
    # block (20 480 steps, 40 960 for discount) are scanned block by block from the end with a float64 carry.  50 000 steps against
    # the REAL reference's output, every entry bit-exact (SHA-256 of the float32 bytes, tests/golden/gae_long.npz), through all
    # three entry points.
    from dotaclient_amd import ops
    from dotaclient_amd.optimizer import advantage_returns, discount
    dev = _dev()
    g = np.load(util.GOLDEN + '/gae_long.npz')
    n = int(g['n'])
    r, v, x = util.gae_long_inputs(n, int(g['seed']))
    r0, v0 = r.copy(), v.copy()
    r0[-1] = 0; v0[-1] = 0
    for tag, rr, vv in (('zero_terminal', r0, v0), ('any_terminal', r, v)):
        adv, ret = advantage_returns(rr, vv, 0.98, 0.97)
        assert np.array_equal(adv[::997], g[tag + '_adv_samples']) and np.array_equal(ret[::997], g[tag + '_ret_samples']), tag
        assert np.array_equal(util.sha256_of(adv), g[tag + '_adv_sha256']) and np.array_equal(util.sha256_of(ret), g[tag + '_ret_sha256']), tag
    d = discount(x, 0.98 * 0.97)
    assert np.array_equal(d[::997], g['discount_samples']) and np.array_equal(util.sha256_of(d), g['discount_sha256'])
    # the batched scan of the rollout pass (sub-rewards summed on the way; terminal zeros)
    rew = np.zeros((n, 10), np.float32); rew[:, 7] = r0[:-1]
    adv, ret = ops.gae_scan(torch.from_numpy(rew).to(dev), torch.from_numpy(v0[:-1].copy()).to(dev),
                            torch.zeros(1, dtype=torch.int64, device=dev), torch.tensor([n], dtype=torch.int32, device=dev), n)
    assert np.array_equal(util.sha256_of(adv.cpu().numpy()), g['zero_terminal_adv_sha256'])
    assert np.array_equal(util.sha256_of(ret.cpu().numpy()), g['zero_terminal_ret_sha256'])


def test_discount_and_advantage_returns_any_terminals():
    # optimizer.py:53-64 with the reference's own signatures: non-zero terminal reward / bootstrap value, `discount` on its
    # own - bit-exact against vectors produced by the real reference (gae_kat.npz) and against the oracle on other lengths
    from dotaclient_amd.optimizer import advantage_returns, discount
    _dev()
    g = np.load(util.GOLDEN + '/gae_kat.npz')
    adv, ret = advantage_returns(g['r3'], g['v3'], 0.98, 0.97)
    assert adv.dtype == np.float32 and adv.shape == (g['r3'].size - 1,)
    assert np.array_equal(adv, g['adv3']) and np.array_equal(ret, g['ret3'])
    assert np.array_equal(discount(g['x4'], 0.98), g['disc4'])
    assert np.array_equal(discount(g['x4'][:65], 0.98 * 0.97), g['disc4b'])
    rng = np.random.Generator(np.random.PCG64(11))
    for n in (2, 3, 64, 65, 129, 1000, 20001, 20481, 20482, 40961, 45000, 90001):
        r = (0.3 * rng.standard_normal(n)).astype(np.float32)
        v = rng.standard_normal(n).astype(np.float32)
        a, t = advantage_returns(r, v, 0.98, 0.97)
        ea, et = RO.advantage_returns(r, v, 0.98, 0.97)
        assert np.array_equal(a, ea) and np.array_equal(t, et), n
        assert np.array_equal(discount(r, 0.9), RO.discount(r, 0.9)), n
    assert advantage_returns(np.zeros(1, np.float32), np.zeros(1, np.float32), 0.98, 0.97)[0].shape == (0,)
    assert discount(np.zeros(0, np.float32), 0.98).shape == (0,)


@pytest.mark.parametrize('akm,bkm', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(64, 64, 32), (200, 154, 256), (1000, 256, 896), (130, 70, 154), (768, 256, 5000),
                                   (33, 12, 7), (2048, 768, 256)])
def test_gemm_layouts(akm, bkm, M, N, K):
    from dotaclient_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    lda = (M if akm else K) + 4
    ldb = (N if bkm else K) + 8
    A = torch.randn((K if akm else M), lda, generator=g)
    B = torch.randn((K if bkm else N), ldb, generator=g)
    bias = torch.randn(N, generator=g)
    Am = (A[:, :M].t() if akm else A[:, :K]).double()
    Bm = (B[:, :N] if bkm else B[:, :K].t()).double()
    ref = Am @ Bm + bias.double()
    ldc = N + 3
    C = torch.full((M, ldc), 7.0, device=dev)
    ops.gemm(A.to(dev), B.to(dev), C, M, N, K, lda, ldb, ldc, akm, bkm, bias=bias.to(dev))
    out = C.cpu()
    assert torch.all(out[:, N:] == 7.0), 'wrote outside the N columns'
    err = (out[:, :N].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err


@pytest.mark.parametrize('prec,tol', [(6, 2e-6), (4, 2e-6), (1, 2e-2)])
@pytest.mark.parametrize('akm,bkm', [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize('M,N,K', [(64, 64, 32), (200, 160, 256), (1000, 256, 896), (130, 72, 160), (768, 256, 5008),
                                   (33, 12, 16), (2048, 768, 256), (154, 256, 8192)])
def test_gemm_x3_layouts(prec, tol, akm, bkm, M, N, K):
    # the split-on-load kernel of gemm_x3.hip (dc_gemm_x3): f32-grade at prec 6 (the same bar as the f32 kernel) and at prec 4 - the
    # two-f16-piece / three-MFMA form that is the network's DEFAULT arithmetic (Engine.products 'f16x2'; unit pre-scales here, the
    # operand classes with their pre-scales and the range edges are below) -, operands
    # rounded to bf16 at prec 1 (error ~ 2^-9 sqrt(K)-ish of the operand scale: 2e-2 of max |C| is loose but catches layout bugs);
    # ragged M / N (row clamp + masked 16-byte stores), padded leading dimensions, split-K on the long-K k-major shapes
    from dotaclient_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + prec)
    if prec in (4, 6) and K > 4096:
        tol = 5e-6                               # f32 accumulation over K > 4096 terms (the f32 fma chain is at 3.5e-6 there as well)
    lda = ((M + 3) // 4 * 4 if akm else K) + 4   # k-major rows are read 16 bytes at a time: leading dimension % 4 == 0
    ldb = (N if bkm else K) + 8
    A = torch.randn((K if akm else M), lda, generator=g)
    B = torch.randn((K if bkm else N), ldb, generator=g)
    bias = torch.randn(N, generator=g)
    Am = (A[:, :M].t() if akm else A[:, :K]).double()
    Bm = (B[:, :N] if bkm else B[:, :K].t()).double()
    tn = akm and bkm
    ref = Am @ Bm + (0 if tn else bias.double())
    ldc = N + 4
    C = torch.full((M, ldc), 7.0, device=dev)
    scratch = torch.empty(max(2 * N * K, 16 * M * N) + 1024, device=dev)
    ops.gemm(A.to(dev), B.to(dev), C, M, N, K, lda, ldb, ldc, akm, bkm, bias=None if tn else bias.to(dev), scratch=scratch, x3=prec)
    out = C.cpu()
    assert torch.all(out[:, N:] == 7.0), 'wrote outside the N columns'
    err = (out[:, :N].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    if prec in (4, 6) and not tn:                 # epilogues: relu, mask, accumulate
        aux = torch.randn(M, N, generator=g)
        C2 = torch.empty(M, N, device=dev)
        ops.gemm(A.to(dev), B.to(dev), C2, M, N, K, lda, ldb, N, akm, bkm, bias=bias.to(dev), relu=True, scratch=scratch, x3=prec)
        assert (C2.cpu().double() - ref.clamp(min=0)).abs().max() / ref.abs().max() < tol
        ops.gemm(A.to(dev), B.to(dev), C2, M, N, K, lda, ldb, N, akm, bkm, bias=bias.to(dev), aux=aux.to(dev), ldaux=N, scratch=scratch, x3=prec)
        assert (C2.cpu().double() - ref * (aux > 0)).abs().max() / ref.abs().max() < tol
        C2.fill_(1.0)
        ops.gemm(A.to(dev), B.to(dev), C2, M, N, K, lda, ldb, N, akm, bkm, bias=bias.to(dev), accumulate=True, scratch=scratch, x3=prec)
        assert (C2.cpu().double() - (ref + 1)).abs().max() / ref.abs().max() < tol
    if tn:                                        # accumulate into live data through the split-K reduce
        C3 = torch.full((M, N), 2.0, device=dev)
        ops.gemm(A.to(dev), B.to(dev), C3, M, N, K, lda, ldb, N, True, True, accumulate=True, scratch=scratch, x3=prec)
        assert (C3.cpu().double() - (ref + 2)).abs().max() / ref.abs().max() < tol


@pytest.mark.parametrize('bkm', [False, True])
@pytest.mark.parametrize('M,N,K', [(65536, 256, 896), (49152, 1024, 256), (50000, 896, 256), (65536, 160, 256), (50001, 256, 160), (33000, 768, 32),
                                   (25000, 1024, 1024), (24577, 288, 64)])
def test_gemm_x3s_row_streaming_kernel(bkm, M, N, K):
    # gemm_x3s.hip (round 6): the kernel the network's x W^T / dy W products take at bench-sized batches - both operands by LDS-DMA, 256 x 128
    # tiles, eight waves that split the rows, epilogue from the registers.  Shapes of the network (pre-rnn, gates, d(xcat), heads, dH), row
    # counts that are / are not a multiple of 8 x 256 (the XCD item map / the plain one), ragged last row tile, a column count that is not a
    # multiple of 128 (the PARTIAL form), K = 32 (one stage per item) .. 1024, padded leading dimensions; bias, relu, mask.  Bar: 2e-6 of
    # max |C| against f64 (the 128 x 128 kernel's), and the two kernels agree to 1e-6 (same pieces, same three MFMAs, another summation order)
    from dotaclient_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(M + 3 * N + K + int(bkm))
    lda, ldb, ldc = K + 4, (N if bkm else K) + 8, N + 4
    A = torch.randn(M, lda, generator=g).to(dev)
    B = (torch.randn((K if bkm else N), ldb, generator=g) / 16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    aux = torch.randn(M, N, generator=g).to(dev)
    ref = A[:, :K].double() @ (B[:, :N] if bkm else B[:, :K].t()).double() + bias.double()
    scale = ref.abs().max().item()
    scratch = torch.empty(2 * N * K + 1024, device=dev)
    outs = {}
    for tile128 in (False, True):
        prec = ops.prec_f16x2(4, 8, tile128=tile128)
        C = torch.full((M, ldc), 7.0, device=dev)
        ops.gemm(A, B, C, M, N, K, lda, ldb, ldc, False, bkm, bias=bias, scratch=scratch, x3=prec)
        assert torch.all(C[:, N:] == 7.0), 'wrote outside the N columns'
        assert ((C[:, :N].double() - ref).abs().max().item() / scale) < 2e-6
        outs[tile128] = C[:, :N].clone()
        C2 = torch.empty(M, N, device=dev)
        ops.gemm(A, B, C2, M, N, K, lda, ldb, N, False, bkm, bias=bias, relu=True, scratch=scratch, x3=prec)
        assert ((C2.double() - ref.clamp(min=0)).abs().max().item() / scale) < 2e-6
        ops.gemm(A, B, C2, M, N, K, lda, ldb, N, False, bkm, bias=bias, aux=aux, ldaux=N, scratch=scratch, x3=prec)
        assert ((C2.double() - ref * (aux > 0)).abs().max().item() / scale) < 2e-6
        ops.gemm(A, B, C2, M, N, K, lda, ldb, N, False, bkm, scratch=scratch, x3=prec)          # no bias
        assert ((C2.double() - (ref - bias.double())).abs().max().item() / scale) < 2e-6
    assert ((outs[False] - outs[True]).abs().max().item() / scale) < 1e-6


def test_gemm_x3s_repeated_launches_are_bit_identical():
    # the DMA pipeline of gemm_x3s.hip is synchronised by hand (counted vmcnt + one barrier per stage): a race shows up as run-to-run
    # differences.  Forty launches of the gates' product on fresh outputs, under a concurrent copy stream, must all be bit-identical.
    from dotaclient_amd import ops
    dev = _dev()
    M, N, K = 65536, 1024, 256
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(dev)
    B = (torch.randn(N, K, generator=g) / 16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    scratch = torch.empty(2 * N * K + 1024, device=dev)
    first = None
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, device=dev)
    for it in range(40):
        C = torch.full((M, N), float(it), device=dev)
        with torch.cuda.stream(side):
            junk.add_(1.0)                      # HBM traffic from another stream while the product runs
        ops.gemm(A, B, C, M, N, K, K, K, N, False, False, bias=bias, scratch=scratch, x3=ops.prec_f16x2(4, 8))
        if first is None:
            first = C.clone()
        else:
            assert torch.equal(C, first), it
    torch.cuda.synchronize()


@pytest.mark.parametrize('form', ['x W^T', 'dy W', 'dy^T x', 'dy^T x (both stored)', 'dy^T x (B stored)'])
@pytest.mark.parametrize('M,N,K', [(1000, 256, 896), (130, 72, 160), (2048, 768, 256), (152, 256, 8192), (33, 12, 16)])
def test_gemm_x3_bf16_storage(form, M, N, K):
    # BASELINE.json configs[4]'s path keeps its gate buffers in HBM as bf16 (policy.hip bf16_store()): the prec-1 products then read an
    # operand with the loader that does no arithmetic (row-major: as one pre-split plane; k-major: pairs along k packed by two v_perm)
    # and may write C / read the relu mask as bf16.  Against f64 on the SAME bf16-rounded operands the only error left is the f32
    # accumulation (1e-5 of max |C| up to K = 8192); a bf16 C must be that result rounded once (<= 2^-8 relative per entry).
    from dotaclient_amd import ops
    dev = _dev()
    akm = form.startswith('dy^T x')
    bkm = form != 'x W^T'
    a_st = form != 'dy^T x (B stored)'
    b_st = form in ('dy^T x (both stored)', 'dy^T x (B stored)')
    g = torch.Generator().manual_seed(M + 3 * N + 5 * K + len(form))
    lda = ((M + 7) // 8 * 8 if akm else K) + 8          # bf16 rows are read 16 / 8 bytes at a time
    ldb = (N if bkm else K) + 8
    A = torch.randn((K if akm else M), lda, generator=g)
    B = torch.randn((K if bkm else N), ldb, generator=g) / 16.0
    bias = torch.randn(N, generator=g)
    A16, B16 = A.bfloat16(), B.bfloat16()
    Am = (A16[:, :M].t() if akm else A16[:, :K]).double()
    Bm = (B16[:, :N] if bkm else B16[:, :K].t()).double()
    tn = akm and bkm
    ref = Am @ Bm + (0 if tn else bias.double())
    scale = ref.abs().max().item()
    scratch = torch.empty(max(2 * N * K, 16 * M * N) + 1024, device=dev)
    Ad = A16.to(dev) if a_st else A.to(dev)
    Bd = B16.to(dev) if b_st else B.to(dev)
    ldc = N + 4
    C = torch.full((M, ldc), 7.0, device=dev)
    ops.gemm(Ad, Bd, C, M, N, K, lda, ldb, ldc, akm, bkm, bias=None if tn else bias.to(dev), scratch=scratch,
             x3=ops.prec_bf16_store(a=a_st, b=b_st))
    out = C.cpu()
    assert torch.all(out[:, N:] == 7.0), 'wrote outside the N columns'
    err = (out[:, :N].double() - ref).abs().max().item() / scale
    assert err < 1e-5, err
    if tn:                                               # accumulate into live data through the split-K reduce
        C3 = torch.full((M, N), 2.0, device=dev)
        ops.gemm(Ad, Bd, C3, M, N, K, lda, ldb, N, True, True, accumulate=True, scratch=scratch, x3=ops.prec_bf16_store(a=a_st, b=b_st))
        assert (C3.cpu().double() - (ref + 2)).abs().max() / scale < 1e-5
        return
    # bf16 output (+ relu), and the relu mask read as bf16 (values around zero: the sign is what matters, bf16 keeps it)
    C16 = torch.full((M, ldc), 7.0, device=dev, dtype=torch.bfloat16)
    ops.gemm(Ad, Bd, C16, M, N, K, lda, ldb, ldc, akm, bkm, bias=bias.to(dev), relu=True, scratch=scratch, x3=ops.prec_bf16_store(a=True, c=True))
    o16 = C16.cpu().float()
    assert torch.all(o16[:, N:] == 7.0), 'wrote outside the N columns'
    want = ref.clamp(min=0)
    assert ((o16[:, :N].double() - want).abs() <= want.abs() * 2.0 ** -8 + 1e-5 * scale).all()
    aux = torch.randn(M, N, generator=g)
    C2 = torch.empty(M, N, device=dev)
    ops.gemm(Ad, Bd, C2, M, N, K, lda, ldb, N, akm, bkm, bias=bias.to(dev), aux=aux.bfloat16().to(dev), ldaux=N, scratch=scratch,
             x3=ops.prec_bf16_store(a=True, aux=True))
    assert (C2.cpu().double() - ref * (aux > 0)).abs().max() / scale < 1e-5


# ---- the default arithmetic at its edges (VERDICT r4 missing 3): prec 4 with the pre-scales policy.hip passes --------------------------
# activations 2^4 (limit 65504 / 16 = 4094), weights 2^8 (limit 255.9), gradients 2^(ceil(log2 rows) + 2) (limit ~16384 / rows).
F16_MAX = 65504.0
FORMS = {                                   # name: (a_kmajor, b_kmajor, class of A, class of B) - the three products of every nn.Linear
    'x W^T': (False, False, 'act', 'w'),    # policy.py:54-75,138-155 forward
    'dy W': (False, True, 'grad', 'w'),     # autograd: input gradient
    'dy^T x': (True, True, 'grad', 'act'),  # autograd: weight gradient (split-K)
}
ROWS = 65536                                # the batch the gradient pre-scale is derived from (configs[2])
LOG2 = {'act': 4, 'w': 8, 'grad': 16 + 2}


def _operand(cls, shape, g):
    if cls == 'act':                        # relu / tanh outputs and unit-variance inputs, a few large entries
        return torch.randn(shape, generator=g) * (1.0 + 30.0 * (torch.rand(shape, generator=g) < 0.01))
    if cls == 'w':                          # U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like nn.Linear's init
        return (torch.rand(shape, generator=g) * 2 - 1) / 16.0
    return torch.randn(shape, generator=g) / ROWS        # gradients of a mean loss over ROWS env-steps


def _x3_product(A, B, M, N, K, akm, bkm, prec):
    from dotaclient_amd import ops
    dev = _dev()
    C = torch.full((M, N), 7.0, device=dev)
    scratch = torch.empty(max(2 * N * K, 16 * M * N) + 1024, device=dev)
    ops.gemm(A.to(dev), B.to(dev), C, M, N, K, A.shape[1], B.shape[1], N, akm, bkm, scratch=scratch, x3=prec)
    return C.cpu()


def _ref(A, B, akm, bkm):
    return (A.t() if akm else A).double() @ (B if bkm else B.t()).double()


@pytest.mark.parametrize('form', list(FORMS))
@pytest.mark.parametrize('M,N,K', [(1000, 256, 896), (130, 72, 160), (2048, 768, 256), (152, 256, 8192)])
def test_gemm_x3_f16x2_operand_classes_with_their_prescales(form, M, N, K):
    # every product class of the network at its own pre-scales, ragged M / N, against f64 at the f32 kernel's bar
    from dotaclient_amd import ops
    akm, bkm, ca, cb = FORMS[form]
    if akm:
        M = (M + 3) // 4 * 4
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + len(form))
    A = _operand(ca, (K, M) if akm else (M, K), g)
    B = _operand(cb, (K, N) if bkm else (N, K), g)
    ref = _ref(A, B, akm, bkm)
    out = _x3_product(A, B, M, N, K, akm, bkm, ops.prec_f16x2(LOG2[ca], LOG2[cb]))
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < (5e-6 if K > 4096 else 2e-6), (form, err)


@pytest.mark.parametrize('form', list(FORMS))
def test_gemm_x3_f16x2_at_the_range_edge_and_beyond(form):
    # 0.9 x the documented limit of each operand class: still f32-grade.  Just over it: the entry becomes inf in its first f16 piece and
    # every output that depends on it is NON-FINITE (-> NaN loss -> the reference's own NaN guard -> Engine.use_safe_products), the rest
    # of C stays accurate: never a silently wrong finite number.
    from dotaclient_amd import ops
    akm, bkm, ca, cb = FORMS[form]
    M, N, K = 256, 128, 512
    g = torch.Generator().manual_seed(11 + len(form))
    A0 = _operand(ca, (K, M) if akm else (M, K), g)
    B0 = _operand(cb, (K, N) if bkm else (N, K), g)
    prec = ops.prec_f16x2(LOG2[ca], LOG2[cb])
    lim_a, lim_b = F16_MAX / 2.0 ** LOG2[ca], F16_MAX / 2.0 ** LOG2[cb]
    m, n, k = 37, 19, 301
    for which, factor in [('a', 0.9), ('b', 0.9), ('a', 1.01), ('b', 1.01)]:
        A, B = A0.clone(), B0.clone()
        if which == 'a':
            A[(k, m) if akm else (m, k)] = -factor * lim_a
        else:
            B[(k, n) if bkm else (n, k)] = factor * lim_b
        out = _x3_product(A, B, M, N, K, akm, bkm, prec)
        ref = _ref(A, B, akm, bkm)
        if factor < 1:
            assert torch.isfinite(out).all(), (form, which)
            assert (out.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6, (form, which)
        else:
            hit = torch.zeros(M, N, dtype=torch.bool)
            if which == 'a':
                hit[m, :] = True
            else:
                hit[:, n] = True
            assert not torch.isfinite(out[hit]).any(), (form, which, 'an out-of-range operand produced finite outputs')
            assert torch.isfinite(out[~hit]).all()
            ref0 = _ref(A0, B0, akm, bkm)
            assert (out[~hit].double() - ref0[~hit]).abs().max().item() / ref0.abs().max().item() < 2e-6


@pytest.mark.parametrize('form', ['dy W', 'dy^T x'])
def test_gemm_x3_f16x2_small_gradients_second_piece_subnormal(form):
    # gradient entries of 1e-9 .. 1e-7 at rows = 65 536: after the 2^18 pre-scale their FIRST f16 piece is normal, the second falls into
    # f16's subnormal band (absolute resolution 2^-24 / 2^18 = 2.3e-13 per entry instead of 2^-23 relative).  The product must still be
    # f32-grade against the largest result, and no worse than ~2^-12 relative even for an all-tiny operand.
    from dotaclient_amd import ops
    akm, bkm, ca, cb = FORMS[form]
    M, N, K = 256, 256, 4096 if akm else 256
    g = torch.Generator().manual_seed(5 + len(form))
    shape = (K, M) if akm else (M, K)
    mag = 10.0 ** (-9.0 + 2.0 * torch.rand(shape, generator=g))
    A = mag * (torch.randint(0, 2, shape, generator=g) * 2 - 1)
    B = _operand(cb, (K, N) if bkm else (N, K), g)
    ref = _ref(A, B, akm, bkm)
    out = _x3_product(A, B, M, N, K, akm, bkm, ops.prec_f16x2(LOG2['grad'], LOG2[cb]))
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 5e-6, (form, err)
    # mixed with ordinary gradient entries (the realistic case: a few heads act rarely, most entries are O(1 / rows)): f32-grade
    A2 = torch.where(torch.rand(shape, generator=g) < 0.5, A, _operand('grad', shape, g))
    ref2 = _ref(A2, B, akm, bkm)
    out2 = _x3_product(A2, B, M, N, K, akm, bkm, ops.prec_f16x2(LOG2['grad'], LOG2[cb]))
    assert (out2.double() - ref2).abs().max().item() / ref2.abs().max().item() < 2e-6


def test_gemm_epilogues():
    from dotaclient_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 128, 128
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    aux = torch.randn(M, N, generator=g)
    ref = A.double() @ B.double().t()
    C = torch.empty(M, N, device=dev)
    ops.gemm(A.to(dev), B.to(dev), C, M, N, K, K, K, N, relu=True)
    assert (C.cpu().double() - ref.clamp(min=0)).abs().max() < 1e-4
    ops.gemm(A.to(dev), B.to(dev), C, M, N, K, K, K, N, aux=aux.to(dev), ldaux=N)
    assert (C.cpu().double() - ref * (aux > 0)).abs().max() < 1e-4
    C.fill_(1.0)
    ops.gemm(A.to(dev), B.to(dev), C, M, N, K, K, K, N, accumulate=True)
    assert (C.cpu().double() - (ref + 1)).abs().max() < 1e-4
    # split-K with accumulate into live data
    K2 = 4096
    A2, B2 = torch.randn(K2, 96, generator=g), torch.randn(K2, 80, generator=g)
    C2 = torch.full((96, 80), 2.0, device=dev)
    ops.gemm(A2.to(dev), B2.to(dev), C2, 96, 80, K2, 96, 80, 80, True, True, accumulate=True, splits=8)
    ref2 = A2.double().t() @ B2.double() + 2
    assert (C2.cpu().double() - ref2).abs().max() / ref2.abs().max() < 2e-6


def _adam_reference(p, g, m, v, segs, gates, head_on, vf_coef, step, lr, loss):
    """optimizer.py:674-681 on flat f32 arrays (numpy, f32 arithmetic like torch's): per-parameter norms over the segments that have a
    gradient, clip_grad_norm_(0.5), Adam(betas 0.9/0.999, eps 1e-8); the NaN guards leave everything alone."""
    act = [gt < 0 or (gt < 5 and head_on[gt]) or (gt == 5 and vf_coef > 0) for gt in gates]
    norms = np.array([np.sqrt(np.sum(g[o:o + n].astype(np.float64) ** 2)) for (o, n) in segs], np.float64).astype(np.float32)
    on = [i for i, a in enumerate(act) if a]
    unclipped = np.float32(np.mean(norms[on].astype(np.float64)))
    total = np.float32(np.sqrt(np.sum(norms[on].astype(np.float64) ** 2)))
    coef = np.float32(min(1.0, 0.5 / (float(total) + 1e-6)))
    p, g, m, v = p.copy(), g.copy(), m.copy(), v.copy()
    if np.isnan(loss) or not np.isfinite(total):
        return p, g, m, v, unclipped, coef, (1 if np.isnan(loss) else 2), step.copy()
    step = step.copy()
    for i in on:
        o, n = segs[i]
        step[i] += 1
        gj = g[o:o + n] * coef
        g[o:o + n] = gj
        m[o:o + n] = m[o:o + n] + np.float32(1.0 - 0.9) * (gj - m[o:o + n])
        v[o:o + n] = v[o:o + n] * np.float32(0.999) + np.float32(1.0 - 0.999) * gj * gj
        bc1, bc2 = 1.0 - 0.9 ** int(step[i]), 1.0 - 0.999 ** int(step[i])
        denom = np.sqrt(v[o:o + n]) / np.float32(np.sqrt(bc2)) + np.float32(1e-8)
        p[o:o + n] = p[o:o + n] - np.float32(lr / bc1) * m[o:o + n] / denom
    return p, g, m, v, unclipped, coef, 0, step


@pytest.mark.parametrize('case', ['small', 'ragged', 'beyond_one_chunk_per_block', 'nan_loss', 'inf_grad'])
def test_gradnorm_clip_adam_single_launch(case):
    # the one-launch norm -> clip -> Adam kernel (csrc/adam.hip) against optimizer.py:674-681 restated in numpy: segments that are not
    # multiples of the 4096-element chunk, heads without a gradient, a parameter count beyond ADAM_GRID x 4096 (blocks with several
    # chunks: only the first stays in registers), both NaN guards (nothing changes), and - every case - three calls in a row: the arrival
    # counters must be back at zero after each ('small' has no room for the first ticket level: one counter)
    from dotaclient_amd import _lib
    dev = _dev()
    lib = _lib.load()
    rng = np.random.Generator(np.random.PCG64(11))
    if case == 'beyond_one_chunk_per_block':
        lens = [1_500_000, 700_001, 4096, 5, 262_144, 333]
    elif case == 'small':
        lens = [7, 1, 300, 4096, 4097]
    else:
        lens = [262_144, 1024, 65_536, 1024, 40_960, 160, 12_288, 128, 16_384, 4096 * 3 + 1, 26, 2]
    gates = [(-1, 0, 3, 5, 4, -1, 1, 2)[i % 8] for i in range(len(lens))]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    segs = list(zip(offs.tolist(), lens))
    n = int(sum(lens))
    head_on = np.array([1, 0, 1, 1, 0], np.int32)
    vf = 0.5
    p = rng.standard_normal(n).astype(np.float32)
    m = (0.01 * rng.standard_normal(n)).astype(np.float32)
    v = (1e-4 * rng.random(n)).astype(np.float32)
    step = np.arange(len(lens), dtype=np.int32) % 3
    t = lambda a: torch.from_numpy(a).to(dev)
    P, M, V, STEP = t(p), t(m), t(v), t(step)
    nchunk = (max(lens) + 4095) // 4096
    segsq = torch.zeros(len(lens) * (1 + nchunk), dtype=torch.float64, device=dev)
    ctl = torch.zeros(4, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    norms = torch.zeros(2, dtype=torch.float32, device=dev)
    d_off, d_len, d_gate, d_on = t(offs), t(np.array(lens, np.int32)), t(np.array(gates, np.int32)), t(head_on)
    for it in range(3):
        g = (rng.standard_normal(n) * (1e-3 if it == 1 else 1e-1)).astype(np.float32)      # (it 1: norm below the clip, coef = 1)
        loss = np.float32(0.3)
        if case == 'nan_loss' and it == 1: loss = np.float32('nan')
        if case == 'inf_grad' and it == 1: g[5] = np.float32(3e38); g[7] = np.float32(3e38)
        G = t(g)
        before = [x.clone() for x in (P, M, V, G, STEP)]
        losses = t(np.array([loss] + [0] * 8, np.float32))
        _lib.check(lib.dc_gradnorm_clip_adam(_lib.ptr(d_off), _lib.ptr(d_len), _lib.ptr(d_gate), len(lens), max(lens), _lib.ptr(P), _lib.ptr(G),
                                             _lib.ptr(M), _lib.ptr(V), _lib.ptr(segsq), _lib.ptr(d_on), _lib.ptr(losses), _lib.ptr(norms),
                                             _lib.ptr(ctl), _lib.ptr(STEP), _lib.ptr(status), 0.5, vf, 1e-3, 0.9, 0.999, 1e-8,
                                             _lib.stream_ptr()), 'dc_gradnorm_clip_adam')
        torch.cuda.synchronize()
        st_before = int(status.item())
        ep, eg, em, ev, unclipped, coef, st, estep = _adam_reference(p, g, m, v, segs, gates, head_on, vf, step, 1e-3, loss)
        raw = ctl.cpu().numpy().view(np.uint32)
        assert raw[2] == 0 and raw[3] == it + 1, (case, it, raw)          # arrival counter back at zero, one release generation per call
        assert float(segsq[len(lens) + sum((n + 4095) // 4096 for n in lens):].abs().sum()) == 0.0, (case, it)      # first-level counters too
        if st_before == 0 and st == 0:
            assert abs(float(norms[0]) - float(unclipped)) <= 1e-6 * float(unclipped), (case, it)
            assert abs(float(ctl[0]) - float(coef)) <= 2e-6 * float(coef), (case, it, float(ctl[0]), float(coef))
            for name, got, exp in (('p', P, ep), ('g', G, eg), ('m', M, em), ('v', V, ev)):
                assert util.scaled_err(exp, got.cpu().numpy()) < 2e-6, (case, it, name)
            assert np.array_equal(STEP.cpu().numpy(), estep), (case, it)
            p, m, v, step = ep, em, ev, estep
        else:
            # a guard tripped (this call or, sticky, an earlier one): parameters, moments, gradients and step counters untouched
            assert st_before == (1 if case == 'nan_loss' else 2), (case, it, st_before)
            assert all(torch.equal(x, y) for x, y in zip((P, M, V, G, STEP), before)), (case, it)
