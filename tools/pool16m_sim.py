"""Lane-level model of csrc/embed_pool16m.hip (the max-pool backward of the two 16-unit types as dense products on the f16 matrix cores
with every operand generated on chip).  Checks the INDEX MATH of the kernel - which lane builds which operand element, which accumulator
register holds which output - against a dense float64 evaluation of the same gradient, with v_mfma_f32_32x32x16_f16 modelled by its
register layout (the one csrc/embed_fused.hip documents and uses):
    A: lane (fr = lane & 31, fq = lane >> 5) holds A[fr][8 fq + j], j = 0..7;   B: lane (fr, fq) holds B[8 fq + j][fr]
    D: lane (fr, fq), register r holds D[8 (r >> 2) + 4 fq + (r & 3)][fr]
No rounding is modelled (float64 throughout): the f16 piece arithmetic is gemm_x3's, tested on the GPU.  CPU only:  python tools/pool16m_sim.py"""
import numpy as np

LANES = np.arange(64)
FR, FQ = LANES & 31, LANES >> 5


def mfma(A, B, D):
    """A, B [64 lanes][8], D [64][16] -> D + A x B in the 32x32x16 layout."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        Am[FR[l], 8 * FQ[l]:8 * FQ[l] + 8] = A[l]
        Bm[8 * FQ[l]:8 * FQ[l] + 8, FR[l]] = B[l]
    P = Am @ Bm
    out = D.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += P[8 * (r >> 2) + 4 * FQ[l] + (r & 3), FR[l]]
    return out


def sigma(fq, j):            # K slot j of lane group fq <-> unit (the order the first layer's D registers hold the units in)
    return 4 * fq + (j & 3) + 8 * (j >> 2)


def kernel(x, W1, b1, W2, amax, d, dtu, q):
    """x [n][16][12], W1 [128][12], b1 [128], W2 [c 128][k 128], amax [n][128] (unit per channel), d [n][128], dtu [n][16], q [n][128]
    -> dW2 [c][k], part1 [13][128] (dW1^T rows 0..11, db1 row 12), db2 [128]."""
    n = x.shape[0]
    dW2 = np.zeros((128, 128)); part1 = np.zeros((13, 128)); db2 = np.zeros(128)
    pairs = [(2 * p, 2 * p + 1) for p in range((n + 1) // 2)]
    for W in range(8):
        kq, st = W & 3, W >> 2
        acc = [np.zeros((64, 16)) for _ in range(4)]       # phase B: [cb]
        facc = np.zeros((64, 16))
        db2acc = np.zeros((64, 4))
        # constant operand: W1 rows of this wave's k block, lane (col = fr, fq): W1[32 kq + fr][8 fq + j]
        w1 = np.zeros((64, 8))
        for l in range(64):
            for j in range(8):
                f = 8 * FQ[l] + j
                w1[l, j] = W1[32 * kq + FR[l], f] if f < 12 else 0.0
        for pi in range(st, len(pairs), 2):
            items = pairs[pi]
            valid = [it < n for it in items]
            itc = [min(it, n - 1) for it in items]
            # ---- first layer: rows = (e, u)
            xa = np.zeros((64, 8))
            for l in range(64):
                e, u = FR[l] >> 4, FR[l] & 15
                for j in range(8):
                    f = 8 * FQ[l] + j
                    xa[l, j] = x[itc[e], u, f] if f < 12 else 0.0
            g = mfma(xa, w1, np.zeros((64, 16)))
            basic = np.maximum(g + b1[32 * kq + FR][:, None], 0.0)          # reg r: row 8(r>>2)+4fq+(r&3), k = 32 kq + fr
            # ---- phase B, per item e: the operand is ONE-HOT (a select of the staged d pieces); the rank-one attention term goes through K slot 0
            sk = [None, None]
            for e in range(2):
                A = basic[:, 8 * e:8 * e + 8]                                 # K slot j <-> unit sigma(fq, j): exactly registers 8 e + j
                # s[k] = sum_u dtu[u] basic[u][k]: own eight units + lane ^ 32's (zero for a step whose head is off, or an absent item)
                part = np.array([sum(dtu[itc[e], sigma(FQ[l], j)] * A[l, j] for j in range(8)) for l in range(64)]) * (1.0 if valid[e] else 0.0)
                sk[e] = part + part[LANES ^ 32]
                for cb in range(4):
                    Bop = np.zeros((64, 8))
                    for l in range(64):
                        c = 32 * cb + FR[l]
                        a = amax[itc[e], c]
                        jj = (a & 3) + 4 * (a >> 3)
                        if ((a >> 2) & 1) == FQ[l] and valid[e]:
                            Bop[l, jj] = d[itc[e], c]
                    acc[cb] = mfma(A, Bop, acc[cb])
                if kq == 0 and valid[e]:                                      # (the kernel sums these when it stages the item: channels 2 lane, 2 lane + 1)
                    for l in range(64):
                        if FQ[l] == 0:
                            for cb in range(4):
                                c = 32 * cb + FR[l]
                                db2acc[l, cb] += d[itc[e], c] + q[itc[e], c] * dtu[itc[e]].sum()
            for cb in range(4):                                               # rank-one attention term of both items: K slots 0, 1 of lane group 0
                A1 = np.zeros((64, 8)); B1 = np.zeros((64, 8))
                for l in range(64):
                    if FQ[l] == 0:
                        for e in range(2):
                            A1[l, e] = sk[e][l]
                            B1[l, e] = q[itc[e], 32 * cb + FR[l]]
                acc[cb] = mfma(A1, B1, acc[cb])
            # ---- phase C: 8 K steps of 16 channels, one-hot rows; then + dtu[u] R[k] in the accumulators (R = q W2: a dense product of its own)
            cacc = np.zeros((64, 16))
            for ks in range(8):
                A = np.zeros((64, 8)); Bop = np.zeros((64, 8))
                for l in range(64):
                    e, u = FR[l] >> 4, FR[l] & 15
                    for j in range(8):
                        c = 16 * ks + 8 * FQ[l] + j
                        A[l, j] = d[itc[e], c] if (amax[itc[e], c] == u and valid[e]) else 0.0
                        Bop[l, j] = W2[c, 32 * kq + FR[l]]                   # LDS image [ks][fq][k][j]
                cacc = mfma(A, Bop, cacc)
            for e in range(2):
                R = q[itc[e]] @ W2                                            # [128]
                for l in range(64):
                    for j in range(8):
                        if valid[e]:
                            cacc[l, 8 * e + j] += dtu[itc[e], sigma(FQ[l], j)] * R[32 * kq + FR[l]]
            dbm = np.where(basic > 0, cacc, 0.0)
            # ---- fold: K step e, slots <-> sigma
            for e in range(2):
                A = np.zeros((64, 8))
                for l in range(64):
                    f = FR[l]
                    for j in range(8):
                        un = sigma(FQ[l], j)
                        A[l, j] = x[itc[e], un, f] if f < 12 else (1.0 if f == 12 else 0.0)
                facc = mfma(A, dbm[:, 8 * e:8 * e + 8], facc)
        # ---- results of this wave (the two streams are summed through LDS in the kernel: here by +=)
        for l in range(64):
            for cb in range(4):
                for r in range(16):
                    dW2[32 * cb + FR[l], 32 * kq + 8 * (r >> 2) + 4 * FQ[l] + (r & 3)] += acc[cb][l, r]
            for r in range(16):
                f = 8 * (r >> 2) + 4 * FQ[l] + (r & 3)
                if f < 13:
                    part1[f, 32 * kq + FR[l]] += facc[l, r]
            if kq == 0 and FQ[l] == 0:
                for cb in range(4):
                    db2[32 * cb + FR[l]] += db2acc[l, cb]
    return dW2, part1, db2


def reference(x, W1, b1, W2, amax, d, dtu, q):
    n = x.shape[0]
    dW2 = np.zeros((128, 128)); part1 = np.zeros((13, 128)); db2 = np.zeros(128)
    for i in range(n):
        basic = np.maximum(x[i] @ W1.T + b1, 0.0)                               # [16][128]
        demb = np.zeros((16, 128))
        demb[amax[i], np.arange(128)] = d[i]
        demb += np.outer(dtu[i], q[i])
        dW2 += demb.T @ basic
        dbm = np.where(basic > 0, demb @ W2, 0.0)
        part1[:12] += x[i].T @ dbm
        part1[12] += dbm.sum(0)
        db2 += demb.sum(0)
    return dW2, part1, db2


if __name__ == '__main__':
    rng = np.random.default_rng(3)
    for n in (5, 8):
        x = rng.standard_normal((n, 16, 12)); W1 = rng.standard_normal((128, 12)) * 0.3; b1 = rng.standard_normal(128) * 0.3
        W2 = rng.standard_normal((128, 128)) * 0.1; amax = rng.integers(0, 16, (n, 128)); d = rng.standard_normal((n, 128))
        dtu = rng.standard_normal((n, 16)) * (rng.random((n, 1)) < 0.5); q = rng.standard_normal((n, 128))
        got, ref = kernel(x, W1, b1, W2, amax, d, dtu, q), reference(x, W1, b1, W2, amax, d, dtu, q)
        for name, a, b in zip(('dW2', 'part1', 'db2'), got, ref):
            err = np.abs(a - b).max() / np.abs(b).max()
            print(n, name, 'max rel err %.2e' % err)
            assert err < 1e-12, name
    print('index math OK')
