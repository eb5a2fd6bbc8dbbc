"""Lane-level model of csrc/embed_pool16m.hip's UP = 8 variant (the 5-unit type, enemy heroes, padded to 8 unit slots: a 32-row tile = 4
env-steps x 8 slots) - the same check as tools/pool16m_sim.py for the other row <-> (step, unit) mapping:
    tile row rho: step = 2 (rho >> 4) + ((rho >> 2) & 1), unit slot = (rho & 3) + 4 ((rho >> 3) & 1)
so that the D registers 8 h + j of lane group fq hold (step 2 h + fq, unit slot j): K slot 8 fq + j of the unit-contracting products of half h
stands for exactly that, the one-hot table entry of a channel is its arg-max unit itself, and each lane group carries a whole step
(no exchange between the groups for the rank-one term).  CPU only:  python tools/pool8_sim.py"""
import numpy as np
from tools.pool16m_sim import mfma, FR, FQ, LANES

NU = 5          # real units of the type


def row_step_unit(rho):
    return 2 * (rho >> 4) + ((rho >> 2) & 1), (rho & 3) + 4 * ((rho >> 3) & 1)


def kernel(x, W1, b1, W2, amax, d, dtu, q):
    """x [n][5][12], amax [n][128] in 0..4, d [n][128], dtu [n][5], q [n][128] -> dW2 [c][k], part1 [13][128], db2 [128]."""
    n = x.shape[0]
    dW2 = np.zeros((128, 128)); part1 = np.zeros((13, 128)); db2 = np.zeros(128)
    n_tiles = (n + 3) // 4
    xp = np.zeros((n, 8, 12)); xp[:, :NU] = x                 # unit slots 5..7: zero records (staged as zeros)
    dtp = np.zeros((n, 8)); dtp[:, :NU] = dtu
    for W in range(12):
        kq, st = W & 3, W >> 2
        acc = [np.zeros((64, 16)) for _ in range(4)]
        facc = np.zeros((64, 16))
        w1 = np.zeros((64, 8))
        for l in range(64):
            for j in range(8):
                f = 8 * FQ[l] + j
                w1[l, j] = W1[32 * kq + FR[l], f] if f < 12 else 0.0
        for ti in range(st, n_tiles, 3):
            steps = [4 * ti + e for e in range(4)]
            valid = [s_ < n for s_ in steps]
            sc = [min(s_, n - 1) for s_ in steps]
            xa = np.zeros((64, 8))
            for l in range(64):
                e, u = row_step_unit(FR[l])
                for j in range(8):
                    f = 8 * FQ[l] + j
                    xa[l, j] = xp[sc[e], u, f] if f < 12 else 0.0
            basic = np.maximum(mfma(xa, w1, np.zeros((64, 16))) + b1[32 * kq + FR][:, None], 0.0)
            # ---- dW2: half h, lane group fq <-> step 2 h + fq, K slot j <-> unit slot j
            sk = np.zeros((2, 64))
            for h in range(2):
                A = basic[:, 8 * h:8 * h + 8]
                for l in range(64):
                    e = 2 * h + FQ[l]
                    sk[h, l] = sum(dtp[sc[e], j] * A[l, j] for j in range(8)) * (1.0 if valid[e] else 0.0)      # own group: a whole step
                for cb in range(4):
                    Bop = np.zeros((64, 8))
                    for l in range(64):
                        e = 2 * h + FQ[l]
                        c = 32 * cb + FR[l]
                        if valid[e]:
                            Bop[l, amax[sc[e], c]] = d[sc[e], c]          # table entry = the arg-max unit itself
                    acc[cb] = mfma(A, Bop, acc[cb])
            for cb in range(4):                                           # rank-one term: K slot (fq, h) <-> step 2 h + fq, both groups
                A1 = np.zeros((64, 8)); B1 = np.zeros((64, 8))
                for l in range(64):
                    for h in range(2):
                        A1[l, h] = sk[h, l]
                        B1[l, h] = q[sc[2 * h + FQ[l]], 32 * cb + FR[l]]
                acc[cb] = mfma(A1, B1, acc[cb])
            # ---- d(basic) (kernel 2 does this for all four k blocks; here: this wave's quarter) and the fold
            cacc = np.zeros((64, 16))
            for ks in range(8):
                A = np.zeros((64, 8)); Bop = np.zeros((64, 8))
                for l in range(64):
                    e, u = row_step_unit(FR[l])
                    for j in range(8):
                        c = 16 * ks + 8 * FQ[l] + j
                        A[l, j] = d[sc[e], c] if (amax[sc[e], c] == u and valid[e]) else 0.0
                        Bop[l, j] = W2[c, 32 * kq + FR[l]]
                cacc = mfma(A, Bop, cacc)
            for h in range(2):
                for l in range(64):
                    e = 2 * h + FQ[l]
                    R = q[sc[e]] @ W2
                    for j in range(8):
                        if valid[e]:
                            cacc[l, 8 * h + j] += dtp[sc[e], j] * R[32 * kq + FR[l]]
            dbm = np.where(basic > 0, cacc, 0.0)
            for h in range(2):
                A = np.zeros((64, 8))
                for l in range(64):
                    f = FR[l]
                    e = 2 * h + FQ[l]
                    for j in range(8):
                        A[l, j] = (xp[sc[e], j, f] if f < 12 else (1.0 if f == 12 else 0.0)) if j < NU else 0.0
                facc = mfma(A, dbm[:, 8 * h:8 * h + 8], facc)
        for l in range(64):
            for cb in range(4):
                for r in range(16):
                    dW2[32 * cb + FR[l], 32 * kq + 8 * (r >> 2) + 4 * FQ[l] + (r & 3)] += acc[cb][l, r]
            for r in range(16):
                f = 8 * (r >> 2) + 4 * FQ[l] + (r & 3)
                if f < 13:
                    part1[f, 32 * kq + FR[l]] += facc[l, r]
    for i in range(n):
        db2 += d[i] + q[i] * dtu[i].sum()
    return dW2, part1, db2


def reference(x, W1, b1, W2, amax, d, dtu, q):
    n = x.shape[0]
    dW2 = np.zeros((128, 128)); part1 = np.zeros((13, 128)); db2 = np.zeros(128)
    for i in range(n):
        basic = np.maximum(x[i] @ W1.T + b1, 0.0)                               # [5][128]
        demb = np.zeros((NU, 128))
        demb[amax[i], np.arange(128)] = d[i]
        demb += np.outer(dtu[i], q[i])
        dW2 += demb.T @ basic
        dbm = np.where(basic > 0, demb @ W2, 0.0)
        part1[:12] += x[i].T @ dbm
        part1[12] += dbm.sum(0)
        db2 += demb.sum(0)
    return dW2, part1, db2


def check(n, seed=5):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, NU, 12)); W1 = rng.standard_normal((128, 12)) * 0.3; b1 = rng.standard_normal(128) * 0.3
    W2 = rng.standard_normal((128, 128)) * 0.1; amax = rng.integers(0, NU, (n, 128)); d = rng.standard_normal((n, 128))
    dtu = rng.standard_normal((n, NU)) * (rng.random((n, 1)) < 0.5); q = rng.standard_normal((n, 128))
    got, ref = kernel(x, W1, b1, W2, amax, d, dtu, q), reference(x, W1, b1, W2, amax, d, dtu, q)
    return [np.abs(a - b).max() / np.abs(b).max() for a, b in zip(got, ref)]


if __name__ == '__main__':
    for n in (6, 13):
        errs = check(n)
        print(n, ['%.2e' % e for e in errs])
        assert max(errs) < 1e-12
    print('index math OK')
