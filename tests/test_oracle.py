"""CPU: pins the oracle restatement (oracle/) to the golden vectors produced by the real reference
(tests/golden/make_golden.py).  No GPU, no /root/reference needed."""
import numpy as np
import pytest

from oracle import ref_optimizer as RO
from tests import util


def test_gae_known_answer():
    # SURVEY.md section 4 / reference advantage_returns(r=[1,2,3,0], v=[.5,.4,.3,0], .98, .97)
    adv, ret = RO.advantage_returns(np.array([1, 2, 3, 0], np.float32), np.array([.5, .4, .3, 0], np.float32), 0.98, 0.97)
    np.testing.assert_allclose(adv, [5.1322656, 4.4606204, 2.7], rtol=1e-6)
    np.testing.assert_allclose(ret, [5.8412, 4.94, 3.0], rtol=1e-6)


def test_gae_golden_bit_exact():
    g = np.load(util.GOLDEN + '/gae_kat.npz')
    adv, ret = RO.advantage_returns(g['r2'], g['v2'], 0.98, 0.97)
    assert np.array_equal(adv, g['adv2']) and np.array_equal(ret, g['ret2'])


def test_gae_golden_nonzero_terminals_and_discount():
    # optimizer.py:53-64 accept any vectors: terminal reward / bootstrap value != 0, and discount on its own
    g = np.load(util.GOLDEN + '/gae_kat.npz')
    adv, ret = RO.advantage_returns(g['r3'], g['v3'], 0.98, 0.97)
    assert g['r3'][-1] != 0 and g['v3'][-1] != 0
    assert np.array_equal(adv, g['adv3']) and np.array_equal(ret, g['ret3'])
    assert np.array_equal(RO.discount(g['x4'], 0.98), g['disc4'])
    assert np.array_equal(RO.discount(g['x4'][:65], 0.98 * 0.97), g['disc4b'])


def test_oracle_dp_emulation_matches_reference_wrapper():
    # the N-rank emulation (RO.dp_train_step / dp_average_grads) against the reference's own
    # DistributedDataParallelSparseParamCPU (distributed.py:16-79) run under two gloo ranks by make_golden.py; rank 1
    # never uses the ability head, so its has-grad count is 1 and rank 1 keeps grad = None for it
    g, shards = util.load_dp_case()
    out = util.oracle_dp_run(g, shards)
    for r in range(int(g['world'])):
        for key in ('advantages', 'returns'):
            assert util.scaled_err(out['r%d_%s' % (r, key)], g['r%d_%s' % (r, key)]) < 1e-5
        for ep in range(int(g['epochs'])):
            pre = 'r%d_ep%d_' % (r, ep)
            for key in ('losses', 'entropies', 'grad_norms'):
                assert util.rel_err(out[pre + key], g[pre + key]) < 2e-5, (pre + key, out[pre + key], g[pre + key])
            assert np.array_equal(out[pre + 'has_grad'], g[pre + 'has_grad'])
            assert util.scaled_err(out[pre + 'grad_samples'], g[pre + 'grad_samples']) < 1e-4
            assert util.scaled_err(out[pre + 'param_samples'], g[pre + 'param_samples']) < 1e-5
    assert not g['r1_ep0_has_grad'].all() and g['r0_ep0_has_grad'].all()


@pytest.mark.parametrize('case', util.CASES + util.BIG_CASES)
def test_oracle_matches_reference(case):
    g, rollouts = util.load_case(case)
    out, _, _ = util.oracle_run(g, rollouts)
    assert list(out['param_names']) == list(g['param_names'])
    # same torch build generated the fixtures here, so the restatement is expected to agree to
    # the last few ulps; 1e-5 leaves room for op-order differences (einsum vs matmul)
    for key in ['advantages', 'returns', 'values'] + ['old_logp_' + k for k in RO.HEADS]:
        assert util.scaled_err(out[key], g[key]) < 1e-5, key
    assert np.array_equal(out['argmax'], g['argmax'])
    for ep in range(int(g['epochs'])):
        for key in ['losses', 'entropies', 'grad_norms']:
            k = 'ep%d_%s' % (ep, key)
            assert util.rel_err(out[k], g[k]) < 2e-5, (k, out[k], g[k])
        assert np.array_equal(out['ep%d_has_grad' % ep], g['ep%d_has_grad' % ep])
        assert util.scaled_err(out['ep%d_param_samples' % ep], g['ep%d_param_samples' % ep]) < 1e-5
        assert util.scaled_err(out['ep%d_grad_samples' % ep], g['ep%d_grad_samples' % ep]) < 1e-4


def test_c_oracle_matches_numpy_oracle_and_golden():
    from oracle import build_c
    g = np.load(util.GOLDEN + '/gae_kat.npz')
    r, v = g['r2'][:-1], g['v2'][:-1]
    rew = np.zeros((r.size, 10), np.float32); rew[:, 3] = r
    adv, ret = build_c.gae_ref(rew, v, [0], [r.size])
    assert np.array_equal(adv, g['adv2']) and np.array_equal(ret, g['ret2'])
    rng = np.random.Generator(np.random.PCG64(3))
    lens = [5, 300, 17]
    rew = (0.05 * rng.standard_normal((sum(lens), 10))).astype(np.float32)
    val = rng.standard_normal(sum(lens)).astype(np.float32)
    adv, ret = build_c.gae_ref(rew, val, [0, 5, 305], lens)
    o = 0
    for L in lens:
        a, t = RO.advantage_returns(np.append(rew[o:o + L].sum(axis=1), np.float32(0)),
                                    np.append(val[o:o + L], np.float32(0)), 0.98, 0.97)
        assert np.array_equal(adv[o:o + L], a) and np.array_equal(ret[o:o + L], t)
        o += L


def test_gae_long_rollouts_golden_sha256():
    # 50 000 steps (longer than one 20 480-step LDS block of the HIP scan): the oracle against the real reference's output, bit for
    # bit over every entry (SHA-256 of the float32 bytes, tests/golden/gae_long.npz)
    g = np.load(util.GOLDEN + '/gae_long.npz')
    r, v, x = util.gae_long_inputs(int(g['n']), int(g['seed']))
    r0, v0 = r.copy(), v.copy()
    r0[-1] = 0; v0[-1] = 0
    for tag, rr, vv in (('zero_terminal', r0, v0), ('any_terminal', r, v)):
        adv, ret = RO.advantage_returns(rr, vv, 0.98, 0.97)
        assert np.array_equal(adv[::997], g[tag + '_adv_samples']) and np.array_equal(ret[::997], g[tag + '_ret_samples'])
        assert np.array_equal(util.sha256_of(adv), g[tag + '_adv_sha256']) and np.array_equal(util.sha256_of(ret), g[tag + '_ret_sha256'])
    assert np.array_equal(util.sha256_of(RO.discount(x, 0.98 * 0.97)), g['discount_sha256'])
