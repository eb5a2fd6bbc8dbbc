"""GPU: end-to-end parity of the HIP hot path against the golden vectors of the real reference
(tests/golden/*.npz) and against the oracle restatement, through the C ABI."""
import numpy as np
import pytest
import torch

from dotaclient_amd import layout as L
from dotaclient_amd import synth
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-4          # BASELINE.json north_star: losses/advantages within 1e-4 relative


def run_hip(g, rollouts, cell='gru', hidden=256, layers=1, epochs=None, kernel_flags=0, reuse_forward=False, products=None):
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    eng = Engine(cell, hidden, layers, dev)
    eng.kernel_flags = kernel_flags
    eng.reuse_rollout_forward = reuse_forward
    if products is not None:
        eng.products = products          # None: the engine's default ('f16x2')
    eng.load_state_dict(synth.init_state_dict(7, cell, hidden, layers))
    S = int(g['seq_len'])
    batch = pack_rollouts(rollouts, S, dev)
    chunks = eng.rollout_pass(batch, S)
    B = chunks.n_seq
    out = {
        'advantages': batch.adv.view(B, S).cpu().numpy(),
        'returns': batch.ret.view(B, S).cpu().numpy(),
        'values': batch.values.view(B, S).cpu().numpy(),
        'argmax': batch.argmax.view(B, S, 5).cpu().numpy(),
        'hidden': chunks.h0[0].cpu().numpy(),
    }
    act = batch.act.cpu().numpy()
    lp = batch.old_logp.cpu().numpy()
    for k, name in enumerate(L.OUTPUT_KEYS):
        o = L.HEAD_OFFSETS[name]
        sel = act[:, o:o + L.HEAD_COUNTS[name]].any(axis=1)
        out['old_logp_' + name] = lp[sel, k]
    names = list(L.param_shapes(cell, hidden, layers).keys())
    n_ep = int(g['epochs']) if epochs is None else epochs
    for ep in range(n_ep):
        res, status = eng.train_epoch(chunks, float(g['lr']), float(g['entropy_coef']), float(g['vf_coef']))
        r = res.cpu().numpy().astype(np.float64)
        assert int(status.item()) == 0
        out['ep%d_losses' % ep] = r[0:4]
        out['ep%d_entropies' % ep] = r[4:9]
        out['ep%d_grad_norms' % ep] = r[9:11]
        gs, gv, ps, pv = [], [], [], []
        for n in names:
            s, v = util.tensor_summary(eng.param_view(n, eng.grads)); gs.append(s); gv.append(v)
            s, v = util.tensor_summary(eng.param_view(n)); ps.append(s); pv.append(v)
        out['ep%d_grad_summary' % ep] = np.stack(gs)
        out['ep%d_grad_samples' % ep] = np.concatenate(gv)
        out['ep%d_param_summary' % ep] = np.stack(ps)
        out['ep%d_param_samples' % ep] = np.concatenate(pv)
        out['ep%d_steps' % ep] = eng.seg_step.cpu().numpy().copy()
    out['param_names'] = np.array(names)
    return out, eng


def _loss_err(a, b, key):
    """Relative error per component; the four loss numbers by util.loss_rel_err (additive parts that cancel)."""
    if key == 'losses':
        return util.loss_rel_err(a, b)
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-30)))


ELEM_TOL = 2e-3     # element-wise relative error on entries above 1e-3 * max|ref| (util.elementwise_rel_err): recorded for every
                    # compare() into gpurun_out/elementwise_parity.json (tests/conftest.py) and bounded here; the 1e-4 bar is the scaled
                    # one.  Measured worst over the suite: 1.3e-3 (values), 3.9e-5 (advantages).  Where it comes from: an entry of
                    # 1e-3 * max that is a sum of O(max) f32 terms - not one layer's summation order: the fp32 ORACLE itself sits 4.3e-4
                    # (values) / 2.9e-4 (pre-rnn activations, K = 896) element-wise from its own fp64 evaluation on a quarter of the bench
                    # batch while its scaled error is 7.7e-7 (tools/values_error_budget.py, profiles/r05/values_error_budget_*.json).


def compare(out, ref, n_ep, names_ref, tol=TOL):
    from tests import conftest
    rec = {}
    for key in ['advantages', 'returns', 'values'] + ['old_logp_' + k for k in L.OUTPUT_KEYS]:
        assert out[key].shape == ref[key].shape, key
        assert util.scaled_err(out[key], ref[key]) < tol, (key, util.scaled_err(out[key], ref[key]))
        ew, frac = util.elementwise_rel_err(out[key], ref[key])
        rec[key] = {'scaled': util.scaled_err(out[key], ref[key]), 'elementwise': ew, 'entries_above_floor': frac}
        assert ew < ELEM_TOL, (key, ew)
    conftest.record_elementwise(rec)
    # action argmax indices: bit-exact
    assert np.array_equal(out['argmax'], ref['argmax'].astype(out['argmax'].dtype))
    if 'hidden' in ref and 'hidden' in out:
        assert util.scaled_err(out['hidden'], ref['hidden']) < tol
    assert list(out['param_names']) == list(names_ref)
    for ep in range(n_ep):
        for key in ['losses', 'entropies', 'grad_norms']:
            k = 'ep%d_%s' % (ep, key)
            assert _loss_err(out[k], ref[k], key) < tol, (k, out[k], ref[k])
        # clipped gradients: per-tensor L2 norm and strided samples
        gn, rn = out['ep%d_grad_summary' % ep][:, 2], ref['ep%d_grad_summary' % ep][:, 2]
        assert util.scaled_err(gn, rn) < 5 * tol, (ep, np.abs(gn - rn).max(), rn.max())
        assert util.scaled_err(out['ep%d_grad_samples' % ep], ref['ep%d_grad_samples' % ep]) < 5 * tol
        # post-step parameters
        assert util.scaled_err(out['ep%d_param_samples' % ep], ref['ep%d_param_samples' % ep]) < tol
        pn, rpn = out['ep%d_param_summary' % ep][:, 2], ref['ep%d_param_summary' % ep][:, 2]
        assert util.rel_err(pn, rpn) < tol


@pytest.mark.parametrize('case', util.CASES + util.BIG_CASES)
def test_hip_matches_reference_golden(case):
    # incl. the entropy_coef == 0 / vf_coef == 0 branches (optimizer.py:652-663: affine_value then has no gradient and is
    # not stepped) and cfg2_gru_64x256 = the REAL reference on BASELINE.json configs[1]'s batch (64 trajectories x 256
    # steps, GRU-256): default kernel selection - team kernels, fused embedding forward, sparse max-pool backward
    g, rollouts = util.load_case(case)
    out, eng = run_hip(g, rollouts)
    compare(out, g, int(g['epochs']), g['param_names'])
    # parameters without a gradient in the reference (head never acted) must not have been stepped
    for ep in range(int(g['epochs'])):
        assert np.array_equal(out['ep%d_steps' % ep] > 0, g['ep%d_has_grad' % ep])


@pytest.mark.parametrize('cell,hidden,layers', [('lstm', 128, 1), ('lstm', 256, 1), ('gru', 128, 1), ('lstm', 128, 2), ('lstm', 256, 2),
                                                ('lstm', 512, 2)])
def test_hip_matches_oracle_other_cells(cell, hidden, layers):
    # no reference implementation exists for these: the oracle restatement is the bar
    g, rollouts = util.load_case('ragged_s16')
    ref, _, _ = util.oracle_run(g, rollouts, cell, hidden, layers, epochs=2)
    out, _ = run_hip(g, rollouts, cell, hidden, layers, epochs=2)
    ref.pop('hidden', None); out.pop('hidden', None)
    compare(out, ref, 2, ref['param_names'])


_ORACLE_CACHE = {}


def _oracle_cached(key, g, rollouts, cell, hidden, layers, epochs):
    if key not in _ORACLE_CACHE:
        torch.set_num_threads(min(16, torch.get_num_threads()))
        ref, _, _ = util.oracle_run(g, rollouts, cell, hidden, layers, epochs=epochs)
        _ORACLE_CACHE[key] = ref
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize('reuse_forward', [False, True])
@pytest.mark.parametrize('cell,hidden,B', [('lstm', 128, 64), ('lstm', 256, 256), ('lstm', 256, 128)])
def test_hip_matches_oracle_at_baseline_configs(cell, hidden, B, reuse_forward):
    # BASELINE.json configs[1] (LSTM-128, 64 x 256), configs[2] (LSTM-256, 256 x 256) and configs[3]'s per-GPU shard (LSTM-256,
    # 128 x 256: the N > 1 bench line's workload, on the boundary of the recurrent-kernel selection) - the very batches bench.py
    # times (same seed) - against the oracle run live (about 4 / 25 / 12 s of CPU), one epoch, with the DEFAULT kernel
    # selection: persistent VALU LSTM / team kernels with four sequences in flight, fused embedding forward, sparse
    # max-pool backward over many tiles.  Advantages, returns, values, old log-probs, losses, entropies, gradient norms,
    # clipped gradients, post-step parameters < 1e-4; masked argmax bit-exact.  reuse_forward: the same with the first epoch
    # back-propagating the rollout pass's activations (Engine.reuse_rollout_forward) - same bar.
    S = 256
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(1000, [S] * B)
    ref = _oracle_cached((cell, hidden, B), g, rollouts, cell, hidden, 1, 1)
    out, eng = run_hip(g, rollouts, cell, hidden, 1, epochs=1, reuse_forward=reuse_forward)
    out.pop('hidden', None)
    compare(out, ref, 1, ref['param_names'])
    assert eng.fault() is None


def reference_default_lengths():
    """optimizer.py:776-794 defaults: seq_len 16, whole rollouts until >= 1024 chunks (bench.py's `reference_defaults_gru256_s16_ragged`)."""
    rng = np.random.Generator(np.random.PCG64(99))
    lens, chunks = [], 0
    while chunks < 1024:
        t = int(rng.integers(100, 900))
        lens.append(t)
        chunks += (t + 15) // 16
    return lens


@pytest.mark.parametrize('reuse_forward', [False, True])
def test_hip_matches_oracle_at_reference_default_shape(reuse_forward):
    # the reference's own production shape: its GRU-256, seq_len 16, 31 ragged rollouts of 100..899 steps = 1 065 chunks (17 040
    # rows: a multiple of 16, not of 128 - padded embedding blocks; five rounds of the MFMA team kernels) against the oracle, 2 epochs
    S = 16
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 2}
    lens = reference_default_lengths()
    assert sum((t + 15) // 16 for t in lens) >= 1024
    rollouts = synth.make_rollouts(1000, lens)
    ref = _oracle_cached(('gru-defaults',), g, rollouts, 'gru', 256, 1, 2)
    out, eng = run_hip(g, rollouts, 'gru', 256, 1, epochs=2, reuse_forward=reuse_forward)
    compare(out, ref, 2, ref['param_names'])
    assert eng.fault() is None


@pytest.mark.parametrize('case', ['ragged_s16', 'clip_s16', 'emptyhead_s16'])
def test_unfused_embedding_path_matches_reference_golden(case):
    # default: the fused embedding kernels on type-major blocks padded to a multiple of 128 rows (176 / 176 / 80 rows here) with
    # the sparse max-pool backward; DC_DIMS_EMBED_UNFUSED: layer by layer with the dense products - the same golden vectors
    from dotaclient_amd import engine as E
    g, rollouts = util.load_case(case)
    out, _ = run_hip(g, rollouts, kernel_flags=E.DC_DIMS_EMBED_UNFUSED)
    compare(out, g, int(g['epochs']), g['param_names'])
    out, _ = run_hip(g, rollouts, kernel_flags=E.DC_DIMS_DENSE_POOL_BWD)      # fused + padded, dense max-pool backward
    compare(out, g, int(g['epochs']), g['param_names'])


@pytest.mark.parametrize('which', ['x3_all', 'fasttile'])
@pytest.mark.parametrize('case,cell,hidden,layers', [('ragged_s16', 'gru', 256, 1), ('cfg1_4x128', 'gru', 256, 1), ('ragged_s16', 'lstm', 128, 2)])
def test_dense_product_kernels_both_match(which, case, cell, hidden, layers):
    # the default mixes the two dense-product kernels (tile kernel for x W^T / dy W, split-on-load kernel for the weight
    # gradients); each is also run on EVERY dense product of the step against the reference golden / the oracle
    from dotaclient_amd import engine as E
    g, rollouts = util.load_case(case)
    flags = E.DC_DIMS_GEMM_X3_ALL if which == 'x3_all' else E.DC_DIMS_GEMM_FASTTILE
    n_ep = int(g['epochs'])
    out, _ = run_hip(g, rollouts, cell, hidden, layers, kernel_flags=flags)
    if (cell, hidden, layers) == ('gru', 256, 1):
        compare(out, g, n_ep, g['param_names'])
    else:
        ref, _, _ = util.oracle_run(g, rollouts, cell, hidden, layers)
        out.pop('hidden', None)
        compare(out, ref, n_ep, ref['param_names'])


@pytest.mark.parametrize('cell,B', [('lstm', 160), ('gru', 160), ('lstm', 24)])
def test_forward_only_rollout_pass_is_bit_identical(cell, B):
    # DC_DIMS_FWD_ONLY (round 6): the no-grad rollout pass (optimizer.py:344-385) tells the library that no backward will read it; the H = 256
    # MFMA team forward (more than 128 sequences) then skips the stores only a backward needs (activated gates, previous h / c).  Everything
    # the pass is FOR - old log-probs, values, masked argmax, advantages, the chunks' initial states - must not change by a bit; 24
    # sequences: kernels that ignore the flag.
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    S = 64
    rollouts = synth.make_rollouts(77, [S * 2] * B)            # two chunks per rollout: the carried state matters
    outs = []
    for reuse in (False, True):                                 # reuse_rollout_forward = True keeps the full forward (its first epoch reads it)
        eng = Engine(cell, 256, 1, dev)
        eng.reuse_rollout_forward = reuse
        eng.load_state_dict(synth.init_state_dict(7, cell, 256, 1))
        batch = pack_rollouts(rollouts, S, dev)
        chunks = eng.rollout_pass(batch, S)
        outs.append([batch.old_logp.clone(), batch.values.clone(), batch.argmax.clone(), batch.adv.clone(), chunks.h0.clone()] +
                    ([chunks.c0.clone()] if cell == 'lstm' else []))
        assert eng.fault() is None
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('cell,hidden,B', [('lstm', 256, 128), ('gru', 256, 64)])
def test_row_streaming_products_agree_with_the_tile_kernel(cell, hidden, B):
    # round 6: at bench-sized batches the x W^T / dy W products of the default arithmetic run on the row-streaming kernel
    # (csrc/gemm_x3s.hip: LDS-DMA for both operands, two wave groups half a stage apart); DC_DIMS_GEMM_TILE128 keeps them on round 4's
    # 128 x 128 split-on-load kernel.  Same pieces, same three MFMAs per product, another summation order: one epoch of configs[3]'s shard
    # (LSTM-256, 128 x 256: pre-rnn, gates, heads and their input gradients all eligible) and of the reference's GRU at 64 x 256 (gates and
    # d(xcat) eligible, the 256-column products stay on the tile kernel) must agree far inside the parity bar
    from dotaclient_amd import engine as E
    S = 256
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(1000, [S] * B)
    a, _ = run_hip(g, rollouts, cell, hidden, 1, epochs=1)
    b, eng = run_hip(g, rollouts, cell, hidden, 1, epochs=1, kernel_flags=E.DC_DIMS_GEMM_TILE128)
    for key in ['values', 'advantages'] + ['old_logp_' + k for k in L.OUTPUT_KEYS]:
        assert util.scaled_err(a[key], b[key]) < 5e-6, (key, util.scaled_err(a[key], b[key]))
    assert np.array_equal(a['argmax'], b['argmax'])
    for key in ['losses', 'entropies', 'grad_norms']:
        assert _loss_err(a['ep0_' + key], b['ep0_' + key], key) < 1e-5, key
    assert util.scaled_err(a['ep0_grad_samples'], b['ep0_grad_samples']) < 2e-5
    assert eng.fault() is None


@pytest.mark.parametrize('variant', ['mfma', 'valu'])
@pytest.mark.parametrize('hidden,layers', [(128, 1), (64, 2)])
def test_lstm_persist_variants_match_oracle(variant, hidden, layers):
    # both register-resident LSTM kernels (4 sequences/workgroup on the MFMA, 1 sequence/workgroup on the
    # packed-f32 VALU) against the oracle, on ragged rollouts (T = 50, 64, 33; chunks of 16): the size rule
    # in lstm_persist_use_valu would otherwise leave one of them untested
    from dotaclient_amd import engine as E
    g, rollouts = util.load_case('ragged_s16')
    ref, _, _ = util.oracle_run(g, rollouts, 'lstm', hidden, layers, epochs=2)
    out, _ = run_hip(g, rollouts, 'lstm', hidden, layers, epochs=2,
                     kernel_flags=E.DC_DIMS_LSTM_MFMA if variant == 'mfma' else E.DC_DIMS_LSTM_VALU)
    ref.pop('hidden', None); out.pop('hidden', None)
    compare(out, ref, 2, ref['param_names'])


@pytest.mark.parametrize('S,lens', [(256, [256] * 6), (7, [21, 7, 13, 30, 1, 44]), (5, [5, 9, 2])])
def test_lstm_persist_variants_agree(S, lens):
    # 256-step trajectories (the bench shape: whole groups of 4 steps) and chunk lengths 7 / 5 with padded
    # rollouts of 7..49 steps (every remainder of the 4-step groups): the two variants differ only in
    # summation order
    from dotaclient_amd import engine as E
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    outs = {}
    for variant in ('mfma', 'valu'):
        eng = Engine('lstm', 128, 1, dev)
        eng.kernel_flags = E.DC_DIMS_LSTM_MFMA if variant == 'mfma' else E.DC_DIMS_LSTM_VALU
        eng.load_state_dict(synth.init_state_dict(7, 'lstm', 128, 1))
        rollouts = synth.make_rollouts(77, lens)
        batch = pack_rollouts(rollouts, S, dev)
        chunks = eng.rollout_pass(batch, S)
        res, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
        assert int(status.item()) == 0
        outs[variant] = (batch.values.cpu().numpy().copy(), batch.adv.cpu().numpy().copy(), res.cpu().numpy().copy(),
                         eng.grads.cpu().numpy().copy())
    for a, b in zip(outs['mfma'], outs['valu']):
        assert util.scaled_err(a, b) < 2e-5, util.scaled_err(a, b)


@pytest.mark.parametrize('cell', ['gru', 'lstm'])
@pytest.mark.parametrize('S,lens', [(256, [256] * 6), (7, [21, 7, 13, 30, 1, 44]), (5, [350]), (16, [50, 64, 33]), (3, [1200, 2]), (8, [8] * 131 + [24, 16]),
                                    (2, [1100]), (4, [3600, 30]), (16, [16] * 70 + [48, 160, 33, 200])])
def test_rnn_team_kernels_agree_with_per_step(cell, S, lens):
    # H = 256 (the reference's GRU, the LSTM-256 configs): the four-workgroups-per-sequence persistent kernels
    # (rnn_team.hip) against the launch-per-step kernels on the same batch.  6 / 17 / 70 / 11 / 401 chunk sequences:
    # fewer teams than 8 (plain block -> team map), 16 teams with one sequence left over, more sequences than the 64
    # teams (two streams per team, then four with several sequences per stream: tag / ring continuity across sequence
    # boundaries, streams that retire early), every remainder of the step groups; 908 chunks: beyond the VALU team kernels' limit of
    # 768 sequences (the LSTM's MFMA team kernels have none, the others fall back to the per-step kernels).  Each case also with the stream
    # count forced to 1, 2 and 4.  All differ only in summation order.
    from dotaclient_amd import engine as E
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    outs = {}
    # mode '1' / None = the default selection (LSTM with more than 128 sequences: the MFMA team kernel, four sequences per
    # team step); 'v' = the VALU team kernels whatever the batch
    # '8' = teams of eight workgroups (DC_DIMS_TEAM8, rnn_team8.hip: the default's choice for 65 .. 128 sequences only), '4' = never those
    # 'k' = the MFMA team forward with the k halves (round 2's; DC_DIMS_TEAM_NS(2) without the VALU flag) where the MFMA kernels run
    # 'd' = the default selection with DC_DIMS_TEAM_DEVICE_SCOPE: the hand-off as write-through stores even when a team sits on one XCD
    for mode, ns in (('0', None), ('1', None), ('d', None), ('8', None), ('4', None), ('k', '2'), ('v', None), ('v', '1'), ('v', '2'), ('v', '4')):
        eng = Engine(cell, 256, 1, dev)
        eng.kernel_flags = E.DC_DIMS_RNN_PER_STEP if mode == '0' else \
            ((E.DC_DIMS_TEAM_VALU if mode == 'v' else 0) | (E.DC_DIMS_TEAM8 if mode == '8' else 0) | (E.DC_DIMS_TEAM4 if mode == '4' else 0) |
             (E.DC_DIMS_TEAM_DEVICE_SCOPE if mode == 'd' else 0) |
             (E.DC_DIMS_TEAM_NS(int(ns)) if ns else 0))
        eng.load_state_dict(synth.init_state_dict(7, cell, 256, 1))
        rollouts = synth.make_rollouts(78, lens)
        batch = pack_rollouts(rollouts, S, dev)
        chunks = eng.rollout_pass(batch, S)
        res, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
        assert int(status.item()) == 0 and eng.fault() is None, (mode, ns, E.describe_status(eng))
        outs[(mode, ns)] = (batch.values.cpu().numpy().copy(), batch.adv.cpu().numpy().copy(), res.cpu().numpy().copy(),
                            eng.grads.cpu().numpy().copy())
    for key in outs:
        for a, b in zip(outs[('0', None)], outs[key]):
            assert util.scaled_err(a, b) < 2e-5, (key, util.scaled_err(a, b))


@pytest.mark.parametrize('lens', [[128] * 4, [256] * 6, [384, 128, 256], [128] * 3, [100, 150, 77], [33]])
def test_sparse_pool_backward_matches_dense(lens):
    # fused embedding path (rows % 128 == 0): the max-pool backward of the two 16-unit types - default (f16x2 products): dense products on
    # the f16 matrix cores with on-chip operands (embed_pool16m.hip); DC_DIMS_POOL16_VALU / _8W: the sparse VALU kernels (embed_sparse.hip) -
    # against the dense MFMA kernels that read d(emb) from HBM, on the same batch: gradients and post-step parameters.  [128] * 3: 384 steps
    # over 128 workgroups = an odd number of steps per workgroup (the matrix-core kernel works on PAIRS of steps)
    # [100, 150, 77] / [33]: 327 / 33 steps that exist inside 384 / 128 padded ones - the small-type kernel's last tile is partial (33 steps of the 5-unit
    # type: 165 rows = two tiles and 37 rows) and most of its workgroups have no rows at all
    from dotaclient_amd import engine as E
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    outs = {}
    # (round 6: by default the four SMALL types are on chip too - embed_small.hip, no d(emb) for any type; 'sd' = DC_DIMS_SMALL_DENSE keeps
    # them on d(emb) in HBM + the dense kernels next to the on-chip 16-unit kernels)
    # 'db2s' = DC_DIMS_DB2_SCATTER: the small types' second-layer bias gradients from embed_scatter_bwd's own pass instead of embed_small.hip's column sums
    for mode in ('0', '1', 'sd', 'db2s', '16w', '8w'):   # dense kernels / on-chip (default) / on-chip 16-unit types only / .. / sparse sixteen-wave kernel / sparse eight-wave kernel
        eng = Engine('lstm', 128, 1, dev)
        eng.kernel_flags = {'0': E.DC_DIMS_DENSE_POOL_BWD, '1': 0, 'sd': E.DC_DIMS_SMALL_DENSE, 'db2s': E.DC_DIMS_DB2_SCATTER, '16w': E.DC_DIMS_POOL16_VALU,
                            '8w': E.DC_DIMS_POOL16_8W}[mode]
        eng.load_state_dict(synth.init_state_dict(7, 'lstm', 128, 1))
        rollouts = synth.make_rollouts(91, lens)
        batch = pack_rollouts(rollouts, 128, dev)
        chunks = eng.rollout_pass(batch, 128)
        res, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
        assert int(status.item()) == 0
        g = {n: eng.param_view(n, eng.grads).cpu().numpy().copy() for n in
             ('affine_unit_basic_stats.weight', 'affine_unit_basic_stats.bias', 'affine_unit_anh.weight',
              'affine_unit_enh.weight', 'affine_unit_anh.bias', 'affine_unit_enh.bias', 'affine_unit_eh.weight',
              'affine_unit_ah.weight', 'affine_unit_ath.weight', 'affine_unit_eth.weight', 'affine_unit_eh.bias',
              'affine_unit_ah.bias', 'affine_unit_ath.bias', 'affine_unit_eth.bias', 'affine_env.weight', 'affine_env.bias')}
        outs[mode] = (g, res.cpu().numpy().copy(), eng.params.cpu().numpy().copy())
    # All variants evaluate the relu mask of the first layer with exact-f32 MFMAs (a first version of the on-chip kernel took the forward's
    # two-f16-piece sequence: one pre-activation in ~10^6 got the other sign, a whole term of ONE hidden unit's row of dW1 / db1 - 3.6e-3 of
    # the largest entry at 1 536 steps).  Two exact-f32 evaluations in different summation orders can still disagree on the sign of a
    # pre-activation that is zero to f32 round-off - and only there.  So: every tensor at 2e-5; a hidden unit's row of
    # dW1 / db1 may exceed that ONLY IF that unit has a pre-activation within f32 round-off of zero somewhere in the batch (float64
    # evaluation of z = W1 x + b1 over all unit records: |z| < 4e-6, i.e. ~30 ulp of the O(1) terms that sum to it), and then stays below 1e-2.
    sd = synth.init_state_dict(7, 'lstm', 128, 1)
    units = batch.obs.cpu().double()[:, 3:].reshape(-1, 12)
    z = units @ sd['affine_unit_basic_stats.weight'].double().t() + sd['affine_unit_basic_stats.bias'].double()
    may_flip = (z.abs().min(dim=0).values < 4e-6).numpy()          # [128]: hidden units with a pre-activation on the relu's kink
    for n in outs['0'][0]:
        a, b = outs['1'][0][n], outs['0'][0][n]
        if n.startswith('affine_unit_basic_stats'):
            row_err = np.abs(a - b).reshape(128, -1).max(axis=1) / np.abs(b).max()
            off = row_err >= 2e-5
            assert not (off & ~may_flip).any(), (n, 'rows off without a pre-activation on the kink', np.nonzero(off & ~may_flip)[0], np.sort(row_err)[-4:])
            assert row_err.max() < 1e-2, (n, np.sort(row_err)[-4:])
        else:
            assert util.scaled_err(a, b) < 2e-5, (n, util.scaled_err(a, b))
    assert util.scaled_err(outs['1'][1][:11], outs['0'][1][:11]) < 2e-5
    assert util.scaled_err(outs['1'][2], outs['0'][2]) < 5e-5
    for n in outs['0'][0]:              # the 16-unit types on chip, the small ones dense: between the two
        if not n.startswith('affine_unit_basic_stats'):
            assert util.scaled_err(outs['sd'][0][n], outs['0'][0][n]) < 2e-5, n
    assert not np.array_equal(outs['sd'][0]['affine_unit_eh.weight'], outs['1'][0]['affine_unit_eh.weight'])     # really another kernel
    for n in outs['0'][0]:              # the small types' bias gradients from the scatter pass: everything else is the default's bit for bit
        if n in ('affine_unit_eh.bias', 'affine_unit_ah.bias', 'affine_unit_ath.bias', 'affine_unit_eth.bias'):
            # (measured 1e-6 .. 2e-6: the on-chip sums add two f16 pieces per element, and the partials meet in atomics whose order varies)
            assert util.scaled_err(outs['db2s'][0][n], outs['1'][0][n]) < 1e-5, n
            assert util.scaled_err(outs['db2s'][0][n], outs['0'][0][n]) < 2e-5, n
        elif not n.startswith('affine_env'):          # (atomics in the reductions of the env gradient: order varies)
            assert np.array_equal(outs['db2s'][0][n], outs['1'][0][n]) or util.scaled_err(outs['db2s'][0][n], outs['1'][0][n]) < 2e-6, n
        else:
            assert util.scaled_err(outs['db2s'][0][n], outs['1'][0][n]) < 2e-6, n
    assert not np.array_equal(outs['db2s'][0]['affine_unit_eh.bias'], outs['1'][0]['affine_unit_eh.bias'])        # really another summation
    for n in outs['0'][0]:              # the sparse VALU kernels: the same sums in another order
        assert util.scaled_err(outs['8w'][0][n], outs['0'][0][n]) < 2e-5, n
        assert util.scaled_err(outs['16w'][0][n], outs['0'][0][n]) < 2e-5, n
    assert util.scaled_err(outs['8w'][2], outs['0'][2]) < 2e-5 and util.scaled_err(outs['16w'][2], outs['0'][2]) < 2e-5
    assert not np.array_equal(outs['16w'][0]['affine_unit_anh.weight'], outs['1'][0]['affine_unit_anh.weight'])   # really another kernel


@pytest.mark.parametrize('cell,hidden,B', [('lstm', 128, 96), ('gru', 256, 96), ('lstm', 256, 200)])
def test_full_size_batch_is_invariant_to_trajectory_order(cell, hidden, B):
    # BASELINE.json-sized batches (96 / 200 trajectories x 256 steps, fused embedding path; persistent LSTM-128, the
    # reference's GRU-256 and LSTM-256 on the team kernels with two and four sequences in flight per team) - too large
    # for the oracle in a test, so a size-independent property instead: the optimizer step is a sum over trajectories,
    # hence permuting them must leave the losses, the gradient norms and the post-step parameters unchanged (up to fp32
    # summation order) and must permute the per-step values / advantages with them
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    S = 256
    rollouts = synth.make_rollouts(2024, [S] * B)
    perm = np.random.Generator(np.random.PCG64(5)).permutation(B)
    outs = []
    for order in (np.arange(B), perm):
        eng = Engine(cell, hidden, 1, dev)
        eng.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
        batch = pack_rollouts([rollouts[i] for i in order], S, dev)
        chunks = eng.rollout_pass(batch, S)
        res, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
        assert int(status.item()) == 0
        r = res.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(r[:11]))
        outs.append((r[:11], batch.values.view(B, S).cpu().numpy(), batch.adv.view(B, S).cpu().numpy(),
                     eng.params.cpu().numpy().copy()))
    (r0, v0, a0, p0), (r1, v1, a1, p1) = outs
    assert util.rel_err(r1, r0) < 2e-5, (r0, r1)
    assert np.array_equal(v1, v0[perm]) and np.array_equal(a1, a0[perm])      # per-trajectory work is order-independent
    assert util.scaled_err(p1, p0) < 1e-6


@pytest.mark.parametrize('cell,hidden', [('lstm', 128), ('gru', 256)])
def test_backward_in_two_calls_equals_one_call(cell, hidden):
    # DC_DIMS_BWD_UPPER then DC_DIMS_BWD_EMBED (the form a data-parallel caller uses to overlap the all-reduce of the
    # first part's gradients with the second part) against the single call: same gradients, same step
    from dotaclient_amd.engine import Engine, pack_rollouts

    class Hook:                      # what FlatGradAllReducer looks like to train_epoch, without a process group
        overlap = True
        calls = []
        def start_upper(self, eng): self.calls.append(('upper', eng.grads[eng.embed_floats:].abs().sum().item(), eng.grads[:eng.embed_floats].abs().sum().item()))
        def finish(self, eng): self.calls.append(('finish', eng.grads[:eng.embed_floats].abs().sum().item()))

    dev = torch.device('cuda:0')
    outs = []
    for hook in (None, Hook()):
        eng = Engine(cell, hidden, 1, dev)
        eng.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
        rollouts = synth.make_rollouts(31, [128, 256, 128])
        batch = pack_rollouts(rollouts, 128, dev)
        chunks = eng.rollout_pass(batch, 128)
        res, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5, grad_hook=hook)
        assert int(status.item()) == 0
        outs.append((eng.grads.cpu().numpy().copy(), eng.params.cpu().numpy().copy(), res.cpu().numpy().copy()))
    (g0, p0, r0), (g1, p1, r1) = outs
    assert util.scaled_err(g1, g0) < 1e-6 and util.scaled_err(p1, p0) < 1e-7 and util.scaled_err(r1[:11], r0[:11]) < 1e-6
    (k0, up, emb0), (k1, emb1) = Hook.calls[-2:]
    assert k0 == 'upper' and up > 0 and emb0 == 0.0          # after the first call: upper gradients final, embedding ones still zero
    assert k1 == 'finish' and emb1 > 0


@pytest.mark.parametrize('cell,hidden', [('lstm', 128), ('gru', 256)])
def test_epoch_graph_replay_equals_eager(cell, hidden):
    # Engine.train_epoch(graph=True): first epoch eager, second captured into a hipGraph and replayed, third a pure replay -
    # against three eager epochs from the same start (same kernels, same order: differences only from the f64 atomics of
    # the norm reduction)
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    rollouts = synth.make_rollouts(55, [128, 256, 128, 64])
    outs = []
    for graph in (False, True):
        eng = Engine(cell, hidden, 1, dev)
        eng.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
        batch = pack_rollouts(rollouts, 64, dev)
        chunks = eng.rollout_pass(batch, 64)
        per = []
        for ep in range(3):
            res, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5, graph=graph)
            assert int(status.item()) == 0
            per.append(res.cpu().numpy()[:11].copy())
        if graph:
            assert len(eng._graphs) == 1 and next(iter(eng._graphs.values()))['graph'] is not None
        outs.append((np.stack(per), eng.params.cpu().numpy().copy(), eng.seg_step.cpu().numpy().copy()))
    (r0, p0, s0), (r1, p1, s1) = outs
    # (parameters: Adam turns a last-bit difference of a near-zero gradient element into a step of order lr)
    assert util.scaled_err(r1, r0) < 1e-6 and util.scaled_err(p1, p0) < 1e-3 and np.array_equal(s0, s1)


def test_two_engines_on_two_streams_from_two_threads():
    # include/dotaclient_hip.h: no state between calls, all scratch from the caller - so two engines with their own
    # buffers can be driven from two host threads on two streams at once.  GRU-256 on 40 sequences each: two team-kernel
    # grids of 160 workgroups compete for the 256 CUs, so neither is fully resident while the other runs - the ticketed
    # roles of rnn_team.hip must still complete (no spin time-out, no NaN) and give each engine its sequential result.
    import threading
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    S = 64
    data = [synth.make_rollouts(61 + i, [S] * 40) for i in range(2)]

    def run(i, stream, out):
        with torch.cuda.stream(stream):
            eng = Engine('gru', 256, 1, dev)
            eng.load_state_dict(synth.init_state_dict(7 + i, 'gru', 256, 1))
            batch = pack_rollouts(data[i], S, dev)
            res = []
            for it in range(3):
                chunks = eng.rollout_pass(batch, S)
                for ep in range(2):
                    r, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
                    res.append(r.clone())
            stream.synchronize()
            assert int(eng.status.item()) == 0
            out[i] = (torch.stack(res).cpu().numpy()[:, :11], eng.params.cpu().numpy().copy())

    seq, par = {}, {}
    for i in range(2):                                     # one after the other on the default stream
        run(i, torch.cuda.current_stream(), seq)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    threads = [threading.Thread(target=run, args=(i, streams[i], par)) for i in range(2)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert set(par) == {0, 1}
    for i in range(2):
        assert np.all(np.isfinite(par[i][0]))
        assert util.scaled_err(par[i][0], seq[i][0]) < 1e-5, i
        assert util.scaled_err(par[i][1], seq[i][1]) < 1e-3, i        # Adam turns last-bit gradient differences (f64 atomics in the norm) into steps of order lr


@pytest.mark.parametrize('cell,hidden,lens,S', [('lstm', 256, [64] * 8, 64), ('lstm', 128, [50, 64, 33, 16], 16), ('gru', 256, [40, 33], 16)])
def test_first_epoch_can_reuse_the_rollout_pass_forward(cell, hidden, lens, S):
    # Engine.reuse_rollout_forward: epoch 0 runs on the weights the rollout pass has just used, so its forward is the same
    # function of the same inputs (optimizer.py:328-430 then :581-689); skipping it must not change any result beyond the
    # kernels' summation orders (rollout-shaped vs chunk-shaped launches).  Later epochs (new weights) must recompute.
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    outs = {}
    for reuse in (False, True):
        eng = Engine(cell, hidden, 1, dev)
        eng.reuse_rollout_forward = reuse
        eng.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
        batch = pack_rollouts(synth.make_rollouts(321, lens), S, dev)
        chunks = eng.rollout_pass(batch, S)
        res = []
        for _ in range(3):
            out, status = eng.train_epoch(chunks, 3e-4, 5e-4, 0.5)
            assert int(status.item()) == 0
            res.append(out.cpu().numpy().copy())
        outs[reuse] = (np.stack(res), eng.params.cpu().numpy().copy(), eng.grads.cpu().numpy().copy())
    for a, b in zip(outs[True], outs[False]):
        assert util.scaled_err(a, b) < 5e-5, util.scaled_err(a, b)      # (three Adam steps at lr 3e-4 amplify summation-order differences of the
        # rollout-shaped vs chunk-shaped forward: 1e-5 .. 3e-5 depending on the backward kernels' own summation order; the parity bar is 1e-4)


def test_sparse_pool_backward_with_one_unit_taking_every_channel():
    # Degenerate arg-max patterns for embed_sparse.hip: steps whose sixteen non-hero units are IDENTICAL (ties: the first unit
    # wins all 128 channels, torch.max's rule - the kernel's per-unit channel list is 128 long, far beyond its twelve
    # straight-line slots), steps where two units share them, and ordinary steps, mixed in one batch; against the dense kernels.
    from dotaclient_amd import engine as E
    from dotaclient_amd.engine import Engine, pack_rollouts
    from dotaclient_amd import layout as L
    dev = torch.device('cuda:0')
    rollouts = synth.make_rollouts(55, [128, 128, 128])
    for ri, data in enumerate(rollouts):
        for key in ('allied_nonheroes', 'enemy_nonheroes'):
            if key not in data['observations']:
                continue
            x = np.asarray(data['observations'][key]).copy()          # [T, 16, 12]
            x[ri::4] = x[ri::4, :1]                                    # every fourth step: sixteen copies of unit 0
            x[(ri + 2) % 4::8, 1::2] = x[(ri + 2) % 4::8, 1:2]        # and some steps with eight copies of unit 1 among the others
            data['observations'][key] = x
    outs = {}
    for mode in ('dense', 'sparse', 'valu'):
        eng = Engine('lstm', 128, 1, dev)
        eng.kernel_flags = {'dense': E.DC_DIMS_DENSE_POOL_BWD, 'sparse': 0, 'valu': E.DC_DIMS_POOL16_VALU}[mode]
        eng.load_state_dict(synth.init_state_dict(7, 'lstm', 128, 1))
        batch = pack_rollouts(rollouts, 128, dev)
        chunks = eng.rollout_pass(batch, 128)
        res, status = eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
        assert int(status.item()) == 0
        outs[mode] = (eng.grads.cpu().numpy().copy(), res.cpu().numpy().copy())
    for mode in ('sparse', 'valu'):
        assert util.scaled_err(outs[mode][0], outs['dense'][0]) < 2e-5, mode
        assert util.scaled_err(outs[mode][1][:11], outs['dense'][1][:11]) < 2e-5, mode


# ---- the two forms of f32-grade products: 'f16x2' (Engine default: two f16 pieces, three MFMAs, DC_DIMS_F16X2) is what every test above
# ran; 'bf16x3' (three bf16 pieces, six MFMAs: f32's exponent range, the fallback) must meet the same bars ----------------------------------
@pytest.mark.parametrize('case', util.CASES + util.BIG_CASES)
def test_bf16x3_products_match_reference_golden(case):
    from dotaclient_amd.engine import Engine
    assert Engine('gru', 256, 1, torch.device('cuda:0')).products == 'f16x2'       # the default the other tests exercise
    g, rollouts = util.load_case(case)
    out, eng = run_hip(g, rollouts, products='bf16x3')
    compare(out, g, int(g['epochs']), g['param_names'])
    for ep in range(int(g['epochs'])):
        assert np.array_equal(out['ep%d_steps' % ep] > 0, g['ep%d_has_grad' % ep])


@pytest.mark.parametrize('cell,hidden,B', [('lstm', 256, 256), ('lstm', 128, 64)])
def test_bf16x3_products_match_oracle_at_baseline_configs(cell, hidden, B):
    S = 256
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(1000, [S] * B)
    ref = _oracle_cached((cell, hidden, B), g, rollouts, cell, hidden, 1, 1)
    out, eng = run_hip(g, rollouts, cell, hidden, 1, epochs=1, products='bf16x3')
    out.pop('hidden', None)
    compare(out, ref, 1, ref['param_names'])
    # and it really is another arithmetic than the default's
    dflt, _ = run_hip(g, rollouts, cell, hidden, 1, epochs=1)
    assert not np.array_equal(out['values'], dflt['values'])


def test_f16x2_out_of_range_operand_trips_the_nan_guard_and_the_bf16_pieces_do_not():
    # f16 has five exponent bits: an activation beyond 65504 / 2^4 becomes inf in its first piece.  That must surface as the
    # reference's own NaN guard (status word, nothing updated) - never as a silently wrong step - and the three-bf16-piece form
    # (f32's exponent range) must handle the same input; Engine.use_safe_products() is the switch the consumer loop throws.
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    rollouts = synth.make_rollouts(5, [32, 32])
    rollouts[0]['observations']['env'][3, 1] = 3.0e6          # -> env embedding ~1e5..1e6 in xcat: beyond 4094
    res = {}
    for name in ('f16x2', 'bf16x3'):
        eng = Engine('gru', 256, 1, dev)
        eng.products = name
        eng.load_state_dict(synth.init_state_dict(7))
        before = eng.params.clone()
        chunks = eng.rollout_pass(pack_rollouts(rollouts, 16, dev), 16)
        eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
        res[name] = (int(eng.status.item()), torch.equal(before, eng.params), bool(torch.isfinite(chunks.values).all()))
        if name == 'f16x2':      # the fallback: same engine, safe products, the iteration repeated - now it steps
            assert eng.use_safe_products() and eng.products == 'bf16x3' and int(eng.status.item()) == 0
            chunks = eng.rollout_pass(pack_rollouts(rollouts, 16, dev), 16)
            eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
            assert int(eng.status.item()) == 0 and not torch.equal(before, eng.params)
            assert not eng.use_safe_products()
    assert res['bf16x3'][0] == 0 and not res['bf16x3'][1] and res['bf16x3'][2]
    assert res['f16x2'][0] != 0 and res['f16x2'][1]


@pytest.mark.parametrize('key,unit', [('allied_nonheroes', 3), ('enemy_heroes', 2), ('allied_towers', 0)])
def test_f16x2_fused_embedding_first_layer_at_the_range_edge(key, unit):
    # VERDICT r4 missing 3: the fused embedding kernels regenerate their FIRST layer on the f16 matrix cores from two-piece splits of the
    # unit records (x 2^4) and of W1 (x 2^8) - policy.py:100-127.  A record entry at 0.9 x the activation limit (65504 / 16 = 4094) must
    # give the same numbers as the layer-by-layer path (DC_DIMS_EMBED_UNFUSED, dense products) and as the bf16x3 products (f32's
    # exponent range); one just over the limit must turn that env-step's outputs NON-FINITE in both f16x2 paths (-> NaN loss -> the
    # reference's NaN guard -> Engine.use_safe_products), never finite and wrong.  A 16-unit type (max-pool from the accumulators), a
    # 5-unit type and a 1-unit type.
    from dotaclient_amd import engine as E
    dev = torch.device('cuda:0')
    S, row = 16, 21
    LIMIT = 65504.0 / 16.0

    def values_for(x, products, flags):
        rollouts = synth.make_rollouts(5, [32, 32])
        rollouts[0]['observations'][key][row, unit, 7] = x
        eng = E.Engine('gru', 256, 1, dev)
        eng.products, eng.kernel_flags = products, flags
        eng.load_state_dict(synth.init_state_dict(7))
        batch = E.pack_rollouts(rollouts, S, dev)
        eng.rollout_pass(batch, S)
        return batch.values.cpu().numpy().copy(), batch.old_logp.cpu().numpy().copy()

    ref_v, ref_lp = values_for(0.9 * LIMIT, 'bf16x3', 0)
    assert np.isfinite(ref_v).all()
    for flags in (0, E.DC_DIMS_EMBED_UNFUSED):
        v, lp = values_for(0.9 * LIMIT, 'f16x2', flags)
        fin = np.isfinite(ref_lp)            # heads that did not act carry the reference's +inf (policy.py:172-177 on an all-False mask)
        assert np.isfinite(v).all() and np.array_equal(np.isfinite(lp), fin)
        assert util.scaled_err(v, ref_v) < 1e-5 and util.scaled_err(lp[fin], ref_lp[fin]) < 1e-5, (flags, util.scaled_err(v, ref_v))
    over_v, over_lp = values_for(1.01 * LIMIT, 'bf16x3', 0)
    assert np.isfinite(over_v).all()
    # just over the limit.  Fused kernels: the record's first f16 piece is inf -> NaN through the (NaN-propagating) relu, the products and the
    # max-pool into that env-step's outputs
    v, lp = values_for(1.01 * LIMIT, 'f16x2', 0)
    assert not np.isfinite(v[row]), 'an out-of-range record entry gave a finite value'
    assert np.isfinite(v[:row]).all() and np.isfinite(v[32:]).all()            # steps before it and the other rollout are untouched
    # layer-by-layer path: its first layer is f32 arithmetic on the records themselves (embed.hip), so this entry is simply in range there
    # (what must stay below 4094 on that path is the first layer's OUTPUT) - finite AND right
    v, lp = values_for(1.01 * LIMIT, 'f16x2', E.DC_DIMS_EMBED_UNFUSED)
    fin = np.isfinite(over_lp)
    assert np.isfinite(v).all() and util.scaled_err(v, over_v) < 1e-5 and util.scaled_err(lp[fin], over_lp[fin]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('lens', [[128] * 4, [256] * 6, [100, 150, 77], [33], [128] * 3, [24] * 16, [120, 8]])
@pytest.mark.parametrize('lazy', [False, True])
def test_fused_forward_pools_every_type_itself(lens, lazy):
    # round 6: the fused embedding forward also produces the env embedding (policy.py:97) and the five-unit type's max-pool (its tiles hold 24
    # whole env-steps, six per 32-row block) - there is no pool_env_fwd launch any more.  Against the same kernel followed by pool_env_fwd
    # (DC_DIMS_POOL_ENV_SEPARATE): xcat and the arg-max bytes bit for bit (the same arithmetic, "first maximum wins"), and everything downstream.
    # Step counts that are / are not multiples of 24, 33 steps inside 128 padded ones, one tile and many; lazy: with the action masks, the
    # five-unit type's emb rows are stored for targetable units only (the attention reads no others)
    from dotaclient_amd import engine as E
    from dotaclient_amd.engine import Engine, pack_rollouts
    dev = torch.device('cuda:0')
    outs = {}
    rollouts = synth.make_rollouts(92, lens)
    # a NaN statistic in one enemy hero of one step: the pooled value of that step must be NaN both ways (torch.max propagates it)
    rollouts[0]['observations']['enemy_heroes'][3, 2, 5] = float('nan')
    for mode in (0, E.DC_DIMS_POOL_ENV_SEPARATE):
        eng = Engine('lstm', 128, 1, dev)
        eng.kernel_flags = mode
        eng.load_state_dict(synth.init_state_dict(7, 'lstm', 128, 1))
        batch = pack_rollouts(rollouts, 128, dev)
        nr = batch.rows
        d, _, _ = eng.forward(batch, lazy_tu=lazy)
        xcat = eng.ws_view(d, 'XCAT')[:nr * 896].view(nr, 896).cpu()
        amax = eng.ws_view(d, 'AMAX', dtype=torch.uint8)[:nr * 3 * 128].view(nr, 3, 128).cpu()
        ho = eng.ws_view(d, 'HEADOUT')[:nr * L.HEADOUT_LD].cpu()
        if lazy:     # the target-unit logits of the unmasked units, through the masked log-softmax (reads the stored emb rows)
            logp, _, am = eng.select_logp(d, batch)
            tu = torch.cat([logp.flatten(), am.flatten().float()]).cpu()
        else:
            tu = eng.ws_view(d, 'TU')[:nr * 40].cpu()
        outs[mode] = (xcat, amax, ho, tu)
    a, b = outs[0], outs[E.DC_DIMS_POOL_ENV_SEPARATE]
    assert torch.isnan(a[0][3, 256:384]).all() and torch.isnan(b[0][3, 256:384]).all()
    for x, y, name in zip(a, b, ('xcat', 'amax', 'headout', 'tu')):
        if x.dtype == torch.uint8:
            assert torch.equal(x, y), name
        else:
            assert torch.equal(torch.nan_to_num(x, nan=12345.0), torch.nan_to_num(y, nan=12345.0)), name
