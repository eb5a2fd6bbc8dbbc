"""GPU: the bf16 MFMA path (BASELINE.json configs[4]: "5v5 hidden=512 2-layer LSTM, bf16 MFMA path") against the fp32 oracle.

DC_DIMS_BF16 rounds the operands of the dense products - affine_pre_rnn, the recurrent cell's input projections, the head
projections and all their gradient products (policy.py:138-155 and autograd's products for them) - to bf16 (8 mantissa bits),
one v_mfma_f32_32x32x16_bf16 per K = 16 with f32 accumulation; at H = 512 (no register-resident recurrent kernel) the
recurrent products W_hh h and their BPTT counterparts too (csrc/rnn_step_bf16.hip: bf16 operands, f32 accumulate, f32 state
and gate maths; since round 5 on the persistent LSTM-512 kernels the gate pre-activations / activations, the gate gradients, `pre`,
`hseq` and `hprev` are also STORED as bf16 - DC_DIMS_BF16_F32_STORE keeps them f32); the unit embeddings, the cell state, the
H <= 256 recurrences, the loss, the norms and Adam stay f32.  There is no bf16 reference ("compared to this fp32 oracle with a looser,
separately-stated tolerance"), so the tolerances below ARE the statement:

  quantity (one epoch from the same weights, oracle fp32)          tolerance      why
  values / old log-probs / advantages (scaled by max |ref|)        1e-2           three chained bf16 products, K up to 896: ~2^-9 * sqrt(depth); measured 2e-3 .. 5e-3
  losses (util.loss_rel_err), entropies, gradient norms            1e-2           means over >= 10^3 steps of the above; measured <= 4e-3
  masked argmax indices                                            >= 99.5 % equal  near-ties flip under a 2^-9 perturbation of the logits; measured 99.9 .. 100 %
  (round 6: the bars were 3e-2 / 97 % - eight times what is measured, VERDICT r5 - and are now ~2x the measured figures)
  post-step parameters (scaled)                                    1e-3           one Adam step moves a parameter by <= lr = 5e-5 whatever the gradient
  bf16 storage against f32 storage of the same kernels             1e-2           one more bf16 rounding of the input projections, the stored gates and gate gradients
  persistent LSTM-512 kernels against the launch-per-step ones      5e-3           same arithmetic (both f32-stored), another summation order
"""
import numpy as np
import pytest
import torch

from dotaclient_amd import synth
from tests import util
from tests.test_gpu_parity import run_hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cell,hidden,layers,lens,S', [('lstm', 512, 2, [64] * 6, 64), ('lstm', 256, 1, [50, 64, 33], 16), ('gru', 256, 1, [128] * 4, 128),
                                                       # rows % 16 != 0 on configs[4]'s cell (5 rollouts x 10 steps = 50 rows): the weight gradients contract
                                                       # over the rows in K = 16 steps, so these dims must keep f32 storage and the f32 fallback
                                                       # products instead of failing with error 1005 (policy.hip bf16_store())
                                                       ('lstm', 512, 2, [10] * 5, 10),
                                                       # a sub-batch of BASELINE.json configs[4]'s shard: 64 of its 256 trajectories x 512 steps
                                                       ('lstm', 512, 2, [512] * 64, 512)])
def test_bf16_path_within_stated_tolerance_of_fp32_oracle(cell, hidden, layers, lens, S):
    from dotaclient_amd import engine as E
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(4242, lens)
    ref, _, _ = util.oracle_run(g, rollouts, cell, hidden, layers, epochs=1)
    out, _ = run_hip(g, rollouts, cell, hidden, layers, epochs=1, kernel_flags=E.DC_DIMS_BF16)
    f32, _ = run_hip(g, rollouts, cell, hidden, layers, epochs=1)
    errs = {}
    for key in ['advantages', 'values'] + ['old_logp_' + k for k in ('enum', 'x', 'y', 'target_unit', 'ability')]:
        errs[key] = util.scaled_err(out[key], ref[key])
        assert util.scaled_err(f32[key], ref[key]) < 1e-4, key          # the default path on the same inputs: the fp32 bar
    assert np.array_equal(out['returns'], f32['returns'])               # returns do not depend on the network
    same = float((out['argmax'] == ref['argmax'].reshape(out['argmax'].shape)).mean())
    errs['losses'] = util.loss_rel_err(out['ep0_losses'], ref['ep0_losses'])
    errs['entropies'] = util.rel_err(out['ep0_entropies'], ref['ep0_entropies'])
    errs['grad_norms'] = util.rel_err(out['ep0_grad_norms'], ref['ep0_grad_norms'])
    print('bf16 path vs fp32 oracle:', {k: float('%.3g' % v) for k, v in errs.items()}, 'argmax equal %.5f' % same)
    for k, v in errs.items():
        assert v < 1e-2, (k, errs)
    assert same >= 0.995, same
    assert util.scaled_err(out['ep0_param_samples'], ref['ep0_param_samples']) < 1e-3
    # and it is really a different arithmetic: the bf16 run must NOT meet the fp32 bar
    assert util.scaled_err(out['values'], ref['values']) > 1e-5


@pytest.mark.parametrize('tile32', [False, True])
@pytest.mark.parametrize('lens,S', [([64] * 6, 64), ([50, 64, 33, 7, 100, 64, 1, 16] * 5, 16), ([256] * 40, 256)])
def test_persistent_lstm512_matches_the_step_kernels(lens, S, tile32):
    # rnn_team512.hip (the whole time loop of an LSTM-512 layer in one launch: sixteen workgroups hold W_hh as bf16 in registers and hand
    # h_t / the partial dh sums around as tagged granules) against rnn_step_bf16.hip (one launch per time step) - the same bf16-operand /
    # f32-accumulate arithmetic in another summation order.  Not bit-comparable: h_t is ROUNDED to bf16 before it becomes the next
    # step's operand, so a 1-ulp f32 difference next to a rounding boundary becomes a 2^-9 difference in that element (the first
    # steps agree to 8 digits, then single elements flip); and the persistent backward rounds its sixteen PARTIAL recurrent sums to
    # bf16 before adding them.  Both are bf16-path approximations of the same f32 function (bar against the oracle: 3e-2, above);
    # against each other they stay an order of magnitude closer than that.
    # Shapes: one partial tile; ragged rollouts -> 4 tiles of chunks incl. a partial one and sequences of 1 step; two tiles x 256 steps.
    from dotaclient_amd import engine as E
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(77, lens)
    # tile32: 32-sequence tiles forced (DC_DIMS_TEAM_NS(2)); the default takes 16-sequence tiles while the batch fits one round of teams
    # (f32 gate buffers on both sides, DC_DIMS_BF16_F32_STORE: the step kernels have no bf16 storage; the default storage is compared below)
    team, eng = run_hip(g, rollouts, 'lstm', 512, 2, epochs=1,
                        kernel_flags=E.DC_DIMS_BF16 | E.DC_DIMS_BF16_F32_STORE | (E.DC_DIMS_TEAM_NS(2) if tile32 else 0))
    step, _ = run_hip(g, rollouts, 'lstm', 512, 2, epochs=1, kernel_flags=E.DC_DIMS_BF16 | E.DC_DIMS_RNN_STEP_BF16)
    assert eng.fault() is None
    errs = {key: util.scaled_err(team[key], step[key]) for key in ['advantages', 'values', 'hidden'] + ['old_logp_' + k for k in ('enum', 'x', 'y', 'target_unit', 'ability')]}
    errs['argmax_mismatch'] = float((team['argmax'] != step['argmax']).mean())
    errs['losses'] = util.loss_rel_err(team['ep0_losses'], step['ep0_losses'])
    errs['grad_norms'] = util.rel_err(team['ep0_grad_norms'], step['ep0_grad_norms'])
    errs['grad_samples'] = util.scaled_err(team['ep0_grad_samples'], step['ep0_grad_samples'])
    errs['grad_tensor_norms'] = util.scaled_err(team['ep0_grad_summary'][:, 2], step['ep0_grad_summary'][:, 2])
    print('team512 vs step kernels:', {k: float('%.3g' % v) for k, v in errs.items()})
    for k, v in errs.items():
        assert v < (5e-3 if k != 'argmax_mismatch' else 5e-3), (k, errs)


@pytest.mark.parametrize('tile32', [False, True])
@pytest.mark.parametrize('lens,S', [([64] * 6, 64), ([50, 64, 33, 7, 100, 64, 1, 16] * 5, 16), ([256] * 40, 256)])
def test_bf16_storage_of_the_gate_buffers_against_f32_storage(lens, S, tile32):
    # Round 5: on this path the gate pre-activations / activated gates and the gate gradients live in HBM as bf16 (policy.hip bf16_store():
    # gemm_x3 writes / reads them with its no-arithmetic loaders, rnn_team512.hip with 8-byte accesses).  Against the same kernels on
    # f32 buffers (DC_DIMS_BF16_F32_STORE) this adds ONE bf16 rounding of the input projections in the forward and one of the stored
    # gate activations / gate gradients in the backward - the same order as the operand roundings of the products; stated bar 1e-2 of
    # max |ref| (the bar against the fp32 oracle stays 3e-2: test_bf16_path_within_stated_tolerance_of_fp32_oracle runs the default).
    from dotaclient_amd import engine as E
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(78, lens)
    ns = E.DC_DIMS_TEAM_NS(2) if tile32 else 0
    b16, eng = run_hip(g, rollouts, 'lstm', 512, 2, epochs=1, kernel_flags=E.DC_DIMS_BF16 | ns)
    f32, _ = run_hip(g, rollouts, 'lstm', 512, 2, epochs=1, kernel_flags=E.DC_DIMS_BF16 | E.DC_DIMS_BF16_F32_STORE | ns)
    assert eng.fault() is None
    errs = {key: util.scaled_err(b16[key], f32[key]) for key in ['advantages', 'values', 'hidden'] + ['old_logp_' + k for k in ('enum', 'x', 'y', 'target_unit', 'ability')]}
    errs['argmax_mismatch'] = float((b16['argmax'] != f32['argmax']).mean())
    errs['losses'] = util.loss_rel_err(b16['ep0_losses'], f32['ep0_losses'])
    errs['grad_norms'] = util.rel_err(b16['ep0_grad_norms'], f32['ep0_grad_norms'])
    errs['grad_samples'] = util.scaled_err(b16['ep0_grad_samples'], f32['ep0_grad_samples'])
    errs['grad_tensor_norms'] = util.scaled_err(b16['ep0_grad_summary'][:, 2], f32['ep0_grad_summary'][:, 2])
    print('bf16 vs f32 storage of the gate buffers:', {k: float('%.3g' % v) for k, v in errs.items()})
    assert max(errs[k] for k in errs if k != 'argmax_mismatch') > 0.0        # it IS another storage
    for k, v in errs.items():
        assert v < 1e-2, (k, errs)


def test_bf16_path_on_the_full_configs4_shard_against_the_oracle_fixture():
    # the persistent H = 512 kernels at BASELINE.json configs[4]'s FULL per-GPU shard (2-layer LSTM-512, 256
    # trajectories x 512 steps: sixteen teams, one 16-sequence tile each, 512 time steps per launch) against the fp32 oracle's outputs on
    # the same seeded inputs - computed offline (tests/golden/make_cfg4_fixture.py, ~30 s and tens of GB of autograd state on the host)
    # and committed as strided samples (tests/golden/cfg4_shard_oracle.npz).  Same stated bf16 tolerances as above.
    from dotaclient_amd import engine as E
    f = np.load(util.GOLDEN + '/cfg4_shard_oracle.npz')
    B, S, stride = int(f['B']), int(f['S']), int(f['stride'])
    assert (B, S) == (256, 512)
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(int(f['seed']), [S] * B)
    out, eng = run_hip(g, rollouts, 'lstm', 512, 2, epochs=1, kernel_flags=E.DC_DIMS_BF16)
    assert eng.fault() is None
    errs = {}
    for key in ['advantages', 'values'] + ['old_logp_' + k for k in ('enum', 'x', 'y', 'target_unit', 'ability')]:
        got = np.asarray(out[key]).ravel()
        assert got.size == int(f[key + '_n']), key
        errs[key] = float(np.abs(got[::stride] - f[key]).max() / float(f[key + '_max']))
        assert errs[key] < 1e-2, (key, errs[key])
    # returns do not depend on the network: bit-exact against the oracle's
    assert np.array_equal(np.asarray(out['returns']).ravel()[::stride], f['returns'])
    same = float((out['argmax'].reshape(-1, 5)[::16] == f['argmax_rows16']).mean())
    errs['argmax_equal'] = same
    assert same >= 0.995, same
    errs['losses'] = util.loss_rel_err(out['ep0_losses'], f['ep0_losses'])
    errs['entropies'] = util.rel_err(out['ep0_entropies'], f['ep0_entropies'])
    errs['grad_norms'] = util.rel_err(out['ep0_grad_norms'], f['ep0_grad_norms'])
    errs['param_samples'] = util.scaled_err(out['ep0_param_samples'], f['ep0_param_samples'])
    errs['grad_tensor_norms'] = util.scaled_err(out['ep0_grad_summary'][:, 2], f['ep0_grad_summary'][:, 2])
    print('configs[4] full shard, bf16 path vs fp32 oracle fixture:', {k: float('%.3g' % v) for k, v in errs.items()})
    assert errs['losses'] < 1e-2 and errs['entropies'] < 1e-2 and errs['grad_norms'] < 1e-2 and errs['param_samples'] < 1e-3
    assert errs['grad_tensor_norms'] < 2e-2
