"""GPU: the bf16 MFMA path (BASELINE.json configs[4]: "5v5 hidden=512 2-layer LSTM, bf16 MFMA path") against the fp32 oracle.

DC_DIMS_BF16 rounds the operands of the dense products - affine_pre_rnn, the recurrent cell's input projections, the head
projections and all their gradient products (policy.py:138-155 and autograd's products for them) - to bf16 (8 mantissa bits),
one v_mfma_f32_32x32x16_bf16 per K = 16 with f32 accumulation; at H = 512 (no register-resident recurrent kernel) the
recurrent products W_hh h and their BPTT counterparts too (csrc/rnn_step_bf16.hip: bf16 operands, f32 accumulate, f32 state
and gate maths); the unit embeddings, the H <= 256 recurrences, the loss, the norms and Adam stay f32.  There is no bf16 reference ("compared to this fp32 oracle with a looser,
separately-stated tolerance"), so the tolerances below ARE the statement:

  quantity (one epoch from the same weights, oracle fp32)          tolerance      why
  values / old log-probs / advantages (scaled by max |ref|)        3e-2           three chained bf16 products, K up to 896: ~2^-9 * sqrt(depth)
  losses (util.loss_rel_err), entropies, gradient norms            3e-2           means over >= 10^3 steps of the above
  masked argmax indices                                            >= 97 % equal  near-ties flip under a 2^-9 perturbation of the logits
  post-step parameters (scaled)                                    1e-3           one Adam step moves a parameter by <= lr = 5e-5 whatever the gradient
"""
import numpy as np
import pytest
import torch

from dotaclient_amd import synth
from tests import util
from tests.test_gpu_parity import run_hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cell,hidden,layers,lens,S', [('lstm', 512, 2, [64] * 6, 64), ('lstm', 256, 1, [50, 64, 33], 16), ('gru', 256, 1, [128] * 4, 128),
                                                       # a sub-batch of BASELINE.json configs[4]'s shard: 64 of its 256 trajectories x 512 steps
                                                       ('lstm', 512, 2, [512] * 64, 512)])
def test_bf16_path_within_stated_tolerance_of_fp32_oracle(cell, hidden, layers, lens, S):
    from dotaclient_amd import engine as E
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    rollouts = synth.make_rollouts(4242, lens)
    ref, _, _ = util.oracle_run(g, rollouts, cell, hidden, layers, epochs=1)
    out, _ = run_hip(g, rollouts, cell, hidden, layers, epochs=1, kernel_flags=E.DC_DIMS_BF16)
    f32, _ = run_hip(g, rollouts, cell, hidden, layers, epochs=1)
    for key in ['advantages', 'values'] + ['old_logp_' + k for k in ('enum', 'x', 'y', 'target_unit', 'ability')]:
        assert util.scaled_err(out[key], ref[key]) < 3e-2, (key, util.scaled_err(out[key], ref[key]))
        assert util.scaled_err(f32[key], ref[key]) < 1e-4, key          # the default path on the same inputs: the fp32 bar
    assert np.array_equal(out['returns'], f32['returns'])               # returns do not depend on the network
    same = (out['argmax'] == ref['argmax'].reshape(out['argmax'].shape)).mean()
    assert same >= 0.97, same
    assert util.loss_rel_err(out['ep0_losses'], ref['ep0_losses']) < 3e-2
    assert util.rel_err(out['ep0_entropies'], ref['ep0_entropies']) < 3e-2
    assert util.rel_err(out['ep0_grad_norms'], ref['ep0_grad_norms']) < 3e-2
    assert util.scaled_err(out['ep0_param_samples'], ref['ep0_param_samples']) < 1e-3
    # and it is really a different arithmetic: the bf16 run must NOT meet the fp32 bar
    assert util.scaled_err(out['values'], ref['values']) > 1e-5
