"""GPU: per-tensor gradient differences between the max-pool backward variants (dense kernels / on-chip dense / sparse VALU) on one small
batch, with the location of the largest difference - a debugging aid for csrc/embed_pool16m.hip."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from dotaclient_amd import engine as E, synth

dev = torch.device('cuda:0')
lens = [128] * 4
outs = {}
for mode, fl in (('dense', E.DC_DIMS_DENSE_POOL_BWD), ('mfma', 0), ('valu', E.DC_DIMS_POOL16_VALU)):
    eng = E.Engine('lstm', 128, 1, dev)
    eng.kernel_flags = fl
    eng.load_state_dict(synth.init_state_dict(7, 'lstm', 128, 1))
    batch = E.pack_rollouts(synth.make_rollouts(91, lens), 128, dev)
    chunks = eng.rollout_pass(batch, 128)
    eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
    outs[mode] = {n: eng.param_view(n, eng.grads).cpu().numpy().astype(np.float64).copy() for n in
                  ('affine_unit_basic_stats.weight', 'affine_unit_basic_stats.bias', 'affine_unit_anh.weight', 'affine_unit_enh.weight',
                   'affine_unit_anh.bias', 'affine_unit_enh.bias')}
for other in ('mfma', 'valu'):
    for n, b in outs['dense'].items():
        a = outs[other][n]
        d = np.abs(a - b)
        i = np.unravel_index(d.argmax(), d.shape)
        big = np.argwhere(d > 0.2 * d.max())
        print('%-5s vs dense %-34s scaled err %.2e  max|ref| %.3e  worst at %s (%.4e vs %.4e)  entries within 5x of worst: %d %s'
              % (other, n, d.max() / np.abs(b).max(), np.abs(b).max(), i, a[i], b[i], len(big), big[:12].tolist()))
