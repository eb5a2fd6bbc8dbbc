"""Offline fixture of the ORACLE (oracle/ref_optimizer.py, fp32 torch CPU) on the FULL per-GPU shard of BASELINE.json configs[4]:
2-layer LSTM-512, 256 trajectories x 512 steps = 131 072 env-steps, rollout pass + one epoch (the persistent
H = 512 kernels were only checked on 64 of the 256 sequences).  CPU minutes and tens of GB of autograd state, so it is run here,
once, and its outputs are committed as strided samples (tests/golden/cfg4_shard_oracle.npz); tests/test_gpu_bf16.py regenerates the
inputs from the seed and compares the HIP bf16 path against it at the stated bf16 tolerance.

No reference implementation exists for this cell: this is the restated oracle, "parity unpinned" by construction -
the fixture pins the GPU path to the oracle at the full shape, not the oracle to the reference.

    python tests/golden/make_cfg4_fixture.py [n_trajectories=256] [steps=512]
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from dotaclient_amd import synth            # noqa: E402
from tests import util                       # noqa: E402

SEED = 4242
STRIDE = 61


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    torch.set_num_threads(int(os.environ.get('THREADS', '8')))
    g = {'seq_len': S, 'lr': 5e-5, 'entropy_coef': 5e-4, 'vf_coef': 0.5, 'epochs': 1}
    t0 = time.time()
    rollouts = synth.make_rollouts(SEED, [S] * B)
    print('inputs %.1fs' % (time.time() - t0), flush=True)
    t0 = time.time()
    ref, _, _ = util.oracle_run(g, rollouts, 'lstm', 512, 2, epochs=1)
    print('oracle %.1fs' % (time.time() - t0), flush=True)
    out = {'B': np.int64(B), 'S': np.int64(S), 'seed': np.int64(SEED), 'stride': np.int64(STRIDE)}
    for key in ['advantages', 'returns', 'values'] + [k for k in ref if k.startswith('old_logp_')]:
        out[key] = np.ascontiguousarray(np.asarray(ref[key]).ravel()[::STRIDE], np.float32)
        out[key + '_max'] = np.float64(np.abs(ref[key]).max())
        out[key + '_n'] = np.int64(np.asarray(ref[key]).size)
    out['argmax_rows16'] = np.asarray(ref['argmax']).reshape(-1, 5)[::16].astype(np.int8)
    for key in ('ep0_losses', 'ep0_entropies', 'ep0_grad_norms', 'ep0_param_samples', 'ep0_grad_samples', 'ep0_grad_summary',
                'ep0_param_summary'):
        out[key] = np.asarray(ref[key])
    path = os.path.join(HERE, 'cfg4_shard_oracle.npz' if (B, S) == (256, 512) else 'cfg4_shard_oracle_%dx%d.npz' % (B, S))
    np.savez_compressed(path, **out)
    print(path, '%.1f KB' % (os.path.getsize(path) / 1024), 'losses', out['ep0_losses'])


if __name__ == '__main__':
    main()
