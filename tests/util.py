"""Shared helpers for the parity tests: load a golden case, rebuild its inputs, run the oracle."""
import os

import numpy as np
import torch

from dotaclient_amd import synth
from oracle import ref_optimizer as RO

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CASES = ['ragged_s16', 'clip_s16', 'cfg1_4x128', 'emptyhead_s16', 'noent_s16', 'novf_s16']
BIG_CASES = ['cfg2_gru_64x256']        # the real reference at a BASELINE.json batch (64 trajectories x 256 steps), 1 epoch
SAMPLE_STRIDE = 251


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    rollouts = synth.make_rollouts(int(g['data_seed']), [int(x) for x in g['lengths']],
                                   forbid_enum=tuple(int(x) for x in g['forbid_enum']))
    return g, rollouts


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-30))) if a.size else 0.0


def scaled_err(a, b):
    """max |a-b| / max|b| : the right yardstick for vectors whose entries cross zero."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if a.size else 0.0


def elementwise_rel_err(a, b, floor=1e-3):
    """The literal reading of north_star's "within 1e-4 relative": max over the entries with |b| > floor * max|b| of
    |a - b| / |b| (entries closer to zero than that carry no relative information in float32: a value of 1e-3 * max computed
    from O(max) terms has an absolute error of O(1e-7 * max), i.e. 1e-4 of itself, from rounding alone).  Returned as
    (worst, fraction of entries above the floor); reported next to scaled_err by the parity tests and bench.py."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    if not a.size:
        return 0.0, 0.0
    keep = np.abs(b) > floor * np.max(np.abs(b))
    if not keep.any():
        return 0.0, 0.0
    return float(np.max(np.abs(a[keep] - b[keep]) / np.abs(b[keep]))), float(keep.mean())


def loss_rel_err(a, b):
    """Relative error of the four loss numbers [loss, policy, entropy, value] (north_star: 1e-4).  `loss` is the SUM of
    the three parts and cancels (clip_s16, third epoch at lr 3e-3: -0.0028 = -0.0836 - 0.0040 + 0.0848, condition number
    62): an error of 1e-4 in each part is an error of 1e-4 * (|policy| + |entropy| + |value|) in the sum, so that is what
    the sum is measured against.  `policy_loss` is itself a mean of signed terms; each part is measured against
    max(|itself|, 1 % of the largest part)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), 0.01 * np.max(np.abs(b[1:4])))
    den[0] = np.sum(np.abs(b[1:4]))
    return float(np.max(np.abs(a - b) / (den + 1e-30)))


def tensor_summary(t):
    f = t.detach().double().flatten().cpu()
    return np.array([f.sum().item(), f.abs().sum().item(), f.norm().item()]), \
        t.detach().flatten()[::SAMPLE_STRIDE][:1000].float().cpu().numpy().copy()


def oracle_run(g, rollouts, cell='gru', hidden=256, layers=1, epochs=None):
    """Runs the oracle restatement on a golden case's inputs; returns dict shaped like the fixture."""
    torch.manual_seed(0)
    pol = RO.make_policy(synth.init_state_dict(7, cell, hidden, layers), cell, hidden, layers)
    opt = torch.optim.Adam(pol.parameters(), lr=float(g['lr']))
    S = int(g['seq_len'])
    chunks = []
    for r in rollouts:
        chunks.extend(RO.rollout_pass(pol, r, S))
    out = {
        'advantages': torch.stack([c.advantages for c in chunks]).numpy(),
        'returns': torch.stack([c.returns for c in chunks]).numpy(),
        'values': torch.stack([c.values for c in chunks]).numpy(),
        'argmax': RO.masked_argmax(pol, chunks).numpy(),
    }
    for k in RO.HEADS:
        out['old_logp_' + k] = torch.cat([c.old_logp[k] for c in chunks]).numpy()
    n_ep = int(g['epochs']) if epochs is None else epochs
    for ep in range(n_ep):
        parts, ent, norms = RO.train_step(pol, opt, chunks, float(g['entropy_coef']), float(g['vf_coef']))
        out['ep%d_losses' % ep] = np.array([float(parts[k]) for k in ('loss', 'policy_loss', 'entropy_loss', 'value_loss')])
        out['ep%d_entropies' % ep] = np.array([float(ent[k]) for k in RO.HEADS])
        out['ep%d_grad_norms' % ep] = np.array([float(norms['unclipped']), float(norms['clipped'])])
        gs, gv, ps, pv, hg = [], [], [], [], []
        for n, p in pol.named_parameters():
            hg.append(p.grad is not None)
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            s, v = tensor_summary(gr); gs.append(s); gv.append(v)
            s, v = tensor_summary(p); ps.append(s); pv.append(v)
        out['ep%d_grad_summary' % ep] = np.stack(gs)
        out['ep%d_grad_samples' % ep] = np.concatenate(gv)
        out['ep%d_param_summary' % ep] = np.stack(ps)
        out['ep%d_param_samples' % ep] = np.concatenate(pv)
        out['ep%d_has_grad' % ep] = np.array(hg)
    out['param_names'] = np.array([n for n, _ in pol.named_parameters()])
    return out, pol, chunks


def load_dp_case(name='dp2_s16'):
    """Fixture of the reference's own DP wrapper under gloo (make_golden.py::run_dp_case): returns the fixture and the
    per-rank rollouts."""
    g = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    shards = [synth.make_rollouts(int(g['r%d_data_seed' % r]), [int(x) for x in g['r%d_lengths' % r]],
                                  forbid_enum=tuple(int(x) for x in g['r%d_forbid_enum' % r])) for r in range(int(g['world']))]
    return g, shards


def oracle_dp_run(g, shards):
    """The oracle's N-rank emulation (RO.dp_train_step) on a DP fixture's inputs; dict shaped like the fixture."""
    world, S = int(g['world']), int(g['seq_len'])
    sd = synth.init_state_dict(7)
    pols = [RO.make_policy(sd) for _ in range(world)]
    opts = [torch.optim.Adam(p.parameters(), lr=float(g['lr'])) for p in pols]
    chunk_shards = [[c for r in sh for c in RO.rollout_pass(pol, r, S)] for pol, sh in zip(pols, shards)]
    out = {}
    for r in range(world):
        out['r%d_advantages' % r] = torch.stack([c.advantages for c in chunk_shards[r]]).numpy()
        out['r%d_returns' % r] = torch.stack([c.returns for c in chunk_shards[r]]).numpy()
    for ep in range(int(g['epochs'])):
        res = RO.dp_train_step(pols, opts, chunk_shards, float(g['entropy_coef']), float(g['vf_coef']))
        for r, (parts, ent, norms) in enumerate(res):
            pre = 'r%d_ep%d_' % (r, ep)
            out[pre + 'losses'] = np.array([float(parts[k]) for k in ('loss', 'policy_loss', 'entropy_loss', 'value_loss')])
            out[pre + 'entropies'] = np.array([float(ent[k]) for k in RO.HEADS])
            out[pre + 'grad_norms'] = np.array([float(norms['unclipped']), float(norms['clipped'])])
            gv, pv, hg = [], [], []
            for n, p in pols[r].named_parameters():
                hg.append(p.grad is not None)
                gr = p.grad if p.grad is not None else torch.zeros_like(p)
                gv.append(tensor_summary(gr)[1]); pv.append(tensor_summary(p)[1])
            out[pre + 'grad_samples'] = np.concatenate(gv)
            out[pre + 'param_samples'] = np.concatenate(pv)
            out[pre + 'has_grad'] = np.array(hg)
    return out


def gae_long_inputs(n=50000, seed=50):
    """Inputs of tests/golden/gae_long.npz (rollouts longer than one LDS block of the HIP scan): (n+1)-vectors of rewards and
    values for advantage_returns, a (2n+1)-vector for discount.  make_golden.py runs the real reference on exactly these."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = (0.3 * rng.standard_normal(n + 1)).astype(np.float32)
    v = rng.standard_normal(n + 1).astype(np.float32)
    x = rng.standard_normal(2 * n + 1).astype(np.float32)
    return r, v, x


def sha256_of(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).digest(), np.uint8)
