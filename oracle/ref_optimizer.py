"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference PPO optimizer hot path in numpy / plain PyTorch fp32:

    discount, advantage_returns      /root/reference/optimizer.py:53-64
    experiences_from_rollout         /root/reference/optimizer.py:328-430
    train (+ mean_gradient_norm)     /root/reference/optimizer.py:581-695
    DP gradient averaging            /root/reference/distributed.py:24-57

Third-party arithmetic the reference delegates to (absent from /root/reference): scipy==1.2.0
`lfilter` (docker/Dockerfile:18) - restated here as the float64 direct-form recurrence it
implements; torch==1.0.0 autograd / Adam / clip_grad_norm_ (docker/Dockerfile:16) - used here
as torch 2.10 CPU ops (same maths; `zero_grad()` now leaves `grad=None`, so a head that had no
action in the batch is skipped by Adam - SURVEY.md section 8(c)).

Parity status: PINNED - tests/test_oracle.py checks every function below against
tests/golden/*.npz (outputs of the real reference imported from /root/reference by
tests/golden/make_golden.py) and against the known-answer vector of SURVEY.md section 4.
Masks/actions are torch.bool (uint8 no longer works as a mask in torch>=1.2; bool reproduces the
torch-1.0 semantics the reference was written for).
"""
import numpy as np
import torch

from .ref_policy import RefPolicy, masked_log_softmax, HEAD_WIDTHS

EPS = float(np.finfo(np.float32).eps)          # optimizer.py:38
HEADS = list(HEAD_WIDTHS.keys())
OBS_KEYS = ['env', 'allied_heroes', 'enemy_heroes', 'allied_nonheroes', 'enemy_nonheroes',
            'allied_towers', 'enemy_towers']


def discount(x, gamma):
    """optimizer.py:53-54.  lfilter([1],[1,-gamma]) on the reversed signal: y[t] = x[t] + gamma*y[t+1],
    accumulated in float64 (lfilter promotes to the coefficient dtype), cast to float32 at the end."""
    x = np.asarray(x)
    y = np.empty(x.shape[0], dtype=np.float64)
    acc = 0.0
    g = float(gamma)
    for t in range(x.shape[0] - 1, -1, -1):
        acc = float(x[t]) + g * acc
        y[t] = acc
    return y.astype(np.float32)


def advantage_returns(rewards, values, gamma, lam):
    """optimizer.py:57-64.  rewards, values: float32 (L+1,), last element the appended terminal 0."""
    rewards = np.asarray(rewards, dtype=np.float32)
    values = np.asarray(values, dtype=np.float32)
    deltas = rewards[:-1] + np.float32(gamma) * values[1:] - values[:-1]       # float32 arithmetic
    advantages = discount(deltas, gamma * lam)
    returns = discount(rewards, gamma)[:-1]
    return advantages, returns


class Chunk:
    """One seq_len slice of a rollout (the reference's `Sequence`, optimizer.py:176-190)."""
    __slots__ = ('obs', 'actions', 'masks', 'values', 'rewards', 'hidden', 'old_logp',
                 'advantages', 'returns', 'logits')


def _pad_time(t, pad):
    if pad == 0:
        return t
    shape = (pad,) + tuple(t.shape[1:])
    return torch.cat([t, torch.zeros(shape, dtype=t.dtype)], dim=0)


def rollout_pass(policy, data, seq_len, gamma=0.98, lam=0.97):
    """optimizer.py:328-430 for one rollout dict -> list[Chunk].  Runs under no_grad like run() does
    (optimizer.py:452)."""
    T = data['rewards'].shape[0]
    chunks, all_values, all_rewards = [], [], []
    hidden = policy.init_hidden(1)
    with torch.no_grad():
        for lo in range(0, T, seq_len):                                # optimizer.py:343-350
            hi = min(lo + seq_len, T)
            pad = seq_len - (hi - lo)
            c = Chunk()
            c.obs = {k: _pad_time(data['observations'][k][lo:hi].float(), pad) for k in OBS_KEYS}
            c.masks = {k: _pad_time(data['masks'][k][lo:hi].bool(), pad) for k in HEADS}
            c.actions = {k: _pad_time(data['actions'][k][lo:hi].bool(), pad) for k in HEADS}
            rew = np.asarray(data['rewards'][lo:hi], dtype=np.float32)
            if pad:
                rew = np.concatenate([rew, np.zeros((pad, rew.shape[1]), np.float32)], axis=0)
            c.rewards = rew
            c.hidden = hidden                                          # optimizer.py:384,408
            logits, values, hidden = policy({k: v.unsqueeze(0) for k, v in c.obs.items()}, hidden)
            c.logits = {k: v[0] for k, v in logits.items()}
            c.old_logp = {}
            for k in HEADS:                                            # optimizer.py:387-390
                lp = masked_log_softmax(logits[k], c.masks[k].unsqueeze(0))
                c.old_logp[k] = lp[0][c.actions[k]]
            c.values = values[0, :, 0]
            all_values.append(c.values.numpy())
            all_rewards.append(rew.sum(axis=1))                        # optimizer.py:397
            chunks.append(c)
    values = np.append(np.concatenate(all_values), np.float32(0.))     # optimizer.py:417-420
    rewards = np.append(np.concatenate(all_rewards), np.float32(0.))
    adv, ret = advantage_returns(rewards, values, gamma, lam)          # optimizer.py:421
    for i, c in enumerate(chunks):                                     # optimizer.py:424-428
        c.advantages = torch.from_numpy(adv[i * seq_len:(i + 1) * seq_len].copy())
        c.returns = torch.from_numpy(ret[i * seq_len:(i + 1) * seq_len].copy())
    return chunks


def mean_gradient_norm(params):
    """optimizer.py:691-695: mean over parameters (that have a grad) of the per-parameter L2 norm."""
    return torch.stack([p.grad.norm(2) for p in params if p.grad is not None]).mean()


def stack_hidden(chunks):
    if isinstance(chunks[0].hidden, tuple):                            # lstm: (h, c)
        return tuple(torch.cat([c.hidden[i] for c in chunks], dim=1) for i in range(2))
    return torch.cat([c.hidden for c in chunks], dim=1)                # optimizer.py:591


def ppo_loss(policy, chunks, entropy_coef, vf_coef, e_clip=0.1):
    """optimizer.py:587-665.  Returns (loss, parts dict, entropies dict, logits dict, values)."""
    adv = torch.stack([c.advantages for c in chunks])
    adv = (adv - adv.mean()) / (adv.std() + EPS)                       # optimizer.py:588 (unbiased std)
    ret = torch.stack([c.returns for c in chunks])
    obs = {k: torch.stack([c.obs[k] for c in chunks]) for k in OBS_KEYS}
    logits, values, _ = policy(obs, stack_hidden(chunks))
    pol, ent = {}, {}
    for k in HEADS:
        act = torch.stack([c.actions[k] for c in chunks])
        msk = torch.stack([c.masks[k] for c in chunks])
        step_sel = act.sum(dim=-1) != 0                                # optimizer.py:626
        n_sel = step_sel.sum()
        if n_sel == 0:                                                 # optimizer.py:627-630
            pol[k] = torch.zeros([])
            ent[k] = torch.zeros([])
            continue
        lp = masked_log_softmax(logits[k], msk)
        ratio = torch.exp(lp[act] - torch.cat([c.old_logp[k] for c in chunks]))
        a = adv[step_sel]
        pol[k] = -torch.min(ratio * a, torch.clamp(ratio, 1.0 - e_clip, 1.0 + e_clip) * a).mean()
        lpm = lp[msk]
        ent[k] = -(torch.exp(lpm) * lpm).sum() / n_sel                 # optimizer.py:643-646
    policy_loss = torch.stack(list(pol.values())).mean()               # optimizer.py:649-650
    entropy_loss = -entropy_coef * torch.stack(list(ent.values())).sum() if entropy_coef > 0 \
        else torch.tensor(0.)
    value_loss = vf_coef * 0.5 * (ret - values.squeeze(-1)).pow(2).mean() if vf_coef > 0 \
        else torch.tensor(0.)
    loss = policy_loss + entropy_loss + value_loss
    parts = {'loss': loss, 'policy_loss': policy_loss, 'entropy_loss': entropy_loss, 'value_loss': value_loss}
    return loss, parts, ent, logits, values


def train_step(policy, optimizer, chunks, entropy_coef, vf_coef, e_clip=0.1, max_grad_norm=0.5,
               grad_hook=None):
    """optimizer.py:581-689: one full-batch epoch.  `grad_hook(params)` runs after backward (the
    slot where the reference's DP wrapper all-reduces, distributed.py:24-57)."""
    loss, parts, ent, _, _ = ppo_loss(policy, chunks, entropy_coef, vf_coef, e_clip)
    if torch.isnan(loss):
        raise ValueError('loss is NaN')
    optimizer.zero_grad()
    loss.backward()
    params = list(policy.parameters())
    if grad_hook is not None:
        grad_hook(params)
    unclipped = mean_gradient_norm(params)
    torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
    clipped = mean_gradient_norm(params)
    if torch.isnan(unclipped):
        raise ValueError('grad norm is NaN')
    optimizer.step()
    detach = lambda d: {k: v.detach() for k, v in d.items()}
    return detach(parts), detach(ent), {'unclipped': unclipped, 'clipped': clipped}


def masked_argmax(policy, chunks):
    """Per-step argmax over the masked log-probs of every head ((B,S,5) int64; -1 where the mask row is
    empty) - the 'action argmax' of BASELINE.json's north_star as defined in SURVEY.md section 8(c)."""
    with torch.no_grad():
        obs = {k: torch.stack([c.obs[k] for c in chunks]) for k in OBS_KEYS}
        logits, _, _ = policy(obs, stack_hidden(chunks))
        out = []
        for k in HEADS:
            msk = torch.stack([c.masks[k] for c in chunks])
            lp = masked_log_softmax(logits[k], msk)
            lp = torch.where(msk, lp, torch.full_like(lp, -float('inf')))
            idx = lp.argmax(dim=-1)
            idx[~msk.any(dim=-1)] = -1
            out.append(idx)
    return torch.stack(out, dim=-1)


def dp_average_grads(per_rank_grads):
    """distributed.py:24-57 emulated for N ranks in one process: per parameter, sum the gradients of
    the ranks that have one and divide by how many did; ranks without a grad keep None.
    per_rank_grads: list (rank) of list (param) of tensor-or-None.  Returns the same structure."""
    n_ranks, n_params = len(per_rank_grads), len(per_rank_grads[0])
    out = [[None] * n_params for _ in range(n_ranks)]
    for j in range(n_params):
        have = [r for r in range(n_ranks) if per_rank_grads[r][j] is not None]
        if not have:
            continue
        total = sum(per_rank_grads[r][j] for r in have) / len(have)
        for r in have:
            out[r][j] = total.clone()
    return out


def dp_train_step(policies, optimizers, shards, entropy_coef, vf_coef, e_clip=0.1, max_grad_norm=0.5):
    """One epoch of N data-parallel ranks emulated in one process: every rank runs optimizer.py:581-672 on ITS shard
    (own advantage normalisation, own loss means), the gradients are exchanged as distributed.py:24-57 does
    (dp_average_grads), then every rank finishes optimizer.py:674-681 (norm metric, clip, Adam) on what it holds.
    Returns one (parts, entropies, norms) triple per rank.  Pinned against the reference's own wrapper run under two
    gloo ranks (tests/golden/dp2_s16.npz)."""
    parts_all, ent_all = [], []
    for pol, opt, chunks in zip(policies, optimizers, shards):
        loss, parts, ent, _, _ = ppo_loss(pol, chunks, entropy_coef, vf_coef, e_clip)
        opt.zero_grad()
        loss.backward()
        parts_all.append(parts); ent_all.append(ent)
    avg = dp_average_grads([[p.grad for p in pol.parameters()] for pol in policies])
    out = []
    for r, (pol, opt) in enumerate(zip(policies, optimizers)):
        params = list(pol.parameters())
        for p, g in zip(params, avg[r]):
            p.grad = g
        unclipped = mean_gradient_norm(params)
        torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
        clipped = mean_gradient_norm(params)
        opt.step()
        detach = lambda d: {k: v.detach() for k, v in d.items()}
        out.append((detach(parts_all[r]), detach(ent_all[r]), {'unclipped': unclipped, 'clipped': clipped}))
    return out


def make_policy(state_dict, cell='gru', hidden=256, layers=1):
    p = RefPolicy(cell, hidden, layers)
    p.load_state_dict(state_dict, strict=True)
    return p
