"""Deterministic synthetic rollouts and parameter initialisation.

The generator follows SURVEY.md section 8(d): the produced dict has exactly the wire format that the
reference optimizer unpickles (/root/reference/optimizer.py:314-326, packed by
/root/reference/agent.py:351-416) and honours the actor's invariants
(`selected_heads_mask = head_mask & action_mask`, /root/reference/agent.py:666-671;
enum -> sub-head routing, /root/reference/policy.py:204-212).

All randomness comes from numpy's PCG64 `Generator`, whose streams are stable across numpy versions,
so the same (seed, shape) arguments give bit-identical inputs in the golden-fixture generator
(tests/golden/make_golden.py, run where /root/reference exists) and in the GPU tests.
"""
import math

import numpy as np
import torch

from . import layout as L


def make_rollout(seed, T, game_id='synthetic', team_id=2, player_id=0, weight_version=1,
                 reward_scale=0.05, with_canvas=False, forbid_enum=()):
    """One hero's rollout message of T env-steps (dict, reference wire format)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    obs = {'env': rng.standard_normal((T, L.ENV_FEATS), dtype=np.float32)}
    for key, cnt in L.UNIT_COUNTS.items():
        obs[key] = (0.5 * rng.standard_normal((T, cnt, L.UNIT_FEATS), dtype=np.float32)).astype(np.float32)

    masks = {k: np.zeros((T, c), dtype=np.uint8) for k, c in L.HEAD_COUNTS.items()}
    actions = {k: np.zeros((T, c), dtype=np.uint8) for k, c in L.HEAD_COUNTS.items()}

    def pick(mask_row):
        idx = np.flatnonzero(mask_row)
        return int(idx[rng.integers(0, len(idx))])

    for t in range(T):
        m = (rng.random(4) < 0.75).astype(np.uint8)
        m[0] = 1                                  # no-op is always possible (policy.py:236-237)
        for f in forbid_enum:
            m[f] = 0
        masks['enum'][t] = m
        e = pick(m)
        actions['enum'][t, e] = 1
        sub = {1: ('x', 'y'), 2: ('target_unit',), 3: ('ability',)}.get(e, ())
        for k in sub:
            c = L.HEAD_COUNTS[k]
            mk = (rng.random(c) < 0.6).astype(np.uint8)
            if k == 'target_unit':
                mk[0] = 0                          # own hero is never targetable (policy.py:255)
            a = int(rng.integers(1 if k == 'target_unit' else 0, c))
            mk[a] = 1
            masks[k][t] = mk
            actions[k][t, a] = 1

    rewards = (reward_scale * rng.standard_normal((T, L.N_REWARDS), dtype=np.float32)).astype(np.float32)
    data = {
        'game_id': game_id, 'team_id': team_id, 'player_id': player_id,
        'weight_version': weight_version,
        'canvas': np.zeros((256, 256, 3), dtype=np.uint8) if with_canvas else None,
        'observations': {k: torch.from_numpy(v) for k, v in obs.items()},
        'masks': {k: torch.from_numpy(v) for k, v in masks.items()},
        'actions': {k: torch.from_numpy(v) for k, v in actions.items()},
        'rewards': rewards,
    }
    return data


def make_rollouts(seed, lengths, **kw):
    return [make_rollout(seed * 100003 + i, int(T), game_id='synthetic-%d' % i, **kw)
            for i, T in enumerate(lengths)]


def init_state_dict(seed=7, cell='gru', hidden=256, layers=1):
    """Random-init weights with the reference's names/shapes (OrderedDict name -> float32 tensor).

    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for the Linear layers and U(-1/sqrt(H), 1/sqrt(H)) for the
    recurrent cell: the distribution torch's default initialisers draw from, produced by our own
    numpy stream so that the reference (golden script), the oracle and the HIP path can all be
    loaded with bit-identical weights without shipping them.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shp in L.param_shapes(cell, hidden, layers).items():
        if name.startswith('rnn.'):
            bound = 1.0 / math.sqrt(hidden)
        elif name.endswith('.weight'):
            bound = 1.0 / math.sqrt(shp[1])
            last_fan_in = shp[1]
        else:
            bound = 1.0 / math.sqrt(last_fan_in)
        w = rng.uniform(-bound, bound, size=shp).astype(np.float32)
        sd[name] = torch.from_numpy(w)
    return sd


def flatten_rollout(data):
    """Wire-format dict -> (obs float32[T,483], act uint8[T,65], mask uint8[T,65], rew float32[T,10])
    numpy arrays in the flattened per-step layout of layout.py."""
    T = data['rewards'].shape[0]
    obs = np.empty((T, L.OBS_DIM), dtype=np.float32)
    obs[:, :L.ENV_FEATS] = np.asarray(data['observations']['env'])
    o = L.ENV_FEATS
    for key, cnt in L.UNIT_COUNTS.items():
        w = cnt * L.UNIT_FEATS
        obs[:, o:o + w] = np.asarray(data['observations'][key]).reshape(T, w)
        o += w
    act = np.concatenate([np.asarray(data['actions'][k]).astype(np.uint8) for k in L.OUTPUT_KEYS], axis=1)
    msk = np.concatenate([np.asarray(data['masks'][k]).astype(np.uint8) for k in L.OUTPUT_KEYS], axis=1)
    rew = np.ascontiguousarray(data['rewards'], dtype=np.float32)
    return obs, act, msk, rew
