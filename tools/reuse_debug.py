#!/usr/bin/env python
"""Two engines in lockstep from the same state, one with reuse_rollout_forward, on one workload: reports per iteration / epoch the
relative difference of every parameter's gradient (first divergence = the culprit)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dotaclient_amd import layout as L, synth            # noqa: E402
from dotaclient_amd.engine import Engine, pack_rollouts  # noqa: E402

cell, hidden, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dev = torch.device('cuda:0')
A, Bn = Engine(cell, hidden, 1, dev), Engine(cell, hidden, 1, dev)
A.reuse_rollout_forward = True
sd = synth.init_state_dict(7, cell, hidden, 1)
A.load_state_dict(sd); Bn.load_state_dict(sd)
ro = synth.make_rollouts(1000, [S] * B)
ba, bb = pack_rollouts(ro, S, dev), pack_rollouts(ro, S, dev)
names = list(L.param_shapes(cell, hidden, 1).keys())
for it in range(iters):
    # same starting state for both
    for k in ('params', 'adam_m', 'adam_v', 'seg_step'):
        getattr(Bn, k).copy_(getattr(A, k))
    Bn.params_changed(); A.params_changed()
    ca, cb = A.rollout_pass(ba, S), Bn.rollout_pass(bb, S)
    for ep in range(4):
        A.train_epoch(ca, 5e-5, 5e-4, 0.5)
        Bn.train_epoch(cb, 5e-5, 5e-4, 0.5)
        torch.cuda.synchronize()
        worst = []
        for n in names:
            ga, gb = A.param_view(n, A.grads), Bn.param_view(n, Bn.grads)
            d = float((ga - gb).abs().max() / (gb.abs().max() + 1e-30))
            if d > 1e-4:
                worst.append((n, d))
        oa, ob = A.out[:11].cpu().numpy(), Bn.out[:11].cpu().numpy()
        print('it %d ep %d  norms %.6g %.6g  loss %.8g %.8g  divergent grads: %s' % (it, ep, oa[9], ob[9], oa[0], ob[0], worst[:6]))
