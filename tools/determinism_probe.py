#!/usr/bin/env python
"""Repeats forward + loss + backward on FIXED weights and a fixed batch and compares every repetition's outputs with the first one's:
per-parameter gradient differences above `tol` (relative to the parameter's largest gradient entry) are glitches - summation-order
noise of the atomic accumulations is ~1e-6.  Usage: determinism_probe.py cell hidden B S [reps] [kernel_flags]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dotaclient_amd import layout as L, synth            # noqa: E402
from dotaclient_amd.engine import Engine, pack_rollouts  # noqa: E402

cell, hidden, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
tol = 2e-5
dev = torch.device('cuda:0')
eng = Engine(cell, hidden, 1, dev)
eng.kernel_flags = flags
eng.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
batch = pack_rollouts(synth.make_rollouts(1000, [S] * B), S, dev)
chunks = eng.rollout_pass(batch, S)
names = list(L.param_shapes(cell, hidden, 1).keys())
ref = None
glitches = 0
for r in range(reps):
    d, _, _ = eng.forward(chunks, chunks.h0, chunks.c0, lazy_tu=True)
    ho = eng.ws_view(d, 'HEADOUT').clone()
    eng.loss(d, chunks, 0.1, 5e-4, 0.5)
    eng.backward(d, chunks)
    torch.cuda.synchronize()
    cur = {'grads': eng.grads.clone(), 'headout': ho[:chunks.rows * 160], 'out': eng.out[:9].clone()}
    if ref is None:
        ref = cur
        continue
    bad = []
    dh = float((cur['headout'] - ref['headout']).abs().max() / ref['headout'].abs().max())
    if dh > tol:
        bad.append(('FORWARD headout', dh))
    for n in names:
        off, numel, _ = eng.layout[n]
        a, b = cur['grads'][off:off + numel], ref['grads'][off:off + numel]
        dd = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        if dd > tol:
            bad.append((n, float('%.2g' % dd)))
    if bad:
        glitches += 1
        print('rep %d: %s' % (r, bad[:8]), flush=True)
print('%s-%d B=%d S=%d flags=%d: %d / %d repetitions differ from the first by more than %g' % (cell, hidden, B, S, flags, glitches, reps - 1, tol))
