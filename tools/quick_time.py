"""Scratch timing of one optimizer iteration (not part of the bench contract)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import synth
from dotaclient_amd.engine import Engine, pack_rollouts

cell, hidden, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device('cuda:0')
eng = Engine(cell, hidden, 1, dev)
eng.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
t0 = time.time()
rollouts = synth.make_rollouts(1, [S] * B)
print('gen %.2fs' % (time.time() - t0))
batch = pack_rollouts(rollouts, S, dev)
def it():
    chunks = eng.rollout_pass(batch, S)
    for ep in range(4):
        eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
it(); torch.cuda.synchronize()
for name, fn in [('rollout', lambda: eng.rollout_pass(batch, S)),
                 ('epoch', lambda: eng.train_epoch(eng.rollout_pass(batch, S), 5e-5, 5e-4, 0.5))]:
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); print(name, '%.2f ms' % ((time.time() - t0) * 1e3))
torch.cuda.synchronize(); t0 = time.time()
for _ in range(iters): it()
torch.cuda.synchronize(); dt = (time.time() - t0) / iters
print('%s-%d B=%d S=%d: %.2f ms/iter  %.0f env-steps/s' % (cell, hidden, B, S, dt * 1e3, B * S / dt))
print(eng.out.cpu().numpy())
