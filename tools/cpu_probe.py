import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dotaclient_amd import synth
from oracle import ref_optimizer as RO
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads())
rollouts = synth.make_rollouts(5, [256] * 8)
for nt in [int(x) for x in sys.argv[1:]]:
    torch.set_num_threads(nt)
    pol = RO.make_policy(synth.init_state_dict(7, 'lstm', 128, 1), 'lstm', 128, 1)
    opt = torch.optim.Adam(pol.parameters(), lr=5e-5)
    t0 = time.time(); chunks = [c for r in rollouts for c in RO.rollout_pass(pol, r, 256)]; t1 = time.time()
    RO.train_step(pol, opt, chunks, 5e-4, 0.5); t2 = time.time()
    print('threads %d: rollout %.2fs epoch %.2fs' % (nt, t1 - t0, t2 - t1), flush=True)
