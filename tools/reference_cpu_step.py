"""Times the REAL reference's own functions (imported from /root/reference: optimizer.py:57-64 advantage_returns, :328-430
experiences_from_rollout, :581-689 train - the stub recipe of tests/golden/make_golden.py) next to the port (oracle/ref_optimizer.py) on the
same synthetic trajectories and host cores.  bench.py's `cpu_baseline` is kind "port" because the reference is Python and cannot travel to
the GPU box, while bench.py cannot run where the reference is (no GPU); this script is the link between the two: run it in the build
container (no GPU needed):   python tools/reference_cpu_step.py [n_trajectories=16] [steps=256] [threads=16] > profiles/r05/reference_vs_port_cpu.json
Only the reference's own cell exists there (GRU-256, policy.py:66)."""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
from dotaclient_amd import synth                 # noqa: E402
from oracle import ref_optimizer as RO           # noqa: E402
import make_golden as MG                         # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    torch.set_num_threads(threads)
    lr, ent, vf, E = 5e-5, 5e-4, 0.5, 4
    rollouts = synth.make_rollouts(1000, [S] * B)
    ref_optimizer, ref_policy = MG.import_reference()

    def reference_step():
        opt = ref_optimizer.DotaOptimizer.__new__(ref_optimizer.DotaOptimizer)
        opt.policy_base = ref_policy.Policy()
        opt.policy_base.load_state_dict(synth.init_state_dict(seed=7), strict=True)
        opt.policy = opt.policy_base
        opt.seq_len, opt.e_clip, opt.entropy_coef, opt.vf_coef = S, 0.1, ent, vf
        opt.optimizer = torch.optim.Adam(opt.policy.parameters(), lr=lr)
        t0 = time.time()
        experiences = []
        with torch.no_grad():
            for r in rollouts:
                experiences.extend(opt.experiences_from_rollout(data=MG.boolify(r)))
        t1 = time.time()
        for _ in range(E):
            losses, _, _ = opt.train(experiences=experiences)
        return t1 - t0, time.time() - t1, float(losses['loss'])

    def port_step():
        pol = RO.make_policy(synth.init_state_dict(7), 'gru', 256, 1)
        opt = torch.optim.Adam(pol.parameters(), lr=lr)
        t0 = time.time()
        chunks = [c for r in rollouts for c in RO.rollout_pass(pol, r, S)]
        t1 = time.time()
        for _ in range(E):
            parts, _, _ = RO.train_step(pol, opt, chunks, ent, vf)
        return t1 - t0, time.time() - t1, float(parts['loss'])

    out = {'workload': "GRU-256 (the reference's cell), %d trajectories x %d steps, rollout pass + %d epochs, torch CPU fp32, %d threads" % (B, S, E, threads),
           'host': 'build container (no GPU); the GPU box runs the port only'}
    for name, fn in (('reference', reference_step), ('port', port_step)):
        fn()                                               # warm-up
        runs = [fn() for _ in range(3)]
        best = min(runs, key=lambda x: x[0] + x[1])
        out[name] = {'rollout_pass_s': round(best[0], 3), 'epochs_s': round(best[1], 3),
                     'env_steps_per_s': round(B * S / (best[0] + best[1]), 1), 'final_loss': best[2]}
    out['port_over_reference_rate'] = round(out['port']['env_steps_per_s'] / out['reference']['env_steps_per_s'], 3)
    out['same_final_loss'] = abs(out['port']['final_loss'] - out['reference']['final_loss']) <= 1e-6 * max(1.0, abs(out['reference']['final_loss']))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
