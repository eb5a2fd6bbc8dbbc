"""Where the model publish (optimizer.py:697-716) spends its host time, stage by stage (`flat_snapshot` 9.9 ms vs
2.5 ms for the reference's per-tensor form).  Medians over 30 publishes, each with a fresh Adam-sized kernel in front (so that the
D2H copy has something to wait for).  Usage: python tools/publish_probe.py [out.json]"""
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd import synth                      # noqa: E402
from dotaclient_amd.engine import Engine              # noqa: E402

dev = torch.device('cuda:0')
eng = Engine('lstm', 256, 1, dev)
eng.load_state_dict(synth.init_state_dict(7, 'lstm', 256, 1))
T = {}


def stamp(name, t0):
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)


for it in range(34):
    eng.params.mul_(1.0)                       # device work in front of the snapshot
    torch.cuda.synchronize()
    t0 = time.perf_counter(); i = eng.start_param_snapshot(); stamp('1 start_param_snapshot (enqueue)', t0)
    t0 = time.perf_counter(); eng._snap['done'][i].synchronize(); stamp('2 wait for the D2H copy', t0)
    t0 = time.perf_counter(); views = eng.snapshot_state_dict(i, clone=False); stamp('3 34 views of the pinned buffer', t0)
    t0 = time.perf_counter(); clones = {k: v.clone() for k, v in views.items()}; stamp('4 34 clones', t0)
    t0 = time.perf_counter(); b = io.BytesIO(); torch.save(views, b); stamp('5 torch.save(views)', t0)
    n_views = len(b.getvalue())
    t0 = time.perf_counter(); b = io.BytesIO(); torch.save(clones, b); stamp('6 torch.save(clones)', t0)
    n_clones = len(b.getvalue())
    t0 = time.perf_counter(); sd = {k: v.cpu() for k, v in eng.state_dict().items()}; stamp('7 reference form: 34 x .cpu()', t0)
    t0 = time.perf_counter(); b = io.BytesIO(); torch.save(sd, b); stamp('8 reference form: torch.save', t0)
    t0 = time.perf_counter()
    i = eng.start_param_snapshot(); b = io.BytesIO(); torch.save(eng.snapshot_state_dict(i, clone=False), b)
    stamp('9 publish as DotaOptimizer.upload_model does it (snapshot + save of views)', t0)
    t0 = time.perf_counter()
    i = eng.start_param_snapshot(); b = io.BytesIO(); torch.save(eng.snapshot_state_dict(i), b)
    stamp('10 round-3 form (snapshot + 34 clones + save)', t0)
out = {k: round(float(np.median(v[4:])), 3) for k, v in T.items()}
out['bytes_views'], out['bytes_clones'] = n_views, n_clones
# the blob of views must load into the reference's wire format
sd2 = torch.load(io.BytesIO(b.getvalue()))
assert list(sd2.keys()) == list(eng.state_dict().keys())
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(out, f, indent=1)
