"""Region breakdown of the reference's default shape (GRU-256, seq_len 16, ragged rollouts until >= 1024 chunks): bench.py's
`reference_defaults_gru256_s16_ragged` side measurement with per-launch HIP events.  Usage: python tools/ragged_profile.py [cell] [hidden]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cell = sys.argv[1] if len(sys.argv) > 1 else 'gru'
hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.Generator(np.random.PCG64(99))
lens, chunks = [], 0
while chunks < 1024:
    t = int(rng.integers(100, 900))
    lens.append(t)
    chunks += (t + 15) // 16
dev = torch.device('cuda:0')
r = bench.run_workload(cell, hidden, 1, len(lens), 16, 4, 5, 2, dev, 0, 1, lengths=lens, want_profile=True)
print('%d rollouts, %d chunks, %.3f ms per step, %.0f env-steps/s' % (len(lens), chunks, r['elapsed'] / 5 * 1e3, chunks * 16 * 5 / r['elapsed']))
for reg in sorted(r['regions'], key=lambda x: -x['total_ms'])[:14]:
    print('%-32s n=%5d avg=%8.1f us  %7.3f ms' % (reg['kernel'], reg['launches'], reg['total_ms'] / max(reg['launches'], 1) * 1e3, reg['total_ms']))
