"""Times dc_gradnorm_clip_adam on the bench model's parameter layout (LSTM-256 x 1): N back-to-back calls between two events."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd.engine import Engine
from dotaclient_amd import synth

dev = torch.device('cuda:0')
eng = Engine('lstm', 256, 1, dev)
eng.load_state_dict(synth.init_state_dict(7, 'lstm', 256, 1))
eng.grads.normal_(0, 1e-3)
eng.head_on.fill_(1)
eng.out.zero_(); eng.out[0] = 0.3
for n in (1, 10, 200):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): eng.adam(1e-4, 0.5)
    e1.record(); torch.cuda.synchronize()
    print('%4d calls: %.2f us per call   status %d  params %d' % (n, 1e3 * e0.elapsed_time(e1) / n, int(eng.status.item()), eng.total))
