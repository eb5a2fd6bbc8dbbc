"""Where does an element-wise error of ~1e-3 on `values` come from?  Runs the ORACLE network (oracle/ref_policy.py,
the pinned restatement of policy.py:92-167) on the bench's synthetic trajectories twice - in fp32 (what `parity` compares against) and in
fp64 - and prints, with the same yardsticks as tests/util.py, how far the fp32 REFERENCE ITSELF is from the exact result, layer by layer.
CPU only; run where torch is (no GPU, no reference needed):  python tools/values_error_budget.py [cell hidden B S]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from dotaclient_amd import synth            # noqa: E402
from oracle import ref_optimizer as RO      # noqa: E402
from tests import util                      # noqa: E402


def main():
    cell, hidden, B, S = (sys.argv[1:5] + ['lstm', 256, 64, 256][len(sys.argv) - 1:])[:4]
    hidden, B, S = int(hidden), int(B), int(S)
    torch.set_num_threads(16)
    sd = synth.init_state_dict(7, cell, hidden, 1)
    rollouts = synth.make_rollouts(1000, [S] * B)
    obs = {k: torch.stack([torch.as_tensor(np.asarray(r['observations'][k])) for r in rollouts]) for k in rollouts[0]['observations']}
    outs = {}
    for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
        pol = RO.make_policy(sd, cell, hidden, 1).to(dt)
        taps = {}
        pol.affine_pre_rnn.register_forward_hook(lambda m, i, o: taps.__setitem__('pre_rnn', o.detach()))
        pol.rnn.register_forward_hook(lambda m, i, o: taps.__setitem__('rnn_out', o[0].detach()))
        with torch.no_grad():
            logits, value, _ = pol({k: v.to(dt) for k, v in obs.items()}, tuple(h.to(dt) for h in pol.init_hidden(B)) if cell == 'lstm'
                                   else pol.init_hidden(B).to(dt))
        outs[name] = {'pre_rnn': taps['pre_rnn'].double().numpy(), 'rnn_out': taps['rnn_out'].double().numpy(),
                      'values': value.double().numpy().ravel(), 'enum_logits': logits['enum'].double().numpy()}
    rep = {'workload': '%s-%d %dx%d, oracle fp32 vs oracle fp64 (same weights, same inputs)' % (cell, hidden, B, S)}
    for k in outs['f32']:
        a, b = outs['f32'][k], outs['f64'][k]
        ew, frac = util.elementwise_rel_err(a, b)
        rep[k] = {'scaled_max_abs_over_max': util.scaled_err(a, b), 'elementwise_rel_err_above_1e-3_of_max': ew, 'entries_above_floor': frac}
    a, b = outs['f32']['values'], outs['f64']['values']
    keep = np.abs(b) > 1e-3 * np.abs(b).max()
    i = int(np.argmax(np.where(keep, np.abs(a - b) / np.maximum(np.abs(b), 1e-300), 0)))
    rep['values_worst_entry'] = {'ref': float(b[i]), 'abs_err': float(abs(a[i] - b[i])), 'max_abs_ref': float(np.abs(b).max()),
                                 'abs_err_over_max': float(abs(a[i] - b[i]) / np.abs(b).max())}
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    main()
