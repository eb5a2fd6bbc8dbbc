#!/usr/bin/env python
"""Which buffer is read before it is written?  Poisons ONE region (a workspace sub-buffer or an output buffer of the passes) with
0xFF bytes (NaN as f32, 255 as u8, -1 as an index) before the first pass of a workload and runs two iterations; a region whose
poison changes the result (NaN status, non-finite outputs, a fault) is read uninitialised by some kernel.
  python tools/uninit_probe.py <workload>            every region, one subprocess each
  python tools/uninit_probe.py <workload> <region>   one region in this process"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tools.guard_soak import WORKLOADS, lengths_of       # noqa: E402

BUFS = ['old_logp', 'values', 'argmax', 'adv', 'ret', 'h0', 'c0']


def regions(layers):
    from dotaclient_amd import engine as E
    r = ['ws:' + n for n in E.WS_FIXED if n != 'FAULT']
    for l in range(layers):
        r += ['ws%d:%s' % (l, n) for n in E.WS_LAYER]
    return r + ['buf:' + b for b in BUFS]


def one(name, region):
    import torch
    from dotaclient_amd import synth
    from dotaclient_amd.engine import Engine, pack_rollouts
    cell, hidden, layers, spec, S, flags, reuse = WORKLOADS[name]
    dev = torch.device('cuda:0')
    eng = Engine(cell, hidden, layers, dev)
    eng.kernel_flags = flags
    eng.reuse_rollout_forward = reuse
    eng.load_state_dict(synth.init_state_dict(7, cell, hidden, layers))
    batch = pack_rollouts(synth.make_rollouts(1000, lengths_of(spec, S)), S, dev)
    d = eng.dims(batch, True)
    eng._workspace(d)
    eng._ws.zero_()                                   # everything else starts from zeros: only the chosen region is poisoned
    nbytes = 0
    kind, nm = region.split(':') if ':' in region else ('none', '')
    if kind == 'none':
        pass
    elif kind == 'all':
        eng._ws[256:].fill_(255)
    elif kind.startswith('ws'):
        layer = int(kind[2:]) if len(kind) > 2 else None
        v = eng.ws_view(d, nm, layer, torch.uint8)
        v.fill_(255)
        nbytes = v.numel()
    else:
        B = batch.rows // S
        shape = {'old_logp': (batch.rows, 5), 'values': (batch.rows,), 'argmax': (batch.rows, 5), 'adv': (batch.rows,), 'ret': (batch.rows,),
                 'h0': (layers, B, hidden), 'c0': (layers, B, hidden)}[nm]
        dt = torch.int32 if nm == 'argmax' else torch.float32
        key = nm if nm not in ('h0', 'c0') else '%s_%d' % (nm, S)
        t = eng._buf(batch, key, shape, dt, dev)
        t.view(torch.uint8).fill_(255)
        nbytes = t.numel() * 4
    torch.cuda.synchronize()
    chunks = eng.rollout_pass(batch, S)
    eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)
    torch.cuda.synchronize()
    st = int(eng.status.item())
    o = eng.out[:11].cpu().double().numpy()
    print('RESULT %s %d %d %s %.10g %.10g' % (region, nbytes, st, ' '.join('%.10g' % x for x in o), float(eng.grads.double().abs().sum()),
                                             float(batch.adv.double().abs().sum())))


if __name__ == '__main__':
    name = sys.argv[1]
    if len(sys.argv) > 2:
        one(name, sys.argv[2])
    else:
        import numpy as np
        base = None
        for r in ['none', 'none', 'all:'] + regions(WORKLOADS[name][2]):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), name, r], capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith('RESULT')]
            if p.returncode != 0 or not line:
                print('READ-UNINITIALISED %s: rc %d %s' % (r, p.returncode, ' '.join(p.stderr.strip().splitlines()[-1:])[-150:]), flush=True)
                continue
            f = line[0].split()
            vals = np.array([float(x) for x in f[4:]])
            if base is None:
                base = vals
                continue
            with np.errstate(invalid='ignore', divide='ignore'):
                d = np.abs(vals - base) / (np.abs(base) + 1e-12)
            worst = float(np.nanmax(d)) if np.all(np.isfinite(vals)) else float('inf')
            tag = 'READ-UNINITIALISED' if (worst > 1e-4 or f[3] != '0') else 'clean'
            if tag != 'clean' or r in ('none', 'all:'):
                print('%s %s (%s bytes): status %s, worst relative change vs zero-filled baseline %.3g' % (tag, r, f[2], f[3], worst), flush=True)
        print('probe of %s done' % name, flush=True)
