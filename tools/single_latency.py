"""Where Policy.single's time goes: the one-kernel step (dc_policy_single) timed alone with events, and the host legs around it."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, '.')
from dotaclient_amd import _lib, layout as L, synth
from dotaclient_amd.policy import Policy

for cell, hidden, layers in (('gru', 256, 1), ('lstm', 256, 1), ('lstm', 512, 2)):
    pol = Policy(cell, hidden, layers)
    r = synth.make_rollouts(13, [40])[0]
    hid = pol.init_hidden()
    for t in range(3):
        lg, v, hid = pol.single(**{k: r['observations'][k][t] for k in L.INPUT_KEYS}, hidden=hid)
    st, e = pol._fused_state, pol.engine
    import torch as _t
    st.update(obs=st['rows'][0], h0=_t.zeros(layers, hidden, device='cuda'), c0=_t.zeros(layers, hidden, device='cuda') if cell == 'lstm' else None,
              out=_t.zeros(200, device='cuda'), hT=_t.zeros(layers, hidden, device='cuda'), cT=_t.zeros(layers, hidden, device='cuda') if cell == 'lstm' else None)
    call = lambda: e.lib.dc_policy_single(ctypes.byref(st['dims']), _lib.ptr(e.params), e.poff, _lib.ptr(st['obs']), _lib.ptr(st['h0']),
                                          _lib.ptr(st['c0']), _lib.ptr(st['out']), _lib.ptr(st['hT']), _lib.ptr(st['cT']),
                                          _lib.ptr(st['scratch']), _lib.stream_ptr())
    for _ in range(20):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        call()
    b.record(); torch.cuda.synchronize()
    k_us = a.elapsed_time(b) / 200 * 1e3
    t0 = time.perf_counter()
    for _ in range(200):
        call(); torch.cuda.synchronize()
    sync_us = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    for t in range(200):
        lg, v, hid = pol.single(**{k: r['observations'][k][t % 40] for k in L.INPUT_KEYS}, hidden=hid)
        v.cpu()
    full_us = (time.perf_counter() - t0) / 200 * 1e6
    print('%s-%d x%d: kernel back-to-back %.1f us, launch + sync %.1f us, Policy.single + read %.1f us' % (cell, hidden, layers, k_us, sync_us, full_us))
