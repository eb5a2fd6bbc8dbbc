"""`Policy` - the nn.Module surface of /root/reference/policy.py:36-178, backed by the HIP engine.

What is kept (wire format / call surface the actors and the optimizer consumer rely on):
  * attribute names and `state_dict()` keys/shapes/dtypes (published every iteration,
    optimizer.py:706-716, loaded strict=True by the actors, agent.py:186,315);
  * `forward(env, allied_heroes, ..., hidden) -> (dict of 5 head logits, value, hidden)`
    (policy.py:92-167), `sequence` (:86-90), `single` (:80-84), `init_hidden` (:77-78),
    `masked_softmax` (:169-178), the class constants.
What changes: the parameters are views into ONE flat fp32 device buffer owned by the engine
(`Policy.engine.params`), and forward runs the hand-written HIP kernels; it is inference-only (no
autograd graph) - training goes through `DotaOptimizer.train`, which runs the fused
forward/loss/backward/Adam path on the same buffer.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import layout as L
from .engine import CELL_ID, DcDims, Engine, PackedBatch, device_empty, device_zeros


class _Affine(nn.Module):
    """Parameter holder with nn.Linear's attribute names (weight, bias)."""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)
        self.bias = nn.Parameter(bias, requires_grad=False)
        self.out_features, self.in_features = weight.shape


class _Rnn(nn.Module):
    def __init__(self, tensors):
        super().__init__()
        for k, v in tensors.items():
            setattr(self, k, nn.Parameter(v, requires_grad=False))


class Policy(nn.Module):
    TICKS_PER_OBSERVATION = 15
    TICKS_PER_SECOND = 30
    MAX_MOVE_SPEED = 550
    MAX_MOVE_IN_OBS = (MAX_MOVE_SPEED / TICKS_PER_SECOND) * TICKS_PER_OBSERVATION
    N_MOVE_ENUMS = 9
    MOVE_ENUMS = (np.arange(N_MOVE_ENUMS, dtype=np.float32) - int(N_MOVE_ENUMS / 2)) * (MAX_MOVE_IN_OBS / (N_MOVE_ENUMS - 1) * 2)
    OBSERVATIONS_PER_SECOND = TICKS_PER_SECOND / TICKS_PER_OBSERVATION
    MAX_UNITS = L.MAX_UNITS
    ACTION_OUTPUT_COUNTS = dict(L.HEAD_COUNTS)
    OUTPUT_KEYS = ACTION_OUTPUT_COUNTS.keys()
    INPUT_KEYS = list(L.INPUT_KEYS)

    def __init__(self, cell='gru', hidden=256, layers=1, device=None):
        super().__init__()
        self.engine = Engine(cell, hidden, layers, device)
        self.cell, self.hidden_size, self.layers = cell, hidden, layers
        e = self.engine
        lin = ['affine_env', 'affine_unit_basic_stats'] + ['affine_unit_' + s for s in L.UNIT_SUFFIX.values()] + \
              ['affine_pre_rnn', 'affine_head_enum', 'affine_move_x', 'affine_move_y', 'affine_unit_attention',
               'affine_head_ability', 'affine_value']
        # registration order = the reference's named_parameters() order (policy.py:54-75)
        order = list(L.param_shapes(cell, hidden, layers).keys())
        mods = {n: _Affine(e.param_view(n + '.weight'), e.param_view(n + '.bias')) for n in lin}
        rnn = _Rnn({k[4:]: e.param_view(k) for k in order if k.startswith('rnn.')})
        seen = set()
        for key in order:
            top = key.split('.')[0]
            if top in seen:
                continue
            seen.add(top)
            setattr(self, top, rnn if top == 'rnn' else mods[top])
        self._init_like_torch()

    def _init_like_torch(self):
        """Default init with the distributions nn.Linear / nn.GRU use (policy.py:54-75)."""
        from .synth import init_state_dict
        seed = int(torch.initial_seed() % (2 ** 31))
        self.engine.load_state_dict(init_state_dict(seed, self.cell, self.hidden_size, self.layers))

    # ---- torch plumbing --------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        own = L.param_shapes(self.cell, self.hidden_size, self.layers)
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError('Error(s) in loading state_dict for Policy: missing %s unexpected %s' % (missing, unexpected))
        for k in own:
            if k in state_dict:
                if tuple(state_dict[k].shape) != tuple(own[k]):
                    raise RuntimeError('size mismatch for %s' % k)
                self.engine.param_view(k).copy_(state_dict[k].to(self.engine.device, torch.float32))
        self.engine.params_changed()
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def attach_grads(self):
        """Expose the flat gradient buffer as `.grad` views (what optimizer.py:691-695 reads)."""
        for name, p in self.named_parameters():
            p.grad = self.engine.param_view(name, self.engine.grads)

    # ---- reference call surface ------------------------------------------------------------------
    def init_hidden(self):                                             # policy.py:77-78
        z = torch.zeros([self.layers, 1, self.hidden_size], dtype=torch.float32, device=self.engine.device)
        return z if self.cell == 'gru' else (z, z.clone())

    def single(self, hidden, **kwargs):                                # policy.py:80-84
        """One env-step of one hero (what the rollout actor calls every 0.5 s of game time, agent.py:652).
        `single_kernel` (default on): the whole step is ONE kernel (csrc/policy_single.hip, dc_policy_single) - the 483-float observation
        row is assembled in a pinned host row the kernel reads directly (the actor's tensors are CPU tensors, agent.py:640-650), the
        hidden state is read where the caller holds it, the results land in fresh tensors: one launch, no copies; exact f32 arithmetic.
        Off: the batch path's ~14 launches on a padded tile, replayed as one hipGraph over static buffers (`single_graph`, the path of
        rounds 4-5) or launched eagerly."""
        if self.single_kernel:
            return self._single_fused(hidden, kwargs)
        if self.single_graph:
            return self._single_graphed(hidden, kwargs)
        return self.__call__(**{k: v.unsqueeze(0).unsqueeze(0) for k, v in kwargs.items()}, hidden=hidden)

    single_kernel = True
    single_graph = True

    @staticmethod
    def _single_result(out, hid):
        ho, tu = out[:L.HEADOUT_LD].view(1, 1, -1), out[L.HEADOUT_LD:].view(1, 1, -1)
        logits = {'enum': ho[..., L.HEADOUT_ENUM:L.HEADOUT_ENUM + 4], 'x': ho[..., L.HEADOUT_X:L.HEADOUT_X + 9],
                  'y': ho[..., L.HEADOUT_Y:L.HEADOUT_Y + 9], 'target_unit': tu,
                  'ability': ho[..., L.HEADOUT_ABILITY:L.HEADOUT_ABILITY + 3]}
        return logits, ho[..., L.HEADOUT_VALUE:L.HEADOUT_VALUE + 1], hid

    @torch.no_grad()
    def _single_fused(self, hidden, kw):
        e, dev = self.engine, self.engine.device
        st = getattr(self, '_fused_state', None)
        if st is None:
            st = self._fused_state = {
                'rows': torch.empty(2, L.OBS_DIM, dtype=torch.float32).pin_memory(), 'busy': [None, None], 'turn': 0,
                # zero once: afterwards the kernel keeps its granules and its launch generation there
                'scratch': device_zeros(_lib.DC_SINGLE_SCRATCH_FLOATS, torch.float32, dev),
                'dims': DcDims(CELL_ID[self.cell], self.hidden_size, self.layers, 1, 1, 0, 1)}
        vals = [kw[k] for k in L.INPUT_KEYS]
        if all(not v.is_cuda for v in vals):
            # two pinned rows in turn; a row is rewritten only after the launch that read it has finished (a caller that does not read a
            # result between two calls would otherwise race the kernel: ADVICE r5)
            i = st['turn']
            st['turn'] = i ^ 1
            if st['busy'][i] is not None:
                st['busy'][i].synchronize()
            obs = st['rows'][i]
            torch.cat([v.reshape(-1) for v in vals], out=obs)
        else:                                             # observation tensors already on the device: assembled there
            i = None
            obs = torch.cat([v.reshape(-1).to(dev, torch.float32) for v in vals])
        lstm = self.cell == 'lstm'
        fit = lambda t: t if (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) else t.to(dev, torch.float32).contiguous()
        h0 = fit(hidden[0] if lstm else hidden)
        c0 = fit(hidden[1]) if lstm else None
        if h0.numel() != self.layers * self.hidden_size:
            raise ValueError('Policy.single: hidden must be (layers, 1, hidden)')
        out = device_empty((L.HEADOUT_LD + L.MAX_UNITS,), torch.float32, dev)     # (engine.DEVICE_ALLOC_HOOK: the guard allocator of tools/guard_soak.py)
        hT = device_empty((self.layers, 1, self.hidden_size), torch.float32, dev)
        cT = device_empty((self.layers, 1, self.hidden_size), torch.float32, dev) if lstm else None
        _lib.check(e.lib.dc_policy_single(ctypes.byref(st['dims']), _lib.ptr(e.params), e.poff, _lib.ptr(obs), _lib.ptr(h0), _lib.ptr(c0),
                                          _lib.ptr(out), _lib.ptr(hT), _lib.ptr(cT), _lib.ptr(st['scratch']), _lib.stream_ptr()),
                   'dc_policy_single')
        if i is not None:
            if st['busy'][i] is None:
                st['busy'][i] = torch.cuda.Event()
            st['busy'][i].record()
        return self._single_result(out, (hT, cT) if lstm else hT)

    @torch.no_grad()
    def _single_graphed(self, hidden, kw):
        e, dev = self.engine, self.engine.device
        st = getattr(self, '_single_state', None)
        if st is None:
            st = self._single_state = {
                'obs_host': torch.empty(1, L.OBS_DIM, dtype=torch.float32).pin_memory(),
                'obs': torch.empty(1, L.OBS_DIM, dtype=torch.float32, device=dev),
                'h0': torch.zeros(self.layers, 1, self.hidden_size, dtype=torch.float32, device=dev),
                'c0': torch.zeros(self.layers, 1, self.hidden_size, dtype=torch.float32, device=dev) if self.cell == 'lstm' else None,
                'hT': torch.zeros(self.layers, 1, self.hidden_size, dtype=torch.float32, device=dev),
                'cT': torch.zeros(self.layers, 1, self.hidden_size, dtype=torch.float32, device=dev) if self.cell == 'lstm' else None,
                'out': torch.zeros(L.HEADOUT_LD + L.MAX_UNITS, dtype=torch.float32, device=dev),
                'off': torch.zeros(1, dtype=torch.int64, device=dev), 'len': torch.ones(1, dtype=torch.int32, device=dev),
                'graph': None, 'key': None, 'calls': 0}
            st['batch'] = PackedBatch(st['obs'], None, None, None, st['off'], st['len'], 1)
        # the observation row on the host (the actor's tensors are CPU tensors, agent.py:640-650), one pinned H2D copy
        if all(not kw[k].is_cuda for k in L.INPUT_KEYS):
            # (the previous call's asynchronous H2D copy may still be reading the pinned row if the caller did not read a result in
            #  between: wait for it before the host writes the row again - ADVICE r5)
            if st.get('h2d_done') is not None:
                st['h2d_done'].synchronize()
            row, o = st['obs_host'][0], 0
            for k in L.INPUT_KEYS:
                v = kw[k].reshape(-1)
                row[o:o + v.numel()].copy_(v)
                o += v.numel()
            st['obs'].copy_(st['obs_host'], non_blocking=True)
            if st.get('h2d_done') is None:
                st['h2d_done'] = torch.cuda.Event()
            st['h2d_done'].record()
        else:                                             # observation tensors already on the device: assembled there
            st['obs'].copy_(torch.cat([kw[k].reshape(1, -1).to(dev, torch.float32) for k in L.INPUT_KEYS], dim=1))
        if self.cell == 'gru':
            st['h0'].copy_(hidden)
        else:
            st['h0'].copy_(hidden[0]); st['c0'].copy_(hidden[1])

        def run():
            d, _, _ = e.forward(st['batch'], st['h0'], st['c0'], want_final=True, hT_out=st['hT'], cT_out=st['cT'])
            st['out'][:L.HEADOUT_LD].copy_(e.ws_view(d, 'HEADOUT')[:L.HEADOUT_LD])
            st['out'][L.HEADOUT_LD:].copy_(e.ws_view(d, 'TU')[:L.MAX_UNITS])

        key = (0 if e._ws is None else e._ws.data_ptr(), e.params.data_ptr(), e.kernel_flags, e.products)
        if st['graph'] is None or st['key'] != key:
            run()                                        # eager: the first call also allocates the workspace and sets kernel attributes
            st['calls'] += 1
            key = (e._ws.data_ptr(), e.params.data_ptr(), e.kernel_flags, e.products)
            if st['calls'] >= 2 or st['key'] is not None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run()
                st['graph'] = g
            st['key'] = key
        else:
            st['graph'].replay()
        hid = st['hT'].clone() if self.cell == 'gru' else (st['hT'].clone(), st['cT'].clone())
        return self._single_result(st['out'].clone(), hid)

    def sequence(self, hidden, **kwargs):                              # policy.py:86-90
        return self.__call__(**{k: v.unsqueeze(0) for k, v in kwargs.items()}, hidden=hidden)

    @torch.no_grad()
    def forward(self, env, allied_heroes, enemy_heroes, allied_nonheroes, enemy_nonheroes, allied_towers,
                enemy_towers, hidden):
        """Input as batch (b, s, ...), policy.py:92-167."""
        dev = self.engine.device
        b, s = env.shape[0], env.shape[1]
        parts = [env.reshape(b * s, -1)] + [t.reshape(b * s, -1) for t in
                                            (allied_heroes, enemy_heroes, allied_nonheroes, enemy_nonheroes,
                                             allied_towers, enemy_towers)]
        obs = torch.cat([p.to(dev, torch.float32) for p in parts], dim=1).contiguous()
        batch = PackedBatch(obs, None, None, None, torch.arange(b, device=dev, dtype=torch.int64) * s,
                            torch.full((b,), s, device=dev, dtype=torch.int32), s)
        if self.cell == 'gru':
            h0, c0 = hidden.to(dev).contiguous(), None
        else:
            h0, c0 = hidden[0].to(dev).contiguous(), hidden[1].to(dev).contiguous()
        d, hT, cT = self.engine.forward(batch, h0, c0, want_final=True)
        ho = self.engine.ws_view(d, 'HEADOUT')[:b * s * L.HEADOUT_LD].view(b, s, L.HEADOUT_LD)
        tu = self.engine.ws_view(d, 'TU')[:b * s * L.MAX_UNITS].view(b, s, L.MAX_UNITS)
        logits = {
            'enum': ho[..., L.HEADOUT_ENUM:L.HEADOUT_ENUM + 4].clone(),
            'x': ho[..., L.HEADOUT_X:L.HEADOUT_X + 9].clone(),
            'y': ho[..., L.HEADOUT_Y:L.HEADOUT_Y + 9].clone(),
            'target_unit': tu.clone(),
            'ability': ho[..., L.HEADOUT_ABILITY:L.HEADOUT_ABILITY + 3].clone(),
        }
        value = ho[..., L.HEADOUT_VALUE:L.HEADOUT_VALUE + 1].clone()
        return logits, value, (hT if self.cell == 'gru' else (hT, cT))

    @classmethod
    def masked_softmax(cls, logits, mask, dim=2):
        """Returns log-probabilities (policy.py:169-178); plain tensor plumbing for API users -
        the optimizer's own masked log-softmax runs inside the fused HIP loss kernel."""
        e = torch.exp(logits)
        e = torch.where(mask.bool(), e, torch.zeros_like(e))
        return logits - torch.log(e.sum(dim, keepdim=True))
