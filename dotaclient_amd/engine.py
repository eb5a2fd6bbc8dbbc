"""Host-side driver of the HIP library: owns the flat parameter/gradient/Adam buffers, the workspace
and the packed batches, and sequences the C-ABI calls.  No arithmetic happens here.

Mirrors what the reference does implicitly through torch objects:
  flat params + views      <->  Policy's nn.Parameters            (policy.py:54-75)
  PackedBatch               <->  the stacked tensors of train()    (optimizer.py:587-615)
  Engine.rollout_pass       <->  experiences_from_rollout          (optimizer.py:328-430)
  Engine.train_epoch        <->  train                             (optimizer.py:581-689)
"""
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib
from . import layout as L

CELL_ID = {'gru': 0, 'lstm': 1}
MAX_GRAD_NORM = 0.5        # optimizer.py:204
ADAM_BETAS = (0.9, 0.999)  # torch.optim.Adam defaults used at optimizer.py:275
ADAM_EPS = 1e-8


class DcDims(ctypes.Structure):
    _fields_ = [('cell', ctypes.c_int32), ('hidden', ctypes.c_int32), ('layers', ctypes.c_int32),
                ('n_seq', ctypes.c_int32), ('max_len', ctypes.c_int32), ('flags', ctypes.c_int32),
                ('rows', ctypes.c_int64)]


DC_DIMS_LAZY_TU = 1     # include/dotaclient_hip.h
DC_DIMS_BWD_UPPER = 2   # dc_policy_backward: heads .. pre-rnn projection only (zeroes the gradient buffer first)
DC_DIMS_BWD_EMBED = 4   # dc_policy_backward: the embedding parameters only (after UPPER)
# kernel-selection overrides (A/B measurements, tests that pit one kernel family against another): Engine.kernel_flags
DC_DIMS_DENSE_POOL_BWD = 8
DC_DIMS_RNN_PER_STEP = 16
DC_DIMS_LSTM_MFMA = 32
DC_DIMS_LSTM_VALU = 64
DC_DIMS_TEAM_DEVICE_SCOPE = 128
DC_DIMS_TEAM_NS = lambda n: n << 8
DC_DIMS_GEMM_FASTTILE = 2048
DC_DIMS_BF16 = 4096
DC_DIMS_GEMM_X3_ALL = 8192
DC_DIMS_TEAM_VALU = 16384
DC_DIMS_EMBED_UNFUSED = 32768
DC_DIMS_RNN_STEP_BF16 = 65536
DC_DIMS_POOL16_8W = 262144
DC_DIMS_TEAM8 = 524288      # H = 256 recurrent core in teams of eight workgroups (csrc/rnn_team8.hip): forced for any number of sequences
DC_DIMS_TEAM4 = 1048576     # ... never (by itself the library takes them for 65 .. 128 sequences)
DC_DIMS_POOL16_VALU = 2097152   # keep the sparse VALU max-pool backward in f16x2 mode too (default there: csrc/embed_pool16m.hip)
DC_DIMS_BF16_F32_STORE = 4194304   # configs[4]: keep the gate buffers f32 (default on that path since round 5: bf16 storage)
DC_DIMS_GEMM_TILE128 = 8388608   # keep x W^T / dy W on the 128 x 128 split-on-load kernel (default since round 6: csrc/gemm_x3s.hip's row-streaming kernel)
DC_DIMS_FWD_ONLY = 16777216      # dc_policy_forward: no backward will follow this pass (the no-grad rollout pass)
DC_DIMS_DB2_SCATTER = 134217728   # small types' second-layer bias gradients by embed_scatter_bwd's own pass (A/B)
DC_DIMS_POOL_ENV_SEPARATE = 67108864   # env embedding + five-unit pool as their own launch behind the fused embedding forward (A/B)
DC_DIMS_SMALL_DENSE = 33554432   # keep the small unit types' backward on d(emb) in HBM + the dense kernels (A/B): include/dotaclient_hip.h
DC_DIMS_F16X2 = 131072   # f32-grade products from two f16 pieces (three MFMAs) instead of three bf16 pieces (six): include/dotaclient_hip.h

WS_FIXED = ['FAULT', 'BASIC', 'EMB', 'DEMB', 'XCAT', 'AMAX', 'PRE', 'HEADOUT', 'TU', 'DHEADOUT', 'DTU', 'DPRE', 'DXCAT',
            'STATS', 'WHHT', 'SCRATCH', 'HEADW_PAD', 'TEAM_XBUF', 'WPLANES']
WS_LAYER = ['GATES', 'HN', 'HSEQ', 'HPREV', 'CSEQ', 'CPREV', 'DGX', 'DGH', 'DC', 'DH']


# Every device buffer the C ABI gets to see is allocated here.  DEVICE_ALLOC_HOOK (tests only: tools/guard_soak.py) replaces the
# allocation by one that places the buffer at the edge of its own mapping with unmapped memory beyond, so that a kernel overrun
# faults; torch's own temporaries keep torch's allocator.
DEVICE_ALLOC_HOOK = None


def default_device():
    """The GPU of THIS process (one process per GPU): `LOCAL_RANK` when a launcher set it (torch.distributed.run; optimizer.py:726-734
    reads the same environment), else torch's current device.  What `Engine` / `Policy` / `DotaOptimizer` use for `device=None` and
    where the module-level `discount` / `advantage_returns` put their operands - rank r never lands on GPU 0 by default."""
    lr = os.environ.get('LOCAL_RANK')
    if torch.cuda.is_available():
        if lr is not None and lr.isdigit() and int(lr) < torch.cuda.device_count():
            return torch.device('cuda', int(lr))
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cuda', int(lr) if lr is not None and lr.isdigit() else 0)


def device_empty(shape, dtype, device):
    if DEVICE_ALLOC_HOOK is not None:
        return DEVICE_ALLOC_HOOK(tuple(shape) if not isinstance(shape, int) else (shape,), dtype, torch.device(device))
    return torch.empty(shape, dtype=dtype, device=device)


def device_zeros(shape, dtype, device):
    return device_empty(shape, dtype, device).zero_() if DEVICE_ALLOC_HOOK is not None else torch.zeros(shape, dtype=dtype, device=device)


def device_copy(t, device, non_blocking=False):
    """`t` (host or device tensor) as a device buffer of its own."""
    if DEVICE_ALLOC_HOOK is None:
        return t.to(device, non_blocking=non_blocking) if t.device != torch.device(device) else t
    out = device_empty(t.shape, t.dtype, device)
    out.copy_(t, non_blocking=non_blocking)
    return out


FAULT_KERNELS = {1: 'rnn_team_fwd', 2: 'rnn_team_bwd', 3: 'team_mfma_fwd', 4: 'team_mfma_bwd', 5: 'lstm512_team_fwd', 6: 'lstm512_team_bwd', 7: 'team8_fwd', 8: 'team8_bwd'}
DC_FAULT_TEAM_TIMEOUT = 16


def describe_fault(engine):
    """'' or '; team kernel timeout: <kernel>, layer, team, member, time step, sequence, tag' from the workspace's DC_WS_FAULT record."""
    f = engine.fault()
    if f is None:
        return ''
    return '; team kernel timeout: %s, layer %d, team %d, member %d, time step %d, sequence %d, waited for tag %d' % (
        FAULT_KERNELS.get(f[0] - DC_FAULT_TEAM_TIMEOUT, 'kernel %d' % f[0]), f[1], f[2], f[3], f[4], f[5], f[6])


def describe_status(engine):
    """The engine's status word (0 ok, 1 NaN loss, 2 NaN gradient norm: dc_gradnorm_clip_adam) and, if a team kernel recorded a
    timeout in the workspace's DC_WS_FAULT block, which launch it was - as one line of text."""
    st = int(engine.status.item())
    return {0: '0 (ok)', 1: '1 (NaN loss)', 2: '2 (NaN gradient norm)'}.get(st, str(st)) + describe_fault(engine)


class PackedBatch:
    """Env-steps of several sequences back to back ("packed rows"), all on the GPU."""

    def __init__(self, obs, act, mask, rew, seq_off, seq_len, max_len):
        self.obs, self.act, self.mask, self.rew = obs, act, mask, rew
        self.seq_off, self.seq_len = seq_off, seq_len        # int64 / int32 device tensors
        self.max_len = int(max_len)
        self.rows = int(obs.shape[0])
        self.n_seq = int(seq_off.numel())
        # filled by the rollout pass
        self.old_logp = self.values = self.adv = self.ret = self.argmax = None
        self.h0 = self.c0 = None
        self._chunk_meta = {}
        self._bufs = {}            # output buffers of the passes over THIS batch, allocated once (Engine._buf)
        self.is_first = self.prev_row = None
        self.ready = None          # IncrementalPacker: event behind the H2D copies of this batch (the first pass waits on it)
        self.host_lens = None      # padded rollout lengths as the packer knew them on the host (saves a device read-back)

    def as_chunks(self, seq_len):
        """Same rows viewed as B = rows/seq_len sequences of seq_len steps (rollouts are stored padded to a
        multiple of seq_len, so the chunk batch of train() is this very memory)."""
        assert self.rows % seq_len == 0
        b = self.rows // seq_len
        dev = self.obs.device
        # layout metadata of the chunk view (depends on the batch's shape only): built once per batch
        meta = self._chunk_meta.get(seq_len)
        if meta is None:
            starts = torch.arange(b, device=dev, dtype=torch.int64) * seq_len
            is_first = torch.isin(starts, self.seq_off)
            # row that holds the state a chunk starts from: the last row of the previous chunk of the same rollout, -1
            # (= zeros) for a rollout's first chunk
            meta = {'starts': device_copy(starts, dev), 'lens': device_copy(torch.full((b,), seq_len, device=dev, dtype=torch.int32), dev),
                    'is_first': is_first, 'prev_row': device_copy(torch.where(is_first, torch.full_like(starts, -1), starts - 1), dev)}
            self._chunk_meta[seq_len] = meta
        out = PackedBatch(self.obs, self.act, self.mask, self.rew, meta['starts'], meta['lens'], seq_len)
        out.is_first, out.prev_row = meta['is_first'], meta['prev_row']
        out.old_logp, out.values, out.adv, out.ret, out.argmax = self.old_logp, self.values, self.adv, self.ret, self.argmax
        return out


def _as_np(x):
    """Wire tensors are CPU torch tensors (agent.py:406-416) or numpy arrays: a zero-copy numpy view either way."""
    return x.numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class _StagingSet:
    def __init__(self, capacity, pin):
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=pin)
        self.capacity = capacity
        self.obs = mk((capacity, L.OBS_DIM), torch.float32)
        self.act = mk((capacity, L.ACT_DIM), torch.uint8)
        self.msk = mk((capacity, L.ACT_DIM), torch.uint8)
        self.rew = mk((capacity, 10), torch.float32)
        self.off = mk((4096,), torch.int64)          # sequence offsets / lengths of the batch (IncrementalPacker: no pageable copy,
        self.len = mk((4096,), torch.int32)          # which would block the host until the current stream has drained)
        self.event = None


class StagingPair:
    """Page-locked host buffers for one batch, reused across iterations (a fresh 32 MB allocation costs more in page
    faults than filling it).  ONE double-buffered pair, sized to the largest batch seen so far and grown geometrically
    (the consumer loop adds whole rollouts until min_seq_per_epoch is reached, so the row count changes almost every
    iteration: a pair per exact size would pin host memory without bound); a batch uses the leading `rows` of each
    buffer.  A set is handed out again only after the H2D copies that read it have completed (event recorded behind
    them); the old set is released when a larger one replaces it.

    A pair belongs to whoever packs with it: `pack_rollouts` holds the pair's lock from taking a set to recording the
    event behind its copies, so two threads that share a pair (the process-wide default) are serialised instead of
    filling the same buffers; a caller that wants to pack concurrently brings its own pair (DotaOptimizer's prefetcher)."""

    def __init__(self, pin):
        self.pin = pin
        self.sets = [None, None]
        self.next = 0
        self.lock = threading.Lock()

    def take(self, rows):
        """Call with self.lock held."""
        i = self.next
        self.next = i ^ 1
        st = self.sets[i]
        if st is not None and st.event is not None:
            st.event.synchronize()
            st.event = None
        if st is None or st.capacity < rows:
            cap = rows if st is None else max(rows, st.capacity + st.capacity // 2)
            self.sets[i] = st = None               # release the old pinned buffers before allocating the larger ones
            self.sets[i] = st = _StagingSet(cap, self.pin)
        return st


_DEFAULT_STAGING = {}
_DEFAULT_STAGING_LOCK = threading.Lock()


def default_staging(pin):
    with _DEFAULT_STAGING_LOCK:
        if pin not in _DEFAULT_STAGING:
            _DEFAULT_STAGING[pin] = StagingPair(pin)
        return _DEFAULT_STAGING[pin]


PACK_THREADS = int(os.environ.get('DC_PACK_THREADS', '8'))
_TORCH_OF = {np.float32: torch.float32, np.uint8: torch.uint8}


def pack_rollouts(rollouts, seq_len, device, staging=None):
    """Wire-format rollout dicts (optimizer.py:314-326) -> PackedBatch with one sequence per rollout, each
    zero-padded to a multiple of seq_len (optimizer.py:343-382).

    Replaces the per-key slicing + `.to(device)` of optimizer.py:353-365.  Every key of every
    rollout is copied ONCE, straight into its column block of one page-locked staging buffer per field
    ([rows,483] f32 observations, [rows,65] u8 actions and masks, [rows,10] f32 sub-rewards; reused across
    iterations) - no per-rollout temporaries, no concatenation; the ~17 copies per rollout are one native call for the
    whole batch (`dc_pack_rows`, a few host threads) - followed by one asynchronous H2D copy per field (four per batch
    instead of the reference's 17 per chunk).  `staging`: the StagingPair to pack through (default: the process-wide one)."""
    if len(rollouts) == 0:
        raise ValueError('pack_rollouts: no rollouts')
    lens = [(int(d['rewards'].shape[0]) + seq_len - 1) // seq_len * seq_len for d in rollouts]
    rows = int(sum(lens))
    dev = torch.device(device)
    pin = dev.type == 'cuda'
    pair = staging if staging is not None else default_staging(pin)
    with pair.lock:
        return _pack_locked(rollouts, seq_len, dev, pin, pair.take(rows), lens, rows)


_OBS_B, _ACT_B, _REW_B = 4 * L.OBS_DIM, L.ACT_DIM, 40


def _obs_columns():
    cols = [('env', 0, L.ENV_FEATS)]
    c = L.ENV_FEATS
    for key, cnt in L.UNIT_COUNTS.items():
        cols.append((key, c, cnt * L.UNIT_FEATS))
        c += cnt * L.UNIT_FEATS
    return cols


def _rollout_items(d, r0, lp, st, keep, items):
    """Appends the copy descriptors (src, dst, rows, row_bytes, dst_stride) of ONE rollout whose rows start at r0 of staging set `st`
    and which is stored padded to lp rows."""
    T = int(d['rewards'].shape[0])

    def ptr_of(x, dtype, width):
        if type(x) is torch.Tensor and x.dtype is _TORCH_OF[dtype] and x.numel() == T * width and x.is_contiguous():
            return x.data_ptr()                                   # the usual case (agent.py:406-416): no numpy detour
        a = _as_np(x)
        if a.dtype != dtype or not a.flags['C_CONTIGUOUS']:
            if a.dtype == np.bool_ and dtype == np.uint8:
                a = a.view(np.uint8) if a.flags['C_CONTIGUOUS'] else np.ascontiguousarray(a).view(np.uint8)
            else:
                a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
        if a.size != T * width:
            raise ValueError('pack_rollouts: a %d-step rollout holds an array of %d elements where %d x %d were expected'
                             % (T, a.size, T, width))
        return a.__array_interface__['data'][0]

    obs_p, act_p, msk_p, rew_p = st.obs.data_ptr(), st.act.data_ptr(), st.msk.data_ptr(), st.rew.data_ptr()
    o, acts, msks = d['observations'], d['actions'], d['masks']
    for key, c0, w in _obs_columns():
        items += (ptr_of(o[key], np.float32, w), obs_p + r0 * _OBS_B + 4 * c0, T, 4 * w, _OBS_B)
    for key in L.OUTPUT_KEYS:
        h0, hc = L.HEAD_OFFSETS[key], L.HEAD_COUNTS[key]
        items += (ptr_of(acts[key], np.uint8, hc), act_p + r0 * _ACT_B + h0, T, hc, _ACT_B)
        items += (ptr_of(msks[key], np.uint8, hc), msk_p + r0 * _ACT_B + h0, T, hc, _ACT_B)
    items += (ptr_of(d['rewards'], np.float32, 10), rew_p + r0 * _REW_B, T, _REW_B, _REW_B)
    if lp > T:      # the zero pad of optimizer.py:367-382 (the buffers are reused: clear just these rows)
        n = lp - T
        items += (0, obs_p + (r0 + T) * _OBS_B, n, _OBS_B, _OBS_B)
        items += (0, act_p + (r0 + T) * _ACT_B, n, _ACT_B, _ACT_B)
        items += (0, msk_p + (r0 + T) * _ACT_B, n, _ACT_B, _ACT_B)
        items += (0, rew_p + (r0 + T) * _REW_B, n, _REW_B, _REW_B)


def _run_items(items, threads=None):
    tab = np.ascontiguousarray(np.array(items, dtype=np.int64).reshape(-1, 5).T)          # [5, n_items] (items: flat list)
    col = lambda k: ctypes.c_void_p(tab[k].ctypes.data)
    _lib.check(_lib.load().dc_pack_rows(col(0), col(1), col(2), col(3), col(4), tab.shape[1], PACK_THREADS if threads is None else threads),
               'dc_pack_rows')


def _pack_locked(rollouts, seq_len, dev, pin, st, lens, rows):
    lens_n = np.asarray(lens, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens_n)[:-1]]).astype(np.int64)
    keep, items = [], []      # five numbers per key per rollout, executed by dc_pack_rows in one call
    for i, d in enumerate(rollouts):
        _rollout_items(d, int(off[i]), lens[i], st, keep, items)
    _run_items(items)
    to = lambda x: device_copy(x[:rows], dev, non_blocking=True) if pin else x[:rows].clone()
    batch = PackedBatch(to(st.obs), to(st.act), to(st.msk), to(st.rew), device_copy(torch.from_numpy(off), dev),
                        device_copy(torch.from_numpy(lens_n.astype(np.int32)), dev), int(lens_n.max()))
    batch.host_lens = [int(x) for x in lens]
    if pin:
        st.event = torch.cuda.Event()
        st.event.record()
    return batch


class IncrementalPacker:
    """pack_rollouts one rollout at a time - the consumer loop's form (optimizer.py:448-462 receives the rollouts of a batch one by
    one from the experience queue): `add` copies a rollout into the page-locked staging set THE MOMENT IT ARRIVES (host work that
    hides behind the wait for the next message - and, when the loop prefetches, behind the previous iteration's epochs on the GPU);
    `finish` enqueues the four H2D copies on a side stream and returns the PackedBatch, whose `ready` event the rollout pass waits on.
    The batch is byte-for-byte what pack_rollouts makes of the same rollouts in the same order (tests/test_host_logic.py).

    Owns its StagingPair (two sets: the one being filled and the one whose H2D copies may still be in flight)."""

    def __init__(self, seq_len, device, expected_rows=0):
        self.seq_len, self.dev = int(seq_len), torch.device(device)
        self.pin = self.dev.type == 'cuda'
        self.pair = StagingPair(self.pin)
        self.expected_rows = int(expected_rows)
        self.stream = torch.cuda.Stream(device=self.dev) if self.pin else None
        self._begin()

    def _begin(self):
        self.st = None
        self.rows, self.lens, self.n = 0, [], 0
        self.dst, self.copied = None, 0          # device buffers of the batch being packed / rows of it already on their way
        self._items, self._keep, self.packed = [], [], 0     # copy descriptors of rollouts not yet in the staging set / rows that are

    # The H2D copies go out WHILE the batch is being packed (every COPY_ROWS rows), not behind its last rollout: a 256 x 256 batch is
    # ~140 MB = ~3 ms on the link, which otherwise sits between the host's last memcpy and the rollout pass (tools/ingest_probe.py:
    # 22.3 ms per step with the copies at the end against 19.5 for the step alone - host enqueue 0.9 + packing 17.4 + copy 3.3).
    COPY_ROWS = 4096
    # ... and the host copies of the rollouts into the staging set are done PACK_ROWS rows at a time, not rollout by rollout: a rollout is
    # ~0.5 MB in ~4 600 row pieces - too little for dc_pack_rows to start threads for, and on one thread a 256 x 256 batch takes 17-20 ms,
    # more than its step takes on the GPU (the loop was then bound by the host: 20.4 ms per step against 17.9; now 18.1,
    # profiles/r04/v29_bench_extras.json).  The descriptors are built (and the rollout validated) when it arrives; the bytes move in pieces of ~17 MB on
    # PACK_THREADS threads.
    PACK_ROWS = 8192

    def _pack_pending(self):
        if self._items:
            _run_items(self._items, threads=PACK_THREADS)
            self._items, self._keep = [], []
        self.packed = self.rows

    def _flush(self, upto):
        """Enqueues the copies of staging rows [copied, upto) on the packer's stream."""
        if upto <= self.copied:
            return
        st = self.st
        with torch.cuda.stream(self.stream):
            if self.dst is None or self.dst[0].shape[0] < st.capacity:
                # allocated UNDER the packer's stream (see finish); a staging set that had to grow starts the copies over
                self.dst = [device_empty((st.capacity,) + tuple(x.shape[1:]), x.dtype, self.dev) for x in (st.obs, st.act, st.msk, st.rew)]
                self.copied = 0
            a, b = self.copied, upto
            for t, x in zip(self.dst, (st.obs, st.act, st.msk, st.rew)):
                t[a:b].copy_(x[a:b], non_blocking=True)
            st.event = torch.cuda.Event()            # the set is not handed out again before these copies are done (StagingPair.take),
            st.event.record(self.stream)             # also when the batch is abandoned before finish()
        self.copied = upto

    def __len__(self):
        return self.n

    @property
    def n_seq(self):
        return self.rows // self.seq_len

    def add(self, d):
        lp = (int(d['rewards'].shape[0]) + self.seq_len - 1) // self.seq_len * self.seq_len
        need = self.rows + lp
        if self.st is None:
            with self.pair.lock:
                self.st = self.pair.take(max(need, self.expected_rows))
        elif self.st.capacity < need:                 # rare (a batch larger than any before): move what is packed into a larger set
            self._pack_pending()
            old = self.st
            new = _StagingSet(max(need, old.capacity + old.capacity // 2), self.pin)
            for name in ('obs', 'act', 'msk', 'rew'):
                getattr(new, name)[:self.rows].copy_(getattr(old, name)[:self.rows])
            with self.pair.lock:
                self.pair.sets[self.pair.sets.index(old)] = new
            self.st = new
        # The descriptors of THIS rollout are built into local lists and joined to the pending ones only when the whole rollout has
        # validated: a malformed rollout (ValueError half-way through its keys) leaves nothing queued, so the next good rollout - which
        # targets the same staging rows - cannot race stale copies in dc_pack_rows.
        keep, items = [d], []                         # the rollout's arrays stay alive until their bytes have moved
        _rollout_items(d, self.rows, lp, self.st, keep, items)
        self._keep += keep
        self._items += items
        self.rows = need
        self.lens.append(lp)
        self.n += 1
        self.expected_rows = max(self.expected_rows, need)
        if need - self.packed >= self.PACK_ROWS:
            self._pack_pending()
            if self.pin and self.packed - self.copied >= self.COPY_ROWS:
                self._flush(self.packed)

    def finish(self):
        if self.n == 0:
            raise ValueError('IncrementalPacker.finish: no rollouts')
        self._pack_pending()
        lens_n = np.asarray(self.lens, dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(lens_n)[:-1]]).astype(np.int64)
        st, rows, dev = self.st, self.rows, self.dev
        if not self.pin:
            batch = PackedBatch(st.obs[:rows].clone(), st.act[:rows].clone(), st.msk[:rows].clone(), st.rew[:rows].clone(),
                                torch.from_numpy(off), torch.from_numpy(lens_n.astype(np.int32)), int(lens_n.max()))
            batch.host_lens = [int(x) for x in self.lens]
            self._begin()
            return batch
        cur = torch.cuda.current_stream(dev)
        # The copies run on the packer's stream, which the copy engine serves while the current stream's kernels keep computing -
        # so nothing here may wait for the current stream.  The destination buffers are therefore allocated UNDER the packer's stream
        # (the caching allocator only hands that stream blocks whose earlier uses it has seen complete) and marked as also used
        # by the current stream, where the passes read them (record_stream: not recycled before that work is done).  Most rows are
        # on their way already (add -> _flush); what is left goes now.
        self._flush(rows)
        base = self.dst
        with torch.cuda.stream(self.stream):
            dst = [t[:rows] for t in base]
            n = len(self.lens)
            if n > st.off.numel():
                st.off = torch.empty(2 * n, dtype=torch.int64, pin_memory=True)
                st.len = torch.empty(2 * n, dtype=torch.int32, pin_memory=True)
            st.off[:n] = torch.from_numpy(off)
            st.len[:n] = torch.from_numpy(lens_n.astype(np.int32))
            seq_off = device_empty((n,), torch.int64, dev)
            seq_len = device_empty((n,), torch.int32, dev)
            seq_off.copy_(st.off[:n], non_blocking=True)
            seq_len.copy_(st.len[:n], non_blocking=True)
            dst += [seq_off, seq_len]
            # the chunk view's layout tables (PackedBatch.as_chunks builds them with a handful of torch kernels, one of them a sort, per
            # fresh batch: ~0.3 ms of the consumer loop's iteration - tools/ingest_probe.py B2): the host knows the lengths, so they
            # come along with the batch
            S = self.seq_len
            nb = rows // S
            starts_h = np.arange(nb, dtype=np.int64) * S
            first_h = np.zeros(nb, dtype=np.bool_)
            first_h[off // S] = True                                   # every rollout is stored padded to a multiple of S
            prev_h = np.where(first_h, -1, starts_h - 1)
            if getattr(st, 'm64', None) is None or st.m64.shape[1] < nb:
                st.m64 = torch.empty((2, 2 * nb), dtype=torch.int64, pin_memory=True)
                st.m32 = torch.empty(2 * nb, dtype=torch.int32, pin_memory=True)
                st.mb = torch.empty(2 * nb, dtype=torch.bool, pin_memory=True)
            st.m64[0, :nb] = torch.from_numpy(starts_h)
            st.m64[1, :nb] = torch.from_numpy(prev_h)
            st.m32[:nb] = S
            st.mb[:nb] = torch.from_numpy(first_h)
            meta = {'starts': device_empty((nb,), torch.int64, dev), 'prev_row': device_empty((nb,), torch.int64, dev),
                    'lens': device_empty((nb,), torch.int32, dev), 'is_first': device_empty((nb,), torch.bool, dev)}
            meta['starts'].copy_(st.m64[0, :nb], non_blocking=True)
            meta['prev_row'].copy_(st.m64[1, :nb], non_blocking=True)
            meta['lens'].copy_(st.m32[:nb], non_blocking=True)
            meta['is_first'].copy_(st.mb[:nb], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        for t in base + [seq_off, seq_len] + list(meta.values()):
            t.record_stream(cur)
        st.event = ready
        batch = PackedBatch(dst[0], dst[1], dst[2], dst[3], seq_off, seq_len, int(lens_n.max()))
        batch._chunk_meta[S] = meta
        batch.ready = ready
        batch.host_lens = [int(x) for x in self.lens]
        self._begin()
        return batch


class Engine:
    def __init__(self, cell='gru', hidden=256, layers=1, device=None):
        self.lib = _lib.load()
        self.cell, self.hidden, self.layers = cell, int(hidden), int(layers)
        self.device = default_device() if device is None else torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.DotaHipError('the PPO hot path only runs on the GPU (no CPU fallback)')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        # One process per GPU: the library launches on the calling thread's current HIP device and on torch's current stream, so the
        # engine's device is made the current one (the reference's ranks are CPU processes).
        if torch.cuda.is_available() and torch.cuda.current_device() != self.device.index:
            torch.cuda.set_device(self.device)
        self.layout, self.total = L.flat_layout(cell, hidden, layers)
        z = lambda: device_zeros(self.total, torch.float32, self.device)
        self.params, self.grads, self.adam_m, self.adam_v = z(), z(), z(), z()
        # offsets in dc_param_index order
        names = ['affine_env.weight', 'affine_env.bias', 'affine_unit_basic_stats.weight',
                 'affine_unit_basic_stats.bias', 'affine_unit_ah.weight', 'affine_unit_ah.bias',
                 'affine_pre_rnn.weight', 'affine_pre_rnn.bias', 'affine_unit_attention.weight',
                 'affine_unit_attention.bias']
        for l in range(layers):
            names += ['rnn.weight_ih_l%d' % l, 'rnn.weight_hh_l%d' % l, 'rnn.bias_ih_l%d' % l, 'rnn.bias_hh_l%d' % l]
        self.poff = (ctypes.c_int64 * len(names))(*[self.layout[n][0] for n in names])
        # Adam segments = the named parameters (optimizer.py:275 iterates policy.parameters())
        seg_names = list(L.param_shapes(cell, hidden, layers).keys())
        gate = []
        for n in seg_names:
            g = -1
            if n.startswith('affine_head_enum'): g = 0
            elif n.startswith('affine_move_x'): g = 1
            elif n.startswith('affine_move_y'): g = 2
            elif n.startswith('affine_unit_attention') or n.startswith('affine_unit_eth'): g = 3
            elif n.startswith('affine_head_ability'): g = 4
            elif n.startswith('affine_value'): g = 5
            gate.append(g)
        self.seg_names = seg_names
        self.seg_gate_host = gate
        dev = self.device
        self.seg_off = device_copy(torch.tensor([self.layout[n][0] for n in seg_names], dtype=torch.int64), dev)
        self.seg_len = device_copy(torch.tensor([self.layout[n][1] for n in seg_names], dtype=torch.int32), dev)
        self.seg_gate = device_copy(torch.tensor(gate, dtype=torch.int32), dev)
        self.max_seg_len = max(self.layout[n][1] for n in seg_names)
        self.seg_step = device_zeros(len(seg_names), torch.int32, dev)
        self.segsq = device_zeros(len(seg_names) * (1 + (self.max_seg_len + 4095) // 4096), torch.float64, dev)   # totals + per-chunk partial sums
        self.out = device_zeros(16, torch.float32, dev)     # 0..8 losses/entropies, 9..10 norms
        self.ctl = device_zeros(4, torch.float32, dev)     # clip coefficient, ok flag, arrival counter (u32), release generation (u32)
        self.head_on = device_zeros(8, torch.int32, dev)
        self.status = device_zeros(1, torch.int32, dev)
        # flat offset of the first parameter that is not part of the unit / env embeddings (they come first in the layout)
        self.embed_floats = self.layout['affine_pre_rnn.weight'][0]
        self._ws = None
        self._ws_off = None
        self._ws_dims_key = None
        self.kernel_flags = 0          # DC_DIMS_* overrides OR-ed into every call's dims (0 = the library picks by shape)
        # f32-grade products: 'f16x2' (default) = two f16 pieces per operand, three MFMAs, fixed power-of-two pre-scales (DC_DIMS_F16X2);
        # 'bf16x3' = three bf16 pieces, six MFMAs (f32's exponent range).  An operand outside f16's range (activation > 4094, gradient
        # entry > ~16384 / rows) turns the f16x2 loss NaN, the update is skipped on the device (sticky status word) and the caller
        # falls back: use_safe_products() and a repeat of the iteration (DotaOptimizer does that by itself).
        self.products = os.environ.get('DC_PRODUCTS', 'f16x2')
        self.use_graphs = os.environ.get('DC_EPOCH_GRAPH', '0') == '1'   # default of train_epoch(graph=None)
        # The first epoch of an iteration runs the policy on the weights the rollout pass has just used (optimizer.py:328-430 then
        # :581-689): the same function of the same inputs, whose activations are still in the workspace.  True: that epoch's forward
        # is skipped (results identical up to the kernels' summation orders).  Off by default: bench.py's headline counts the
        # reference's five passes per step; the saving is reported beside it.
        self.reuse_rollout_forward = False
        self._param_version = 0        # bumped whenever the flat parameter buffer changes
        self._ws_holds = None          # (rows, obs pointer, parameter version) of the forward whose activations the workspace holds
        self._graphs = {}

    # ---- parameters ------------------------------------------------------------------------------
    def param_view(self, name, buf=None):
        off, numel, shape = self.layout[name]
        return (self.params if buf is None else buf)[off:off + numel].view(shape)

    def load_state_dict(self, sd):
        for n in self.layout:
            self.param_view(n).copy_(sd[n].to(self.device, torch.float32))
        self.params_changed()

    def params_changed(self):
        """To be called by whoever writes the flat parameter buffer behind the engine's back (Policy.load_state_dict through the
        parameter views, the data-parallel broadcast): activations held in the workspace no longer belong to the weights."""
        self._param_version += 1
        self._ws_holds = None

    def state_dict(self):
        return {n: self.param_view(n).detach().clone() for n in L.param_shapes(self.cell, self.hidden, self.layers)}

    # ---- model publish (optimizer.py:697-716: state_dict -> torch.save -> model exchange, every iteration) ----------
    def start_param_snapshot(self):
        """Enqueues ONE device -> host copy of the flat parameter buffer into a page-locked buffer on a side stream,
        ordered after everything already enqueued on the current stream (the last optimizer step) and returning at
        once: the copy runs beside whatever the caller enqueues next.  The reference's publish does 34 synchronous
        per-tensor `.cpu()` copies instead.  Double-buffered: a snapshot stays valid until the one after the next starts."""
        if not hasattr(self, '_snap'):
            self._snap = {'host': [torch.empty(self.total, dtype=torch.float32).pin_memory() for _ in range(2)],
                          'stream': torch.cuda.Stream(device=self.device), 'done': [None, None], 'cur': 1}
        sn = self._snap
        sn['cur'] ^= 1
        i = sn['cur']
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(sn['stream']):
            sn['stream'].wait_event(ready)
            sn['host'][i].copy_(self.params, non_blocking=True)
            done = torch.cuda.Event()
            done.record(sn['stream'])
        sn['done'][i] = done
        return i

    def snapshot_state_dict(self, which=None, clone=True):
        """The last (or the given) snapshot as the reference's wire format: the 34 names / shapes / dtypes of
        `Policy.state_dict()` (optimizer.py:706, loaded with strict=True by the actors, agent.py:186,315), as host
        tensors that own their memory (the snapshot buffer is reused two publishes later; clone=False: views of that buffer, for a
        caller that serialises them at once).  Blocks only on that snapshot's copy."""
        sn = getattr(self, '_snap', None)
        if sn is None or sn['done'][sn['cur'] if which is None else which] is None:
            raise _lib.DotaHipError('snapshot_state_dict: no snapshot started')
        i = sn['cur'] if which is None else which
        sn['done'][i].synchronize()
        host = sn['host'][i]
        out = {}
        for n in L.param_shapes(self.cell, self.hidden, self.layers):
            off, numel, shape = self.layout[n]
            t = host[off:off + numel].view(shape)
            out[n] = t.clone() if clone else t
        return out

    # ---- workspace -------------------------------------------------------------------------------
    def dims(self, batch, lazy_tu=False):
        """lazy_tu: DC_DIMS_LAZY_TU - the target-unit logits are produced by select_logp / loss for the unmasked
        units only (the optimizer's passes); False gives DC_WS_TU for every unit (Policy.forward)."""
        return DcDims(CELL_ID[self.cell], self.hidden, self.layers, batch.n_seq, batch.max_len,
                      (DC_DIMS_LAZY_TU if lazy_tu else 0) | self.kernel_flags | (DC_DIMS_F16X2 if self.products == 'f16x2' else 0), batch.rows)

    def use_safe_products(self):
        """Switches to the three-bf16-piece products (f32's exponent range) for good and clears the sticky status word: the caller repeats
        what the NaN guard stopped (no parameter was touched).  Returns False if the engine was on them already."""
        if self.products != 'f16x2':
            return False
        self.products = 'bf16x3'
        self.status.zero_()
        self._ws_holds = None
        self._graphs.clear()
        return True

    def use_safe_recurrent(self):
        """If a team kernel recorded a timeout (DC_WS_FAULT: its workgroups wait for each other and assume a compute unit each - a CU
        mask, a partitioned device or a co-tenant breaks that, and the launch poisons its outputs with NaN instead of hanging), switches
        the recurrent core to the launch-per-step kernels for good (DC_DIMS_RNN_PER_STEP; LSTM-512 in bf16 mode:
        DC_DIMS_RNN_STEP_BF16 - no workgroup waits for another), clears the record and the status word.  Returns True if it did."""
        flag = DC_DIMS_RNN_STEP_BF16 if (self.kernel_flags & DC_DIMS_BF16) and self.hidden == 512 else DC_DIMS_RNN_PER_STEP
        if (self.kernel_flags & flag) or self.fault() is None:
            return False
        self.kernel_flags |= flag
        self.clear_fault()
        self.status.zero_()
        self._ws_holds = None
        self._graphs.clear()
        return True

    def recover_from_nan(self, fault_on_any_rank=None):
        """What the consumer loop tries ONCE when an iteration turned NaN (no parameter was touched: the status word is sticky) before
        raising like the reference: a recorded team-kernel timeout -> the launch-per-step recurrent kernels; otherwise f16x2 products
        -> the bf16x3 products.  Returns a line for the log, or '' when there is nothing left to try.

        fault_on_any_rank (data parallel): whether ANY rank recorded a team-kernel timeout - every rank then takes the same branch (the
        fault record is rank-local; ranks on different kernels / products would still be correct, but not reproducible)."""
        what = describe_fault(self)
        # ONE decision variable: the local record in a single process, the all-reduced flag under data parallelism - every rank then takes
        # the same branch below (a stale record on one rank could send it down another path than its peers)
        local = self.fault() is not None
        fault = local if fault_on_any_rank is None else bool(fault_on_any_rank)
        if fault:
            flag = DC_DIMS_RNN_STEP_BF16 if (self.kernel_flags & DC_DIMS_BF16) and self.hidden == 512 else DC_DIMS_RNN_PER_STEP
            if local:
                self.clear_fault()               # consumed: a record left behind must not steer a later recovery
            if not (self.kernel_flags & flag):
                self.kernel_flags |= flag
                self.status.zero_()
                self._ws_holds = None
                self._graphs.clear()
                return 'launch-per-step recurrent kernels from here on (' + (what.lstrip('; ') if local else 'a team kernel timed out on another rank') + ')'
            # (already on the launch-per-step kernels: the record was stale - go on to the products, like every other rank)
        if self.use_safe_products():
            return 'bf16x3 products (f32 exponent range) - an operand may have left f16\'s range'
        return ''

    def use_fast_products(self):
        """Back to the default two-f16-piece products (the consumer loop probes them again some iterations after a fallback: one
        out-of-range batch should not cost the fast path for the rest of the run).  Returns False if they were on already."""
        if self.products == 'f16x2':
            return False
        self.products = 'f16x2'
        self._ws_holds = None
        self._graphs.clear()
        return True

    def _workspace(self, d):
        n = len(WS_FIXED) + len(WS_LAYER) * self.layers
        offs = (ctypes.c_int64 * n)()
        total = self.lib.dc_workspace_layout(ctypes.byref(d), offs)
        key = (d.rows, d.n_seq)
        if self._ws is None or self._ws.numel() < total:
            self._ws = None
            self._ws = device_empty(int(total), torch.uint8, self.device)
            self._ws[:256].zero_()          # DC_WS_FAULT (offset 0 for every dims): the owner clears it once, the library only sets it
            self._ws_holds = None
        self._ws_off = list(offs)
        self._ws_dims_key = key
        return self._ws

    def fault(self):
        """The sticky fault record of the H = 256 team kernels (include/dotaclient_hip.h, DC_WS_FAULT) as a list of 8 ints, or
        None when nothing was recorded (synchronises)."""
        if self._ws is None:
            return None
        f = self._ws[:32].view(torch.int32).cpu().tolist()
        return f if f[0] != 0 else None

    def clear_fault(self):
        if self._ws is not None:
            self._ws[:256].zero_()

    def ws_view(self, d, name, layer=None, dtype=torch.float32):
        """Tensor view of a workspace buffer (tests / plumbing)."""
        n = len(WS_FIXED) + len(WS_LAYER) * self.layers
        offs = (ctypes.c_int64 * n)()
        total = self.lib.dc_workspace_layout(ctypes.byref(d), offs)
        idx = WS_FIXED.index(name) if layer is None else len(WS_FIXED) + layer * len(WS_LAYER) + WS_LAYER.index(name)
        nxt = offs[idx + 1] if idx + 1 < n else total
        return self._ws[offs[idx]:nxt].view(dtype)

    # ---- C-ABI calls -----------------------------------------------------------------------------
    def forward(self, batch, h0=None, c0=None, want_final=False, lazy_tu=False, hT_out=None, cT_out=None, fwd_only=False):
        """fwd_only: DC_DIMS_FWD_ONLY - no backward will read this pass (the no-grad rollout pass): kernels skip what only a backward needs."""
        d = self.dims(batch, lazy_tu)
        if fwd_only:
            d = DcDims(d.cell, d.hidden, d.layers, d.n_seq, d.max_len, d.flags | DC_DIMS_FWD_ONLY, d.rows)
        ws = self._workspace(d)
        # what the workspace's activations are a function of (Engine.reuse_rollout_forward compares it; a forward-only pass leaves
        # nothing a backward could start from)
        self._ws_holds = None if fwd_only else (batch.rows, batch.obs.data_ptr(), self._param_version, bool(lazy_tu), self.kernel_flags, self.products)
        hT = cT = None
        if want_final:       # (hT_out / cT_out: caller-owned result buffers - nothing is allocated, e.g. inside a graph capture)
            hT = hT_out if hT_out is not None else device_empty((self.layers, batch.n_seq, self.hidden), torch.float32, self.device)
            cT = None
            if self.cell == 'lstm':
                cT = cT_out if cT_out is not None else device_empty((self.layers, batch.n_seq, self.hidden), torch.float32, self.device)
        _lib.check(self.lib.dc_policy_forward(ctypes.byref(d), _lib.ptr(self.params), self.poff, _lib.ptr(batch.obs),
                                              _lib.ptr(h0), _lib.ptr(c0), _lib.ptr(batch.seq_off), _lib.ptr(batch.seq_len),
                                              _lib.ptr(ws), _lib.ptr(hT), _lib.ptr(cT), _lib.ptr(batch.mask if lazy_tu else None),
                                              _lib.stream_ptr()),
                   'dc_policy_forward')
        return d, hT, cT

    @staticmethod
    def _buf(batch, name, shape, dtype, device):
        """Output buffer `name` of a pass over `batch`: allocated on first use, reused by every later pass over the same
        batch (the epochs of an iteration, the steps of the bench) - no allocation inside the steady-state step."""
        t = batch._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = batch._bufs[name] = device_empty(shape, dtype, device)
        return t

    def select_logp(self, d, batch, want_argmax=True):
        dev = self.device
        logp = self._buf(batch, 'old_logp', (batch.rows, 5), torch.float32, dev)
        values = self._buf(batch, 'values', (batch.rows,), torch.float32, dev)
        argmax = self._buf(batch, 'argmax', (batch.rows, 5), torch.int32, dev) if want_argmax else None
        _lib.check(self.lib.dc_select_logp(ctypes.byref(d), _lib.ptr(self._ws), _lib.ptr(batch.act), _lib.ptr(batch.mask),
                                           _lib.ptr(logp), _lib.ptr(values), _lib.ptr(argmax), _lib.stream_ptr()),
                   'dc_select_logp')
        return logp, values, argmax

    def loss(self, d, batch, e_clip, entropy_coef, vf_coef):
        _lib.check(self.lib.dc_ppo_loss_fwd_bwd(ctypes.byref(d), _lib.ptr(self._ws), _lib.ptr(batch.act), _lib.ptr(batch.mask),
                                                _lib.ptr(batch.old_logp), _lib.ptr(batch.adv), _lib.ptr(batch.ret),
                                                _lib.ptr(self.out), _lib.ptr(self.head_on), float(e_clip),
                                                float(entropy_coef), float(vf_coef), _lib.stream_ptr()),
                   'dc_ppo_loss_fwd_bwd')

    def backward(self, d, batch, part=0):
        """part: 0 = the whole backward; DC_DIMS_BWD_UPPER / DC_DIMS_BWD_EMBED = its two halves as separate calls (the
        gradients of every parameter from affine_pre_rnn on are final after UPPER: a data-parallel caller starts their
        all-reduce there, Engine.train_epoch)."""
        if part:
            d = DcDims(d.cell, d.hidden, d.layers, d.n_seq, d.max_len, d.flags | part, d.rows)
        _lib.check(self.lib.dc_policy_backward(ctypes.byref(d), _lib.ptr(self.params), self.poff, _lib.ptr(self.grads),
                                               self.total, _lib.ptr(batch.obs), _lib.ptr(batch.seq_off),
                                               _lib.ptr(batch.seq_len), _lib.ptr(self._ws), _lib.stream_ptr()),
                   'dc_policy_backward')

    def adam(self, lr, vf_coef):
        _lib.check(self.lib.dc_gradnorm_clip_adam(
            _lib.ptr(self.seg_off), _lib.ptr(self.seg_len), _lib.ptr(self.seg_gate), len(self.seg_names),
            self.max_seg_len, _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v),
            _lib.ptr(self.segsq), _lib.ptr(self.head_on), _lib.ptr(self.out), _lib.ptr(self.out[9:]), _lib.ptr(self.ctl),
            _lib.ptr(self.seg_step), _lib.ptr(self.status), MAX_GRAD_NORM, float(vf_coef), float(lr), ADAM_BETAS[0],
            ADAM_BETAS[1], ADAM_EPS, _lib.stream_ptr()), 'dc_gradnorm_clip_adam')
        self._param_version += 1

    # ---- the two passes of one optimizer iteration -------------------------------------------------
    def rollout_pass(self, batch, seq_len, gamma=0.98, lam=0.97):
        """optimizer.py:328-430 for all rollouts of `batch` at once: no-grad forward with the hidden state
        carried across a rollout's chunks, old log-probs, values, GAE.  Returns the chunk view."""
        from . import ops
        if batch.ready is not None:                   # the batch's H2D copies ran on a side stream
            torch.cuda.current_stream(self.device).wait_event(batch.ready)
            batch.ready = None
        # (with reuse_rollout_forward the first epoch back-propagates THIS pass's activations: then it is not forward-only)
        d, _, _ = self.forward(batch, lazy_tu=True, fwd_only=not self.reuse_rollout_forward)
        batch.old_logp, batch.values, batch.argmax = self.select_logp(d, batch)
        dev = self.device
        batch.adv, batch.ret = ops.gae_scan(batch.rew, batch.values, batch.seq_off, batch.seq_len, batch.max_len, gamma, lam,
                                            adv=self._buf(batch, 'adv', (batch.rows,), torch.float32, dev),
                                            ret=self._buf(batch, 'ret', (batch.rows,), torch.float32, dev))
        chunks = batch.as_chunks(seq_len)
        # initial state of every chunk = state after the previous chunk of the same rollout (detached,
        # optimizer.py:384,408), zeros for a rollout's first chunk (policy.py:77-78): one gather launch per layer and state
        B, H = chunks.n_seq, self.hidden
        chunks.h0 = self._buf(batch, 'h0_%d' % seq_len, (self.layers, B, H), torch.float32, dev)
        chunks.c0 = self._buf(batch, 'c0_%d' % seq_len, (self.layers, B, H), torch.float32, dev) if self.cell == 'lstm' else None
        _lib.check(self.lib.dc_chunk_initial_state(ctypes.byref(d), _lib.ptr(self._ws), _lib.ptr(chunks.prev_row), B,
                                                   _lib.ptr(chunks.h0), _lib.ptr(chunks.c0), _lib.stream_ptr()),
                   'dc_chunk_initial_state')
        return chunks

    def train_epoch(self, chunks, lr, entropy_coef, vf_coef, e_clip=0.1, grad_hook=None, graph=None):
        """optimizer.py:581-689: one full-batch epoch.  Returns the device tensor `out`
        (0 loss, 1 policy, 2 entropy, 3 value, 4..8 entropies, 9 unclipped, 10 clipped) and status.

        graph (default Engine.use_graphs): replay the epoch as ONE hipGraph launch instead of its ~45 kernel launches, memsets
        and copies.  The first epoch over a given chunk batch and hyper-parameters runs eagerly, the second is captured,
        later ones are replays; a data-parallel epoch (grad_hook: a collective in the middle) always runs eagerly."""
        use_graph = self.use_graphs if graph is None else graph
        if use_graph and grad_hook is None:
            return self._train_epoch_graphed(chunks, lr, entropy_coef, vf_coef, e_clip)
        return self._train_epoch_eager(chunks, lr, entropy_coef, vf_coef, e_clip, grad_hook)

    def _train_epoch_eager(self, chunks, lr, entropy_coef, vf_coef, e_clip, grad_hook):
        if (self.reuse_rollout_forward and self._ws is not None
                and self._ws_holds == (chunks.rows, chunks.obs.data_ptr(), self._param_version, True, self.kernel_flags, self.products)):
            # same rows, same weights: the rollout pass's activations are what this forward would write (a chunk's initial state is
            # the state the rollout pass carried into its first row)
            d = self.dims(chunks, True)
            ws_before = self._ws.data_ptr()
            self._workspace(d)
            if self._ws.data_ptr() != ws_before:          # the workspace had to grow: its contents are gone
                d, _, _ = self.forward(chunks, chunks.h0, chunks.c0, lazy_tu=True)
        else:
            d, _, _ = self.forward(chunks, chunks.h0, chunks.c0, lazy_tu=True)
        self._ws_holds = None          # the loss / backward overwrite parts of the saved forward; Adam changes the weights
        self.loss(d, chunks, e_clip, entropy_coef, vf_coef)
        if grad_hook is not None and getattr(grad_hook, 'overlap', False):
            # data-parallel with overlap: the all-reduce of everything but the embedding gradients (82 % of the bucket)
            # runs beside the embedding backward
            self.backward(d, chunks, DC_DIMS_BWD_UPPER)
            grad_hook.start_upper(self)
            self.backward(d, chunks, DC_DIMS_BWD_EMBED)
            grad_hook.finish(self)
        else:
            self.backward(d, chunks)
            if grad_hook is not None:
                grad_hook(self)
        self.adam(lr, vf_coef)
        return self.out, self.status

    def _train_epoch_graphed(self, chunks, lr, entropy_coef, vf_coef, e_clip):
        """Everything an epoch enqueues reads and writes fixed device buffers (the batch, the workspace, the flat
        parameter / gradient / Adam buffers, `out`), and its host-side arguments are part of the key, so the recorded
        launch sequence is valid for every later epoch with the same key."""
        ptr = lambda t: 0 if t is None else t.data_ptr()
        key = (ptr(chunks.obs), ptr(chunks.act), ptr(chunks.mask), ptr(chunks.old_logp), ptr(chunks.adv), ptr(chunks.ret),
               ptr(chunks.h0), ptr(chunks.c0), ptr(chunks.seq_off), ptr(chunks.seq_len), chunks.n_seq, chunks.rows,
               chunks.max_len, float(lr), float(entropy_coef), float(vf_coef), float(e_clip), self.kernel_flags, self.products,
               ptr(self._ws), ptr(self.grads))
        ent = self._graphs.get(key)
        if ent is None:
            # first sight of this key: run eagerly (also takes care of every one-time hipFuncSetAttribute and of the
            # workspace allocation, neither of which belongs inside a capture)
            if len(self._graphs) >= 2:          # a batch is replayed for the epochs of ONE iteration: keep the current and the previous one
                self._graphs.pop(next(iter(self._graphs)))
            self._graphs[key] = {'graph': None, 'keep': (chunks,)}
            out = self._train_epoch_eager(chunks, lr, entropy_coef, vf_coef, e_clip, None)
            if ptr(self._ws) != key[-2]:          # the workspace was (re)allocated by this very call: start over next time
                self._graphs.pop(key, None)
            return out
        if ent['graph'] is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._train_epoch_eager(chunks, lr, entropy_coef, vf_coef, e_clip, None)
            ent['graph'] = g
        ent['graph'].replay()
        self._param_version += 1       # the replay ran Adam
        self._ws_holds = None
        return self.out, self.status
