"""`DotaOptimizer` - drop-in for the experience-queue consumer of /root/reference/optimizer.py.

Same constructor/CLI arguments (optimizer.py:207-209, 776-794), same method names and return shapes:

    advantage_returns(rewards, values, gamma, lam)        optimizer.py:57-64
    Sequence                                              optimizer.py:176-190
    DotaOptimizer.get_rollout()                           optimizer.py:314-326   (kept: pickle off RMQ)
    DotaOptimizer.experiences_from_rollout(data)          optimizer.py:328-430   -> HIP rollout pass
    DotaOptimizer.train(experiences)                      optimizer.py:581-689   -> HIP fwd/loss/bwd/Adam
    DotaOptimizer.run()                                   optimizer.py:436-579   (consumer loop + metrics)
    DotaOptimizer.upload_model(version)                   optimizer.py:697-723

The RabbitMQ / GCS / TensorBoard / checkpoint-directory plumbing (optimizer.py:67-174, 231-266, 287-308, 534-579) is
outside the hot path and is NOT re-implemented here: the caller injects the reference's own
`MessageQueue` object (`mq=`; INTEGRATION.md, option A keeps the reference's class), and metrics go to an optional
`metrics_sink(metrics, iteration)` callable (the reference's TensorBoard writer, or nothing).  Everything arithmetic
happens in libdotaclient_hip.so; there is no CPU path.
"""
import io
import logging
import os
import pickle
import sys
import time

import numpy as np
import torch

from . import layout as L
from . import ops
from .engine import IncrementalPacker, PackedBatch, default_device, describe_fault, pack_rollouts
from .policy import Policy

logger = logging.getLogger(__name__)
REWARD_KEYS = ['enemy', 'win', 'xp', 'hp', 'kills', 'death', 'lh', 'denies', 'tower_hp', 'mana']  # policy.py:20


def _to_dev_f32(x):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(default_device())


def discount(x, gamma):
    """Same contract as optimizer.py:53-54: float32 (n,) in, float32 (n,) numpy out (float64 accumulate inside, like the
    reference's lfilter).  One wavefront-scan HIP launch."""
    x = np.asarray(x, dtype=np.float32)
    if x.shape[0] == 0:
        return np.zeros(0, np.float32)
    return ops.discount(_to_dev_f32(x), gamma).cpu().numpy()


def advantage_returns(rewards, values, gamma, lam):
    """Same contract as optimizer.py:57-64: rewards/values are float32 (L+1,) arrays - the reference's caller appends a
    terminal 0 to each (optimizer.py:417-420), but like the reference function any terminal reward / bootstrap value
    is accepted; returns (advantages, returns) float32 (L,) numpy.  One wavefront-scan HIP launch."""
    rewards = np.asarray(rewards, dtype=np.float32)
    values = np.asarray(values, dtype=np.float32)
    if rewards.shape != values.shape or rewards.ndim != 1:
        raise ValueError('advantage_returns: rewards and values must be 1-D and of equal length')
    if rewards.shape[0] <= 1:
        return np.zeros(0, np.float32), np.zeros(0, np.float32)
    adv, ret = ops.advantage_returns(_to_dev_f32(rewards), _to_dev_f32(values), gamma, lam)
    return adv.cpu().numpy(), ret.cpu().numpy()


class Sequence:
    """One seq_len chunk of a rollout (optimizer.py:176-190).  Thin handle into a PackedBatch: the
    reference-shaped tensors are materialised lazily (tests / API users); `train` reads the packed
    rows directly."""

    def __init__(self, game_id, weight_version, team_id, batch, index, seq_len, hidden, cell_state=None):
        self.game_id, self.weight_version, self.team_id = game_id, weight_version, team_id
        self._batch, self._index, self._seq_len = batch, index, seq_len
        self.hidden = hidden                      # (layers, 1, H)
        self.cell_state = cell_state

    def _rows(self):
        lo = self._index * self._seq_len
        return slice(lo, lo + self._seq_len)

    def _split(self, flat, counts, shape_tail=None):
        out, o = {}, 0
        for k, c in counts.items():
            out[k] = flat[:, o:o + c]
            o += c
        return out

    @property
    def observations(self):
        obs = self._batch.obs[self._rows()]
        out = {'env': obs[:, :L.ENV_FEATS]}
        o = L.ENV_FEATS
        for k, c in L.UNIT_COUNTS.items():
            out[k] = obs[:, o:o + c * L.UNIT_FEATS].reshape(-1, c, L.UNIT_FEATS)
            o += c * L.UNIT_FEATS
        return out

    @property
    def actions(self):
        return self._split(self._batch.act[self._rows()], L.HEAD_COUNTS)

    @property
    def masks(self):
        return self._split(self._batch.mask[self._rows()], L.HEAD_COUNTS)

    @property
    def rewards(self):
        return self._batch.rew[self._rows()].cpu().numpy()

    @property
    def values(self):
        return self._batch.values[self._rows()].view(1, -1, 1)

    @property
    def advantages(self):
        return self._batch.adv[self._rows()]

    @property
    def returns(self):
        return self._batch.ret[self._rows()]

    @property
    def log_probs_sel(self):
        lp = self._batch.old_logp[self._rows()]
        act = self._batch.act[self._rows()]
        out = {}
        for i, (k, c) in enumerate(L.HEAD_COUNTS.items()):
            sel = act[:, L.HEAD_OFFSETS[k]:L.HEAD_OFFSETS[k] + c].bool().any(dim=1)
            out[k] = lp[sel, i]
        return out


class DotaOptimizer:
    MODEL_FILENAME_FMT = "model_%09d.pt"
    BUCKET_NAME = 'dotaservice'
    MODEL_HISTOGRAM_FREQ = 128
    MAX_GRAD_NORM = 0.5
    SPEED_KEY = 'steps per s'
    MessageQueue = None        # the queue class to construct when no mq= object is passed (default: __main__.MessageQueue, see __init__)

    def __init__(self, rmq_host, rmq_port, epochs, min_seq_per_epoch, seq_len, learning_rate, checkpoint,
                 pretrained_model, mq_prefetch_count, log_dir, entropy_coef, vf_coef, run_local,
                 mq=None, metrics_sink=None, cell='gru', hidden=256, layers=1, device=None, reuse_rollout_forward=False,
                 prefetch=True):
        self.rmq_host, self.rmq_port = rmq_host, rmq_port
        self.epochs, self.min_seq_per_epoch, self.seq_len = epochs, min_seq_per_epoch, seq_len
        self.learning_rate, self.checkpoint = learning_rate, checkpoint
        self.mq_prefetch_count, self.log_dir = mq_prefetch_count, log_dir
        self.entropy_coef, self.vf_coef, self.run_local = entropy_coef, vf_coef, run_local
        if not run_local:
            # optimizer.py:231-266,697-723 with run_local=False resumes from / uploads to a GCS bucket: storage plumbing outside the
            # hot path, not re-implemented - fail loudly instead of silently starting from scratch
            raise ValueError('run_local=False (GCS checkpoint resume / upload) is outside this package: keep the reference\'s own '
                             'checkpoint code around DotaOptimizer, or pass run_local=True')
        if self.checkpoint:
            os.makedirs(self.log_dir, exist_ok=True)                        # optimizer.py:231-233 (events + models live there)
            if pretrained_model is None:
                # The reference resumes by itself here: it picks the newest model_*.pt of log_dir, loads it and carries the version
                # counter on behind it (optimizer.py:243-253).  That directory scan is storage plumbing this package leaves to the
                # integrator's launcher - but starting from random weights at version 1 on top of an existing
                # run would overwrite its model files and publish versions that go BACKWARDS to the actors: refuse.
                import glob
                have = sorted(glob.glob(os.path.join(self.log_dir, 'model_*.pt')))
                if have:
                    raise ValueError('log_dir %r already holds %d checkpoint(s) (newest: %s) and no pretrained_model was given: the reference '
                                     'would resume from the newest one (optimizer.py:243-253); this class does not scan directories - pass '
                                     'pretrained_model=<that file> to resume (the version counter carries on behind its number), or '
                                     'point log_dir at an empty directory to start a new run' % (self.log_dir, len(have), os.path.basename(have[-1])))
        self.iteration_start = 1
        self.iterations = 100000
        self.model_upload_freq = 10
        self.eventfile_refresh_freq = 100
        self.e_clip = 0.1
        self._snapshot = None
        self.bucket = None
        self.policy_base = Policy(cell, hidden, layers, device)
        self.policy = self.policy_base
        self.engine = self.policy_base.engine
        self.device = self.engine.device
        # run()'s first epoch evaluates the policy on the weights experiences_from_rollout has just used (optimizer.py:328-430, then
        # :581-689): the engine back-propagates those activations instead of recomputing the same forward (Engine.reuse_rollout_forward;
        # results equal, one forward pass in five saved).  False restores the reference's pass count.
        self.engine.reuse_rollout_forward = bool(reuse_rollout_forward)
        # Ingest: every rollout is copied into page-locked staging the moment it arrives (IncrementalPacker),
        # and with prefetch=True run_iteration drains what ALREADY WAITS in the experience queue into the next batch's staging while
        # the GPU works through this iteration's epochs (never blocking on an empty queue: the model publish is not delayed).
        self.prefetch = bool(prefetch)
        self._packer = IncrementalPacker(seq_len, self.device, expected_rows=(min_seq_per_epoch + 64) * seq_len)
        # the gather state of the batch being assembled lives next to the packer that holds its rows (a retry after an
        # MQ / unpickling error continues the same batch, the two can not drift apart)
        self._acc = self._new_gather()
        # a batch whose rollout pass was ALREADY enqueued behind the previous iteration's epochs: (acc, experiences, chunks)
        self._ready = None
        self._hist_host = None
        self.pipeline_rollout_pass = True
        # f16x2 -> bf16x3 fallback (Engine.recover_from_nan) is held for `_safe_hold` iterations - the repeated one and the `_safe_hold` - 1
        # after it -, then the fast products are probed again;
        # every further fallback doubles the hold (one rare-head batch must not cost the fast path for the rest of the run,
        # data that overflows every time must not run every iteration twice)
        self._safe_hold, self._safe_left, self._auto_safe = 8, 0, False
        self.prefetch_past_gpu_done = False        # tests: keep draining the queue after the epochs have finished (deterministic batches)

        # An explicitly given pretrained model is loaded (optimizer.py:264-267) and, when checkpointing, the version counter carries on
        # behind its file number (optimizer.py:252-253).  Scanning a checkpoint directory / bucket for the newest model
        # (optimizer.py:243-250,287-296) is storage plumbing outside the hot path: the integrator's launcher
        # resolves the path and passes it here (a log_dir that already holds checkpoints is refused above when it does not).
        if pretrained_model is not None:
            if self.checkpoint:
                self.iteration_start = self.iteration_from_model_filename(filename=pretrained_model) + 1
            self.policy_base.load_state_dict(torch.load(pretrained_model, map_location='cpu'), strict=False)

        # data parallel: flat-bucket RCCL all-reduce instead of the reference's per-parameter gloo wrapper
        self.grad_hook = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size() > 1:
            from .distributed import FlatGradAllReducer
            self.grad_hook = FlatGradAllReducer(self.engine)
            self.grad_hook.sync_parameters()
        self.policy_base.attach_grads()
        self.time_last_it = time.time()

        if mq is None:
            # optimizer.py:278-280 builds its MessageQueue itself.  When this class is swapped into the reference's optimizer.py by
            # import (INTEGRATION.md option A) that module still defines MessageQueue (optimizer.py:67-174) and runs as __main__:
            # construct it exactly like the reference does, so that main() (optimizer.py:751-765) needs no edit.
            mq_cls = self.MessageQueue or getattr(sys.modules.get('__main__'), 'MessageQueue', None)
            if mq_cls is None:
                raise ValueError('DotaOptimizer needs the experience / model queue: pass the reference\'s own MessageQueue(host, port, '
                                 'prefetch_count, use_model_exchange) as mq=, set DotaOptimizer.MessageQueue to that class, or run from '
                                 'the reference\'s optimizer.py, whose MessageQueue is picked up (optimizer.py:67-174; the broker client '
                                 'is outside the hot path and is not re-implemented, INTEGRATION.md)')
            mq = mq_cls(host=self.rmq_host, port=self.rmq_port, prefetch_count=mq_prefetch_count, use_model_exchange=self.checkpoint)
        self.mq = mq
        self.metrics_sink = metrics_sink
        self.mq.connect()
        self.upload_model(version=self.iteration_start)

    @staticmethod
    def iteration_from_model_filename(filename):
        """optimizer.py:298-308: 'model_000000123.pt' -> 123 (1 if the name carries no number)."""
        import re
        x = re.search(r'(\d+)(?=\.pt)', os.path.basename(filename))
        return int(x.group(0)) if x else 1

    # ---- experience ingest ---------------------------------------------------------------------------
    def get_rollout(self):
        """optimizer.py:314-326."""
        method, properties, body = self.mq.consume_xp()
        data = pickle.loads(body)
        subrewards = data['rewards'].sum(axis=0)
        return data, subrewards, data['rewards'].shape[0], data['weight_version'], data['canvas']

    def experiences_from_rollouts(self, rollouts):
        """Batched form of optimizer.py:328-430: all rollouts go through ONE rollout pass (the hidden
        state only couples chunks within a rollout).  Returns (list[Sequence], chunk PackedBatch)."""
        return self._experiences_from_batch(rollouts, pack_rollouts(rollouts, self.seq_len, self.device))

    def _experiences_from_batch(self, rollouts, batch):
        chunks = self.engine.rollout_pass(batch, self.seq_len, gamma=0.98, lam=0.97)
        seqs = []
        lens = batch.host_lens if batch.host_lens is not None else batch.seq_len.cpu().tolist()
        i = 0
        for data, Lr in zip(rollouts, lens):
            for _ in range(Lr // self.seq_len):
                h = chunks.h0[:, i:i + 1]
                c = chunks.c0[:, i:i + 1] if chunks.c0 is not None else None
                seqs.append(Sequence(data['game_id'], data['weight_version'], data['team_id'], chunks, i,
                                     self.seq_len, h, c))
                i += 1
        return seqs, chunks

    def experiences_from_rollout(self, data):
        """optimizer.py:328-430 (one rollout)."""
        return self.experiences_from_rollouts([data])[0]

    def _gather(self, experiences):
        """Packed chunk batch for a list of Sequences.  Sequences produced by one
        experiences_from_rollouts call already share their batch; otherwise the rows are concatenated
        on the device (plumbing, cached per experience list)."""
        first = experiences[0]._batch
        if all(e._batch is first for e in experiences) and len(experiences) == first.n_seq and \
                all(e._index == i for i, e in enumerate(experiences)):
            return first
        # cache of the last gathered list, keyed on its CONTENT (which chunk of which batch, in order); the key's batches
        # are kept alive with it, so an id() can never be recycled while the entry exists
        key = tuple((id(e._batch), e._index) for e in experiences)
        if getattr(self, '_gather_key', None) == key:
            return self._gather_val
        S = self.seq_len
        cat = lambda name: torch.cat([getattr(e._batch, name)[e._rows()] for e in experiences])
        dev = self.device
        b = len(experiences)
        out = PackedBatch(cat('obs'), cat('act'), cat('mask'), cat('rew'),
                          torch.arange(b, device=dev, dtype=torch.int64) * S,
                          torch.full((b,), S, device=dev, dtype=torch.int32), S)
        out.old_logp, out.values, out.adv, out.ret = cat('old_logp'), cat('values'), cat('adv'), cat('ret')
        out.h0 = torch.cat([e.hidden for e in experiences], dim=1).contiguous()       # optimizer.py:591
        out.c0 = torch.cat([e.cell_state for e in experiences], dim=1).contiguous() \
            if experiences[0].cell_state is not None else None
        self._gather_key, self._gather_val = key, out
        self._gather_refs = list({id(e._batch): e._batch for e in experiences}.values())
        return out

    # ---- the optimizer step ----------------------------------------------------------------------------
    def train(self, experiences):
        """optimizer.py:581-689: one epoch on the full batch; returns (losses, entropies, grad_norms) as
        dicts of 0-d tensors with the reference's keys (optimizer.py:682-689).

        NaN: like the reference a ValueError, parameters untouched.  Before raising, the epoch is repeated ONCE on the safe kernels /
        products (Engine.recover_from_nan).  That repeat can only help when the experiences themselves are sound: if the overflow or
        timeout happened in `experiences_from_rollout`, their old log-probs / values / advantages are NaN already and the caller has to
        recompute them (run() / run_iteration do: they repeat the rollout pass as well)."""
        chunks = self._gather(experiences)
        out, status = self.engine.train_epoch(chunks, self.learning_rate, self.entropy_coef, self.vf_coef,
                                              e_clip=self.e_clip, grad_hook=self.grad_hook)
        host = torch.cat([out[:11], status.to(torch.float32)]).cpu()      # the one sync of the epoch
        st = int(host[11].item())
        how = self.engine.recover_from_nan() if st != 0 else ''
        if how:
            # a NaN that may be the kernels' doing (a team-kernel timeout; an operand outside the f16 pieces' range): repeat the epoch
            # once on the safe path (nothing was updated; the experiences' old log-probs / values are what experiences_from_rollout computed)
            logger.warning('train: NaN - repeating the epoch with the %s', how)
            out, status = self.engine.train_epoch(chunks, self.learning_rate, self.entropy_coef, self.vf_coef,
                                                  e_clip=self.e_clip, grad_hook=self.grad_hook)
            host = torch.cat([out[:11], status.to(torch.float32)]).cpu()
            st = int(host[11].item())
        if st != 0:
            msg = describe_fault(self.engine) + ('; already repeated with the ' + how if how else '')
            self.engine.status.zero_()                                      # sticky on the device (csrc/adam.hip): the caller clears it
        if st == 1:                                                         # optimizer.py:667-669
            raise ValueError('loss={}, policy_loss={}, entropy_loss={}, value_loss={}'.format(*host[:4].tolist()) + msg)
        if st == 2:                                                         # optimizer.py:678-679
            raise ValueError('grad_norm={}'.format(host[9].item()) + msg)
        losses = {'loss': host[0], 'policy_loss': host[1], 'entropy_loss': host[2], 'value_loss': host[3]}
        entropies = {k: host[4 + i] for i, k in enumerate(L.OUTPUT_KEYS)}
        return losses, entropies, {'unclipped': host[9], 'clipped': host[10]}

    def mean_gradient_norm(self):
        """optimizer.py:691-695 (metric helper on the exposed .grad views; torch plumbing)."""
        return torch.stack([p.grad.norm(2) for p in self.policy_base.parameters() if p.grad is not None]).mean()

    @staticmethod
    def list_of_dicts_to_dict_of_lists(x):
        return {k: torch.stack([d[k] for d in x]) for k in x[0]}

    @staticmethod
    def _new_gather():
        return {'rollouts': [], 'subrewards': [], 'rollout_lens': [], 'weight_versions': [], 'xp_waits': 0.0, 'hidden_s': 0.0,
                'n_prefetched': 0, 'pipelined': 0}

    def _consume_one(self, acc):
        """One message off the experience queue (optimizer.py:452-455), packed into the staging set at once.  The packer's rows and
        `acc` change together, after everything that can raise."""
        t0 = time.time()
        rollout, rollout_subrewards, rollout_len, weight_version, canvas = self.get_rollout()
        acc['xp_waits'] += time.time() - t0
        self._packer.add(rollout)                                           # raises before it commits its rows
        acc['rollouts'].append(rollout)
        acc['subrewards'].append(rollout_subrewards)
        acc['rollout_lens'].append(rollout_len)
        acc['weight_versions'].append(weight_version)

    def _finish_batch(self):
        """The gathered rollouts -> H2D (packer's stream) -> rollout pass enqueued.  Returns (acc, experiences, chunks)."""
        acc = self._acc
        batch = self._packer.finish()                                       # the rollout pass waits on the batch's `ready` event
        self._acc = self._new_gather()
        assert len(acc['rollouts']) == len(batch.host_lens), 'gather state and packed batch disagree'
        acc['batch'] = batch                                                # kept for a repeat of the iteration (_retry_with_safe_products)
        experiences, chunks = self._experiences_from_batch(acc['rollouts'], batch)
        return acc, experiences, chunks

    def _drop_ready(self):
        """An early rollout pass whose iteration will not run (error path): its rollouts go back to the front of the gather state so
        that a caller who catches the error and carries on loses nothing."""
        if self._ready is None:
            return
        acc, _, _ = self._ready
        self._ready = None
        redo, self._acc = self._acc, self._new_gather()
        self._packer._begin()
        for a in (acc, redo):
            for i, r in enumerate(a['rollouts']):
                self._packer.add(r)
                for k in ('rollouts', 'subrewards', 'rollout_lens', 'weight_versions'):
                    self._acc[k].append(a[k][i])
            self._acc['xp_waits'] += a['xp_waits']

    def _queue_has_messages(self):
        """The reference MessageQueue's own `xp_queue_size` (optimizer.py:126-132: a passive queue_declare, None on failure); an
        injected queue object without it simply never prefetches."""
        try:
            return bool(getattr(self.mq, 'xp_queue_size', None))
        except Exception:                                                   # noqa: BLE001 - the reference swallows everything here too
            return False

    def run_iteration(self, it):
        """Body of the reference's run() loop (optimizer.py:437-531); returns the metrics dict.

        Same work in the same order as the reference on the device; what differs is when the HOST waits: the E epochs are enqueued
        back to back (their losses / status land in a device-side history, read with ONE synchronisation at the end - the reference
        synchronises after every epoch through .item()), and while the GPU works the host already drains the experience queue into
        the next batch's staging (`prefetch`).  NaN guards (optimizer.py:667-669,678-679): the optimizer step of a NaN epoch is skipped
        on the device, so raising after the last epoch leaves the same parameters behind as raising in the middle."""
        start_xp = time.time()
        if self._auto_safe and self.engine.products == 'bf16x3':
            self._safe_left -= 1
            if self._safe_left <= 0:        # (a rollout pass already pipelined on bf16x3 is as good: both forms are f32-grade)
                self.engine.use_fast_products()
                self._auto_safe = False
                logger.info('iteration %d: probing the f16x2 products again', it)
        if self._ready is not None:                                         # rollout pass already enqueued during the last iteration
            acc, experiences, chunks = self._ready
            self._ready = None
        else:
            while self._packer.n_seq < self.min_seq_per_epoch:              # optimizer.py:448-462
                self._consume_one(self._acc)
            acc, experiences, chunks = self._finish_batch()
        time_xp = time.time() - start_xp
        subrewards, rollout_lens = acc['subrewards'], acc['rollout_lens']
        weight_ages = [it - v for v in acc['weight_versions']]
        xp_waits = acc['xp_waits']

        start_opt = time.time()
        hist = torch.empty(self.epochs, 12, dtype=torch.float32, device=self.device)
        for ep in range(self.epochs):                                       # optimizer.py:469-475
            self.mq.process_data_events()
            out, status = self.engine.train_epoch(chunks, self.learning_rate, self.entropy_coef, self.vf_coef,
                                                  e_clip=self.e_clip, grad_hook=self.grad_hook)
            hist[ep, :11].copy_(out[:11])
            hist[ep, 11:].copy_(status)
        # the history goes to page-locked memory behind the last epoch; the host waits for THAT copy only, not for what it
        # enqueues further down (the next batch's rollout pass)
        if self._hist_host is None or self._hist_host.shape[0] != self.epochs:
            self._hist_host = torch.empty(self.epochs, 12, dtype=torch.float32).pin_memory()
        host = self._hist_host
        host.copy_(hist, non_blocking=True)
        hist_done = torch.cuda.Event()
        hist_done.record(torch.cuda.current_stream(self.device))
        if self.checkpoint:
            self._snapshot = self.engine.start_param_snapshot()            # D2H for the publish, behind the last epoch
        if self.prefetch:
            # While the GPU works through the epochs: drain what already waits in the experience queue into the next batch's staging.
            # Stops the moment the GPU has finished (event behind the last epoch): from then on gathering would delay the
            # model publish - and with it the agents' weight age - instead of hiding behind device work.
            done = hist_done
            t0 = time.time()
            n0 = len(self._acc['rollouts'])
            while self._packer.n_seq < self.min_seq_per_epoch and (self.prefetch_past_gpu_done or not done.query()) \
                    and self._queue_has_messages():
                self._consume_one(self._acc)
            self._acc['hidden_s'] += time.time() - t0
            self._acc['n_prefetched'] += len(self._acc['rollouts']) - n0
            if self.pipeline_rollout_pass and self._packer.n_seq >= self.min_seq_per_epoch:
                # the next batch is complete: its rollout pass (which needs exactly the weights the last epoch above produces) is
                # enqueued NOW, so the GPU runs it while the host synchronises, builds the metrics and serialises / publishes the
                # model (optimizer.py:697-716) - the publish is off the device's critical path
                self._acc['pipelined'] = 1
                self._ready = self._finish_batch()
        hist_done.synchronize()                                             # the one synchronisation of the iteration
        host = host.clone()
        how = ''
        if int(host[:, 11].max().item()) != 0:
            # data parallel: the all-reduced gradient of a NaN rank is NaN on every rank, so every rank is here; what they do next is
            # agreed on (the team kernels' fault record is rank-local)
            any_fault = None
            if self.grad_hook is not None and getattr(self.grad_hook, 'world', 1) > 1:
                flag = torch.tensor([1.0 if self.engine.fault() is not None else 0.0], device=self.device)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=self.grad_hook.group)
                any_fault = bool(flag.item() > 0)
            was_fast = self.engine.products == 'f16x2'
            how = self.engine.recover_from_nan(any_fault)
            if how and was_fast and self.engine.products == 'bf16x3':
                self._auto_safe, self._safe_left = True, self._safe_hold
                self._safe_hold = min(2 * self._safe_hold, 1024)
                how += ' for this iteration and the next %d' % (self._safe_left - 1)
        if how:
            # The NaN may be the kernels' doing: a team kernel that timed out (DC_WS_FAULT), or an operand outside the exponent range of
            # the two-f16-piece products (Engine.products).  Nothing was updated (the status word is sticky on the device): switch to the
            # safe path for good and repeat this iteration - rollout pass and epochs - on the same batch.  If it is NaN again, it is the
            # data: raise like the reference.
            logger.warning('iteration %d: NaN - repeating it with the %s', it, how)
            self._drop_ready()
            experiences, chunks = self._experiences_from_batch(acc['rollouts'], acc['batch'])
            for ep in range(self.epochs):
                out, status = self.engine.train_epoch(chunks, self.learning_rate, self.entropy_coef, self.vf_coef,
                                                      e_clip=self.e_clip, grad_hook=self.grad_hook)
                hist[ep, :11].copy_(out[:11])
                hist[ep, 11:].copy_(status)
            if self.checkpoint:
                self._snapshot = self.engine.start_param_snapshot()
            host = hist.cpu()
        losses, entropies, grad_norms = [], [], []
        for ep in range(self.epochs):
            row = host[ep]
            st = int(row[11].item())
            if st != 0:
                # the status word is sticky on the device (csrc/adam.hip): epochs behind the first NaN one applied nothing, like the
                # reference, which never reaches them; cleared here, with the early rollout pass of the next batch dropped
                msg = describe_fault(self.engine) + ('; already repeated with the ' + how if how else '')
                self.engine.status.zero_()
                self._drop_ready()
            if st == 1:                                                     # optimizer.py:667-669
                raise ValueError('loss={}, policy_loss={}, entropy_loss={}, value_loss={}'.format(*row[:4].tolist()) + msg)
            if st == 2:                                                     # optimizer.py:678-679
                raise ValueError('grad_norm={}'.format(row[9].item()) + msg)
            losses.append({'loss': row[0], 'policy_loss': row[1], 'entropy_loss': row[2], 'value_loss': row[3]})
            entropies.append({k: row[4 + i] for i, k in enumerate(L.OUTPUT_KEYS)})
            grad_norms.append({'unclipped': row[9], 'clipped': row[10]})
        time_opt = time.time() - start_opt
        losses = self.list_of_dicts_to_dict_of_lists(losses)
        entropies = self.list_of_dicts_to_dict_of_lists(entropies)
        grad_norms = self.list_of_dicts_to_dict_of_lists(grad_norms)
        n_steps = len(experiences) * self.seq_len                           # optimizer.py:486
        sub = np.stack(subrewards) / n_steps * Policy.OBSERVATIONS_PER_SECOND
        time_it = time.time() - self.time_last_it
        self.time_last_it = time.time()
        metrics = {
            self.SPEED_KEY: n_steps / time_it,
            'reward_per_sec/sum': sub.sum(),
            'loss/sum': losses['loss'].mean(), 'loss/policy': losses['policy_loss'].mean(),
            'loss/entropy': losses['entropy_loss'].mean(), 'loss/value': losses['value_loss'].mean(),
            'entropy': torch.stack(list(entropies.values())).sum(dim=0).mean(),
            'avg_rollout_len': float(np.mean(rollout_lens)), 'avg_weight_age': float(np.mean(weight_ages)),
            'timing/it': time_it, 'timing/xp_total': time_xp, 'timing/xp_mq_wait': xp_waits,
            'timing/optimizer': time_opt,
            # not in the reference: the part of gathering + packing this batch that ran during the previous iteration's epochs, and how
            # many of its rollouts came in that way (timing/xp_total is the foreground part only)
            'timing/xp_hidden': acc['hidden_s'], 'xp_rollouts_prefetched': float(acc['n_prefetched']),
            'xp_rollout_pass_pipelined': float(acc['pipelined']),          # 1: this batch's rollout pass ran behind the previous iteration's epochs
        }
        for k, v in entropies.items():
            metrics['entropy/' + k] = v.mean()
        for k, v in grad_norms.items():
            metrics['grad_norm/' + k] = v.mean()
        for k, v in zip(REWARD_KEYS, sub.sum(axis=0)):
            metrics['reward_per_sec/' + k] = v
        return metrics

    def run(self):
        for it in range(self.iteration_start, self.iterations):
            metrics = self.run_iteration(it)
            logger.info('iteration %d steps_per_s=%.2f loss=%.4f', it, metrics[self.SPEED_KEY], float(metrics['loss/sum']))
            if self.metrics_sink is not None:
                self.metrics_sink(metrics, it)
            if self.checkpoint:
                self.upload_model(version=it)

    def upload_model(self, version):
        """optimizer.py:697-723: rank 0 serialises state_dict() -> file + model exchange."""
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_rank() != 0:
            return
        # one D2H copy of the flat parameter buffer (started right after the last epoch was enqueued, run_iteration) instead
        # of 34 synchronous per-tensor copies; same names / shapes / dtypes on the wire
        if self._snapshot is None:
            self._snapshot = self.engine.start_param_snapshot()
        buf = io.BytesIO()
        # 34 tensors that own their storage, like the reference's state_dict (a blob of 34 views would serialise ONE shared 3 MB storage:
        # loads fine with strict=True, but in-place edits / per-tensor slicing of the file would behave differently - ADVICE r4)
        torch.save(self.engine.snapshot_state_dict(self._snapshot, clone=True), buf)
        self._snapshot = None
        blob = buf.getvalue()
        if self.checkpoint:
            with open(os.path.join(self.log_dir, self.MODEL_FILENAME_FMT % version), 'wb') as f:
                f.write(blob)
        self.mq.publish_model(msg=blob, hdr={'version': version})
