"""GPU: the reference's call surface (Policy / DotaOptimizer / advantage_returns) on top of the HIP path."""
import pickle

import numpy as np
import pytest
import torch

from dotaclient_amd import layout as L
from dotaclient_amd import synth
from oracle import ref_optimizer as RO
from tests import util

pytestmark = pytest.mark.gpu


class FakeMQ:
    """In-memory stand-in for the RabbitMQ client (optimizer.py:67-174 surface)."""

    def __init__(self, rollouts):
        self.bodies = [pickle.dumps(r) for r in rollouts]
        self.published = []

    def connect(self): pass
    def process_data_events(self): pass
    def consume_xp(self): return None, None, self.bodies.pop(0)
    def publish_model(self, msg, hdr): self.published.append((hdr, len(msg)))

    @property
    def xp_queue_size(self):                       # optimizer.py:126-132
        return len(self.bodies)


def make_opt(rollouts, g, tmp_path, **kw):
    from dotaclient_amd.optimizer import DotaOptimizer
    opt = DotaOptimizer(rmq_host='x', rmq_port=0, epochs=int(g['epochs']), min_seq_per_epoch=1, seq_len=int(g['seq_len']),
                        learning_rate=float(g['lr']), checkpoint=False, pretrained_model=None, mq_prefetch_count=1,
                        log_dir=str(tmp_path), entropy_coef=float(g['entropy_coef']), vf_coef=float(g['vf_coef']),
                        run_local=True, mq=FakeMQ(rollouts), **kw)
    opt.policy_base.load_state_dict(synth.init_state_dict(7), strict=True)
    return opt


def test_advantage_returns_signature_and_known_answer():
    from dotaclient_amd.optimizer import advantage_returns
    adv, ret = advantage_returns(np.array([1, 2, 3, 0], np.float32), np.array([.5, .4, .3, 0], np.float32), 0.98, 0.97)
    np.testing.assert_allclose(adv, [5.1322656, 4.4606204, 2.7], rtol=1e-6)
    np.testing.assert_allclose(ret, [5.8412, 4.94, 3.0], rtol=1e-6)
    assert adv.dtype == np.float32 and ret.shape == (3,)


def test_state_dict_is_the_reference_wire_format():
    from dotaclient_amd.policy import Policy
    pol = Policy()
    sd = pol.state_dict()
    want = L.param_shapes()
    assert list(sd.keys()) == list(want.keys()) and len(sd) == 34
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(want[k]) and v.dtype == torch.float32
    assert sum(v.numel() for v in sd.values()) == 765210
    assert [n for n, _ in pol.named_parameters()] == list(want.keys())
    with pytest.raises(RuntimeError):
        pol.load_state_dict({'bogus': torch.zeros(1)}, strict=True)


def test_policy_forward_matches_oracle():
    from dotaclient_amd.policy import Policy
    sd = synth.init_state_dict(7)
    pol = Policy()
    pol.load_state_dict(sd)
    ref = RO.make_policy(sd)
    rollouts = synth.make_rollouts(9, [24, 24, 24])
    obs = {k: torch.stack([r['observations'][k] for r in rollouts]) for k in L.INPUT_KEYS}
    h0 = 0.1 * torch.randn(1, 3, 256, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        rl, rv, rh = ref(obs, h0)
    logits, value, hidden = pol(**{k: v.cuda() for k, v in obs.items()}, hidden=h0.cuda())
    for k in L.OUTPUT_KEYS:
        assert logits[k].shape == rl[k].shape
        assert util.scaled_err(logits[k].cpu().numpy(), rl[k].numpy()) < 1e-5, k
    assert util.scaled_err(value.cpu().numpy(), rv.numpy()) < 1e-5
    assert util.scaled_err(hidden.cpu().numpy(), rh.numpy()) < 1e-5
    # single-step / single-sequence entry points (policy.py:80-90)
    lg, v, h = pol.sequence(**{k: v[0].cuda() for k, v in obs.items()}, hidden=pol.init_hidden())
    assert lg['target_unit'].shape == (1, 24, 40) and v.shape == (1, 24, 1) and h.shape == (1, 1, 256)


@pytest.mark.parametrize('mode', ['kernel', 'graph', 'eager'])
@pytest.mark.parametrize('cell,hidden,layers', [('gru', 256, 1), ('lstm', 128, 1), ('lstm', 256, 1), ('lstm', 512, 2), ('gru', 64, 3)])
def test_policy_single_step_matches_oracle(cell, hidden, layers, mode):
    # the actor-side entry point (policy.py:80-84, agent.py:652): B = 1, S = 1, hidden state carried by the caller.  'kernel' = the one-kernel
    # step (csrc/policy_single.hip, the default), 'graph' / 'eager' = the batch path's kernels on a padded tile
    from dotaclient_amd.policy import Policy
    if mode != 'kernel' and (hidden, layers) == (64, 3):
        pytest.skip('the batch kernels take H in (128, 256, 512)')
    sd = synth.init_state_dict(7, cell, hidden, layers)
    pol = Policy(cell, hidden, layers)
    pol.load_state_dict(sd)
    pol.single_kernel, pol.single_graph = mode == 'kernel', mode == 'graph'
    ref = RO.make_policy(sd, cell, hidden, layers)
    r = synth.make_rollouts(12, [6])[0]
    hid = pol.init_hidden()
    rhid = ref.init_hidden(1)
    for t in range(6):
        obs_t = {k: r['observations'][k][t] for k in L.INPUT_KEYS}
        lg, v, hid = pol.single(**{k: (x.cuda() if t % 2 else x) for k, x in obs_t.items()}, hidden=hid)
        with torch.no_grad():
            rl, rv, rhid = ref({k: x[None, None] for k, x in obs_t.items()}, rhid)
        for k in L.OUTPUT_KEYS:
            assert lg[k].shape == rl[k].shape == (1, 1, L.HEAD_COUNTS[k])
            assert util.scaled_err(lg[k].cpu().numpy(), rl[k].numpy()) < 1e-5, (t, k)
        assert v.shape == (1, 1, 1) and util.scaled_err(v.cpu().numpy(), rv.numpy()) < 1e-5
        for h_got, h_ref in zip([hid] if cell == 'gru' else hid, [rhid] if cell == 'gru' else rhid):
            assert h_got.shape == (layers, 1, hidden) and util.scaled_err(h_got.cpu().numpy(), h_ref.numpy()) < 1e-5


def test_policy_single_kernel_back_to_back_calls_do_not_race():
    # the one-kernel step reads the observation straight from a pinned host row: a caller that issues calls WITHOUT reading a result in
    # between (nothing synchronises) must still get every step computed on its own observation (two rows in turn, each guarded by an
    # event - ADVICE r5's hazard); the same steps with a read-back after each are the yardstick (bit-identical: same kernel, same inputs)
    from dotaclient_amd.policy import Policy
    pol = Policy('lstm', 256, 1)
    pol.load_state_dict(synth.init_state_dict(7, 'lstm', 256, 1))
    r = synth.make_rollouts(14, [24])[0]
    runs = {}
    for read_back in (True, False):
        hid, seq = pol.init_hidden(), []
        for t in range(24):
            lg, v, hid = pol.single(**{k: r['observations'][k][t] for k in L.INPUT_KEYS}, hidden=hid)
            if read_back:
                v.cpu()
            seq.append((lg, v, hid))
        runs[read_back] = torch.stack([torch.cat([lg[k].flatten() for k in L.OUTPUT_KEYS] + [v.flatten(), hid[0].flatten(), hid[1].flatten()])
                                       for lg, v, hid in seq]).cpu()
    assert torch.equal(runs[True], runs[False])
    assert not torch.isnan(runs[True]).any()
    # a hidden state handed over as CPU tensors / a non-contiguous view is taken too
    hid = tuple(h.cpu() for h in pol.init_hidden())
    lg, v, hid2 = pol.single(**{k: r['observations'][k][0] for k in L.INPUT_KEYS}, hidden=hid)
    assert torch.equal(torch.cat([lg[k].flatten() for k in L.OUTPUT_KEYS]).cpu(), runs[True][0, :65])


def test_policy_single_kernel_refuses_what_it_cannot_run():
    # dc_policy_single's argument checks (include/dotaclient_hip.h): no silent fallback
    import ctypes
    from dotaclient_amd import _lib
    from dotaclient_amd.engine import DcDims
    from dotaclient_amd.policy import Policy
    pol = Policy('gru', 256, 1)
    e = pol.engine
    buf = torch.zeros(_lib.DC_SINGLE_SCRATCH_FLOATS, device='cuda')
    for dims, code in ((DcDims(0, 96, 1, 1, 1, 0, 1), 1022), (DcDims(0, 1024, 1, 1, 1, 0, 1), 1022), (DcDims(2, 256, 1, 1, 1, 0, 1), 1021),
                       (DcDims(0, 256, 0, 1, 1, 0, 1), 1020)):
        rc = e.lib.dc_policy_single(ctypes.byref(dims), _lib.ptr(e.params), e.poff, _lib.ptr(buf), None, None, _lib.ptr(buf), _lib.ptr(buf), None,
                                    _lib.ptr(buf), _lib.stream_ptr())
        assert rc == code
    rc = e.lib.dc_policy_single(ctypes.byref(DcDims(0, 256, 1, 1, 1, 0, 1)), _lib.ptr(e.params), e.poff, _lib.ptr(buf), None, None, None,
                                _lib.ptr(buf), None, _lib.ptr(buf), _lib.stream_ptr())
    assert rc == 1024


@pytest.mark.parametrize('cell,hidden', [('gru', 256), ('lstm', 256)])
def test_policy_single_graph_replay_equals_eager(cell, hidden):
    # (f4) the actor-latency path: Policy.single replays ONE hipGraph over static buffers (CPU observation tensors in, like the
    # actor's, agent.py:640-652) - bit-identical to the eager forward, step after step with the hidden state carried by the caller,
    # across a weight update (Policy.load_state_dict: same buffers, new values) - and faster
    import time
    from dotaclient_amd.policy import Policy
    pol = Policy(cell, hidden, 1)
    pol.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
    r = synth.make_rollouts(13, [40])[0]
    pol.single_kernel = False
    outs = {}
    for mode in (True, False):
        pol.single_graph = mode
        hid = pol.init_hidden()
        seq = []
        for t in range(12):
            if t == 6:
                pol.load_state_dict(synth.init_state_dict(8, cell, hidden, 1))      # the model exchange delivers new weights mid-game
            lg, v, hid = pol.single(**{k: r['observations'][k][t] for k in L.INPUT_KEYS}, hidden=hid)
            seq.append(torch.cat([lg[k].flatten() for k in L.OUTPUT_KEYS] + [v.flatten(), (hid if cell == 'gru' else hid[0]).flatten()]).cpu())
        pol.load_state_dict(synth.init_state_dict(7, cell, hidden, 1))
        outs[mode] = torch.stack(seq)
    assert torch.equal(outs[True], outs[False])
    assert pol._single_state['graph'] is not None
    lat = {}
    for mode in ('kernel', True, False):
        pol.single_kernel, pol.single_graph = mode == 'kernel', mode is True
        hid = pol.init_hidden()
        for t in range(3):
            _, _, hid = pol.single(**{k: r['observations'][k][t] for k in L.INPUT_KEYS}, hidden=hid)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(30):
            lg, v, hid = pol.single(**{k: r['observations'][k][t] for k in L.INPUT_KEYS}, hidden=hid)
            v.cpu()                                                                 # the actor reads the result every step
        lat[mode] = (time.perf_counter() - t0) / 30 * 1e6
    print('Policy.single latency per env-step (%s-%d): one kernel %.0f us, graph replay %.0f us, eager %.0f us'
          % (cell, hidden, lat['kernel'], lat[True], lat[False]))
    assert lat['kernel'] < lat[True] < lat[False]


@pytest.mark.parametrize('case', ['ragged_s16', 'clip_s16'])
def test_optimizer_surface_matches_golden(case, tmp_path):
    g, rollouts = util.load_case(case)
    opt = make_opt(rollouts, g, tmp_path)
    assert opt.mq.published and opt.mq.published[0][0] == {'version': 1}     # initial model upload
    # the reference's own flow: one get_rollout + experiences_from_rollout per message (optimizer.py:448-458)
    experiences = []
    while opt.mq.bodies:
        data, subrewards, rollout_len, version, canvas = opt.get_rollout()
        assert subrewards.shape == (10,)
        experiences.extend(opt.experiences_from_rollout(data))
    assert len(experiences) == g['advantages'].shape[0]
    adv = torch.stack([e.advantages for e in experiences]).cpu().numpy()
    ret = torch.stack([e.returns for e in experiences]).cpu().numpy()
    assert util.scaled_err(adv, g['advantages']) < 1e-4 and util.scaled_err(ret, g['returns']) < 1e-4
    e0 = experiences[1]
    assert e0.observations['enemy_heroes'].shape == (int(g['seq_len']), 5, 12) and e0.hidden.shape == (1, 1, 256)
    for k in L.OUTPUT_KEYS:
        got = torch.cat([e.log_probs_sel[k] for e in experiences]).cpu().numpy()
        assert util.scaled_err(got, g['old_logp_' + k]) < 1e-4
    for ep in range(int(g['epochs'])):
        losses, entropies, norms = opt.train(experiences=experiences)
        assert set(losses) == {'loss', 'policy_loss', 'entropy_loss', 'value_loss'} and losses['loss'].dim() == 0
        got = np.array([float(losses[k]) for k in ('loss', 'policy_loss', 'entropy_loss', 'value_loss')])
        assert util.loss_rel_err(got, g['ep%d_losses' % ep]) < 1e-4
        assert util.rel_err([float(entropies[k]) for k in L.OUTPUT_KEYS], g['ep%d_entropies' % ep]) < 1e-4
        assert util.rel_err([float(norms['unclipped']), float(norms['clipped'])], g['ep%d_grad_norms' % ep]) < 1e-4
    # .grad views expose the (clipped) gradients like the reference's parameters do
    assert util.rel_err(float(opt.mean_gradient_norm()), g['ep%d_grad_norms' % (int(g['epochs']) - 1)][1]) < 1e-4


def test_run_iteration_and_nan_guard(tmp_path):
    g, rollouts = util.load_case('ragged_s16')
    opt = make_opt(rollouts, g, tmp_path)
    opt.min_seq_per_epoch = 11
    m = opt.run_iteration(1)
    assert m['steps per s'] > 0 and np.isfinite(float(m['loss/sum']))
    # NaN loss -> ValueError and parameters untouched (optimizer.py:667-669)
    before = opt.engine.params.clone()
    bad = synth.make_rollouts(77, [32])
    bad[0]['observations']['env'][3, 0] = float('nan')
    seqs = opt.experiences_from_rollout(bad[0])
    with pytest.raises(ValueError):
        opt.train(experiences=seqs)
    assert torch.equal(before, opt.engine.params)


def test_run_iteration_raises_on_nan_like_the_reference(tmp_path):
    # a NaN observation inside run_iteration itself: the epochs are enqueued back to back and checked once at the end - ValueError
    # (optimizer.py:667-669) and untouched parameters all the same
    g, rollouts = util.load_case('ragged_s16')
    bad = synth.make_rollouts(77, [32, 48])
    bad[1]['observations']['env'][3, 0] = float('nan')
    opt = make_opt(bad, g, tmp_path)
    opt.min_seq_per_epoch = 4
    before = opt.engine.params.clone()
    with pytest.raises(ValueError):
        opt.run_iteration(1)
    assert torch.equal(before, opt.engine.params)


def test_nan_error_names_the_team_kernel_that_timed_out(tmp_path):
    # the reference's NaN guard (optimizer.py:667-669) is how a timed-out team kernel surfaces (it NaN-poisons its outputs); the
    # workspace's DC_WS_FAULT record says which launch it was, and the ValueError carries it.  Here the record is
    # injected by hand next to a NaN observation.
    from dotaclient_amd.engine import describe_status
    g, _ = util.load_case('ragged_s16')
    bad = synth.make_rollouts(77, [32, 48])
    bad[1]['observations']['env'][3, 0] = float('nan')
    opt = make_opt(bad, g, tmp_path)
    opt.min_seq_per_epoch = 4
    seqs = opt.experiences_from_rollouts([dict(r) for r in bad])[0]
    assert opt.engine.fault() is None and 'timeout' not in describe_status(opt.engine)
    opt.engine._ws[:32].view(torch.int32).copy_(torch.tensor([16 + 3, 0, 5, 2, 17, 40, 18, 0], dtype=torch.int32))
    with pytest.raises(ValueError, match='team kernel timeout: team_mfma_fwd, layer 0, team 5, member 2, time step 17, sequence 40'):
        opt.train(experiences=seqs)
    opt.engine.clear_fault()
    assert opt.engine.fault() is None


def test_prefetching_consumer_loop_equals_the_serial_one(tmp_path):
    # run_iteration packs every rollout as it arrives and, while the GPU works through the epochs, already drains
    # the experience queue into the NEXT batch's staging.  Same rollout stream through a prefetching and a serial optimizer: the
    # batches of every iteration are bit-identical, the metrics equal, and the prefetched part is not counted in timing/xp_total.
    g, _ = util.load_case('ragged_s16')
    stream = synth.make_rollouts(31, [40, 64, 21, 33, 50, 16, 64, 48, 17, 80, 30, 64, 25, 70, 44, 16, 90, 35])
    opts = {pf: make_opt([dict(r) for r in stream], g, tmp_path, prefetch=pf) for pf in (True, False)}
    seen = {True: [], False: []}
    for pf, opt in opts.items():
        opt.min_seq_per_epoch = 9
        opt.prefetch_past_gpu_done = True          # deterministic: these tiny epochs may finish before the host has unpickled a rollout
        orig = opt._experiences_from_batch

        def spy(rollouts, batch, _orig=orig, _pf=pf):
            seen[_pf].append({k: getattr(batch, k).clone() for k in ('obs', 'act', 'mask', 'rew', 'seq_off', 'seq_len')})
            return _orig(rollouts, batch)
        opt._experiences_from_batch = spy
    metrics = {pf: [opt.run_iteration(it) for it in (1, 2, 3)] for pf, opt in opts.items()}
    for it in range(3):
        a, b = seen[True][it], seen[False][it]
        for k in a:
            assert torch.equal(a[k], b[k]), (it, k)
        ma, mb = metrics[True][it], metrics[False][it]
        for k in ('loss/sum', 'loss/policy', 'loss/value', 'entropy', 'grad_norm/unclipped', 'avg_rollout_len', 'avg_weight_age', 'reward_per_sec/sum'):
            assert abs(float(ma[k]) - float(mb[k])) <= 2e-5 * max(1.0, abs(float(mb[k]))), (it, k, float(ma[k]), float(mb[k]))
    assert metrics[False][1]['xp_rollouts_prefetched'] == 0 and metrics[False][1]['timing/xp_hidden'] == 0
    # iterations 2 and 3 found their rollouts already packed
    assert metrics[True][1]['xp_rollouts_prefetched'] >= 1 and metrics[True][2]['xp_rollouts_prefetched'] >= 1
    assert metrics[True][1]['timing/xp_hidden'] > 0
    assert torch.allclose(opts[True].engine.params, opts[False].engine.params, rtol=0, atol=2e-6)


def test_pretrained_model_sets_the_weights_and_the_version_counter(tmp_path):
    # optimizer.py:256-267: an explicitly given model file is loaded and the published versions carry on behind its number (finding the
    # newest file of a checkpoint directory / bucket is the integrator's launcher's job: storage plumbing)
    g, rollouts = util.load_case('ragged_s16')
    log_dir = tmp_path / 'fresh' / 'logs'
    from dotaclient_amd.optimizer import DotaOptimizer
    kw = dict(rmq_host='x', rmq_port=0, epochs=1, min_seq_per_epoch=1, seq_len=16, learning_rate=1e-4,
              mq_prefetch_count=1, log_dir=str(log_dir), entropy_coef=5e-4, vf_coef=0.5, run_local=True)
    a = DotaOptimizer(checkpoint=True, pretrained_model=None, mq=FakeMQ(rollouts), **kw)
    assert a.iteration_start == 1 and (log_dir / 'model_000000001.pt').exists()
    a.policy_base.load_state_dict(synth.init_state_dict(11), strict=True)
    a.upload_model(version=7)
    b = DotaOptimizer(checkpoint=True, pretrained_model=str(log_dir / 'model_000000007.pt'), mq=FakeMQ(rollouts), **kw)
    assert b.iteration_start == 8
    assert (log_dir / 'model_000000008.pt').exists() and b.mq.published[-1][0] == {'version': 8}
    assert torch.equal(a.engine.params, b.engine.params)
    assert DotaOptimizer.iteration_from_model_filename('x/model_000000123.pt') == 123
    # ... and a used log_dir WITHOUT a model path is refused (the reference would resume from model_000000008.pt by itself; starting at
    # version 1 here would publish versions that go backwards - tests/test_host_logic.py has the CPU form of this check)
    with pytest.raises(ValueError, match='model_000000008.pt'):
        DotaOptimizer(checkpoint=True, pretrained_model=None, mq=FakeMQ(rollouts), **kw)


def test_model_publish_is_the_reference_wire_format(tmp_path):
    # optimizer.py:697-716: every iteration rank 0 serialises state_dict() and publishes it; the actors load it with
    # strict=True (agent.py:186,315).  Here: ONE asynchronous D2H copy of the flat parameter buffer (started when the
    # last epoch is enqueued) instead of 34 per-tensor copies - same names, shapes, dtypes and values on the wire.
    import io
    g, rollouts = util.load_case('ragged_s16')
    opt = make_opt(rollouts, g, tmp_path)
    opt.checkpoint = True
    opt.min_seq_per_epoch = 11
    opt.run_iteration(1)
    assert opt._snapshot is not None                         # the copy was started inside run_iteration
    want = {k: v.cpu() for k, v in opt.engine.state_dict().items()}
    blobs = []
    opt.mq.publish_model = lambda msg, hdr: blobs.append((hdr, msg))
    opt.upload_model(version=1)
    (hdr, blob), = blobs
    assert hdr == {'version': 1}
    sd = torch.load(io.BytesIO(blob))
    assert list(sd.keys()) == list(L.param_shapes().keys())
    for k, v in sd.items():
        assert v.dtype == torch.float32 and v.device.type == 'cpu' and torch.equal(v, want[k]), k
    RO.make_policy(sd).load_state_dict(sd, strict=True)      # what an actor does with it
    # 34 independent storages like the reference's blob (views of one flat buffer would serialise one shared storage)
    ptrs = {v.untyped_storage().data_ptr() for v in sd.values()}
    assert len(ptrs) == len(sd) and all(v.untyped_storage().nbytes() == v.numel() * 4 for v in sd.values())
    # a later optimizer step must not leak into a snapshot that was already taken (double buffering)
    i = opt.engine.start_param_snapshot()
    before = {k: v.clone() for k, v in opt.engine.snapshot_state_dict(i).items()}
    opt.engine.params.add_(1.0)
    opt.engine.start_param_snapshot()
    for k, v in opt.engine.snapshot_state_dict(i).items():
        assert torch.equal(v, before[k])


def test_next_rollout_pass_is_enqueued_before_the_publish(tmp_path):
    # the model publish (torch.save + file + MQ, optimizer.py:697-716) must not idle the GPU.  With the next batch
    # already waiting in the queue, run_iteration(k) enqueues batch k+1's rollout pass behind its last epoch, so by the time
    # publish_model(k) is called the device already has iteration k+1's kernels; the results equal a non-pipelined optimizer's.
    g, _ = util.load_case('ragged_s16')
    stream = synth.make_rollouts(31, [40, 64, 21, 33, 50, 16, 64, 48, 17, 80, 30, 64, 25, 70, 44, 16, 90, 35])
    seen = {}
    params = {}
    for pipe in (True, False):
        opt = make_opt([dict(r) for r in stream], g, tmp_path, prefetch=True)
        opt.checkpoint, opt.min_seq_per_epoch, opt.prefetch_past_gpu_done, opt.pipeline_rollout_pass = True, 9, True, pipe
        log = seen[pipe] = []
        opt.mq.publish_model = lambda msg, hdr, _opt=opt, _log=log: _log.append(
            (hdr['version'], _opt._ready is not None, None if _opt._ready is None else _opt._ready[2].values is not None))
        metrics = []
        for it in (1, 2, 3):
            metrics.append(opt.run_iteration(it))
            opt.upload_model(version=it)
        params[pipe] = opt.engine.params.clone()
        seen[pipe] = (log, metrics)
    log, metrics = seen[True]
    # iterations 1 and 2 found the next batch complete in the queue: its rollout pass (values filled in) was enqueued before the publish
    assert log[0] == (1, True, True) and log[1] == (2, True, True)
    assert metrics[1]['xp_rollout_pass_pipelined'] == 1.0 and metrics[2]['xp_rollout_pass_pipelined'] == 1.0
    assert metrics[0]['xp_rollout_pass_pipelined'] == 0.0
    assert all(not l[1] for l in seen[False][0])
    for a, b in zip(metrics, seen[False][1]):
        for k in ('loss/sum', 'loss/policy', 'loss/value', 'entropy', 'grad_norm/unclipped', 'avg_weight_age'):
            assert abs(float(a[k]) - float(b[k])) <= 2e-5 * max(1.0, abs(float(b[k]))), k
    assert torch.allclose(params[True], params[False], rtol=0, atol=2e-6)


def test_nan_status_is_sticky_until_the_caller_clears_it(tmp_path):
    # epochs are enqueued back to back and checked once; an epoch behind a NaN one must not apply its update (the
    # reference raises before it would run).  Engine level: NaN batch -> status 1; a GOOD batch's epoch without clearing changes
    # nothing and keeps the word; after clearing it steps.
    g, rollouts = util.load_case('ragged_s16')
    opt = make_opt(rollouts, g, tmp_path)
    eng = opt.engine
    from dotaclient_amd.engine import pack_rollouts
    bad = synth.make_rollouts(77, [32])
    bad[0]['observations']['env'][3, 0] = float('nan')
    good = synth.make_rollouts(78, [32, 48])
    before = eng.params.clone()
    cb = eng.rollout_pass(pack_rollouts(bad, 16, eng.device), 16)
    eng.train_epoch(cb, 1e-4, 5e-4, 0.5)
    st = int(eng.status.item())
    assert st in (1, 2)                          # NaN loss or NaN gradient norm, whichever guard sees it first
    cg = eng.rollout_pass(pack_rollouts(good, 16, eng.device), 16)
    steps = eng.seg_step.clone()
    eng.train_epoch(cg, 1e-4, 5e-4, 0.5)
    assert int(eng.status.item()) == st and torch.equal(before, eng.params) and torch.equal(steps, eng.seg_step)
    eng.status.zero_()
    eng.train_epoch(cg, 1e-4, 5e-4, 0.5)
    assert int(eng.status.item()) == 0 and not torch.equal(before, eng.params)
    # DotaOptimizer.train clears the word when it raises: the next call works (like the reference, whose guards keep no state)
    seqs = opt.experiences_from_rollout(bad[0])
    with pytest.raises(ValueError):
        opt.train(experiences=seqs)
    opt.train(experiences=opt.experiences_from_rollouts(good)[0])


def test_mq_is_built_like_the_reference_when_none_is_passed(tmp_path, monkeypatch):
    # INTEGRATION.md option A: the reference's main() (optimizer.py:751-765) passes no queue object - its constructor builds
    # MessageQueue(host, port, prefetch_count, use_model_exchange) itself (optimizer.py:278-280).  With mq=None this class does the
    # same with the class it finds in __main__ (the reference's optimizer.py, which still defines it) or in DotaOptimizer.MessageQueue.
    import sys
    from dotaclient_amd.optimizer import DotaOptimizer
    made = []

    class MessageQueue(FakeMQ):
        def __init__(self, host, port, prefetch_count, use_model_exchange):
            super().__init__([])
            made.append((host, port, prefetch_count, use_model_exchange))
    kw = dict(rmq_host='h', rmq_port=5672, epochs=1, min_seq_per_epoch=1, seq_len=16, learning_rate=1e-4, checkpoint=False,
              pretrained_model=None, mq_prefetch_count=3, log_dir=str(tmp_path), entropy_coef=5e-4, vf_coef=0.5, run_local=True)
    monkeypatch.setattr(sys.modules['__main__'], 'MessageQueue', MessageQueue, raising=False)
    opt = DotaOptimizer(**kw)
    assert made == [('h', 5672, 3, False)] and isinstance(opt.mq, MessageQueue) and opt.mq.published[0][0] == {'version': 1}
    monkeypatch.delattr(sys.modules['__main__'], 'MessageQueue')
    monkeypatch.setattr(DotaOptimizer, 'MessageQueue', MessageQueue)
    DotaOptimizer(**kw)
    assert len(made) == 2
    monkeypatch.setattr(DotaOptimizer, 'MessageQueue', None)
    with pytest.raises(ValueError, match='MessageQueue'):
        DotaOptimizer(**kw)
    with pytest.raises(ValueError, match='run_local'):
        DotaOptimizer(**dict(kw, run_local=False))


def test_consumer_loop_falls_back_to_the_bf16_pieces_when_an_operand_leaves_the_f16_range(tmp_path):
    # Engine.products: the default two-f16-piece products cannot represent an activation beyond 4094 (DC_DIMS_F16X2); the loss turns
    # NaN, the device skips every update (sticky status), and run_iteration repeats the iteration with the three-bf16-piece products
    # (f32's exponent range), which stay on from there.  The reference handles such an input without a hiccup - so must the drop-in.
    g, _ = util.load_case('ragged_s16')
    stream = synth.make_rollouts(31, [40, 64, 21, 33, 50, 16, 64, 48, 17, 80, 30, 64])
    stream[2]['observations']['env'][5, 1] = 3.0e6
    opt = make_opt([dict(r) for r in stream], g, tmp_path)
    opt.min_seq_per_epoch = 9
    assert opt.engine.products == 'f16x2'
    before = opt.engine.params.clone()
    m = opt.run_iteration(1)
    assert opt.engine.products == 'bf16x3' and int(opt.engine.status.item()) == 0
    assert np.isfinite(float(m['loss/sum'])) and not torch.equal(before, opt.engine.params)
    # the same stream through an optimizer that was on the safe products from the start: same result
    ref = make_opt([dict(r) for r in stream], g, tmp_path)
    ref.engine.products = 'bf16x3'
    ref.min_seq_per_epoch = 9
    mr = ref.run_iteration(1)
    assert abs(float(m['loss/sum']) - float(mr['loss/sum'])) <= 1e-6 * max(1.0, abs(float(mr['loss/sum'])))
    assert torch.allclose(opt.engine.params, ref.engine.params, rtol=0, atol=1e-7)
    m2 = opt.run_iteration(2)                       # and the loop carries on
    assert np.isfinite(float(m2['loss/sum']))


def test_nan_recovery_with_a_pipelined_rollout_pass_loses_and_duplicates_nothing(tmp_path):
    # the riskiest host logic - sticky status + the NEXT batch's rollout pass already enqueued (pipeline_rollout_pass) + the
    # repeat of the poisoned iteration.  A stream whose first batch holds an out-of-range observation, consumed by the pipelined loop and
    # by a plain serial loop that was on the safe products from the start: same batches (every rollout exactly once, in order), same
    # metrics, same parameters; the fallback is held for a few iterations, then the fast products are probed again and stay.
    g, _ = util.load_case('ragged_s16')
    lens = [40, 64, 21, 33, 50, 16, 64, 48, 17, 80, 30, 64, 25, 70, 41, 64, 18, 52, 33, 47, 64, 29, 55, 38, 61, 20, 44, 36, 58, 27]
    stream = synth.make_rollouts(35, lens)
    stream[1]['observations']['env'][5, 1] = 3.0e6
    n_it = 4

    def run(pipelined):
        opt = make_opt([dict(r) for r in stream], g, tmp_path, prefetch=pipelined)
        opt.min_seq_per_epoch = 9
        opt.pipeline_rollout_pass = pipelined
        opt.prefetch_past_gpu_done = True                   # deterministic batches: keep draining the queue
        opt._safe_hold = 2                                  # probe the fast products again after two iterations
        if not pipelined:
            opt.engine.products = 'bf16x3'
        seen, ms, prods, snaps = [], [], [], []
        orig = opt._experiences_from_batch
        opt._experiences_from_batch = lambda rollouts, batch: (seen.append([r['game_id'] for r in rollouts]), orig(rollouts, batch))[1]
        for it in range(1, n_it + 1):
            ms.append(opt.run_iteration(it))
            prods.append(opt.engine.products)
            snaps.append(opt.engine.params.clone())
        return opt, seen, ms, prods, snaps

    a, seen_a, ms_a, prods_a, snaps_a = run(True)
    b, seen_b, ms_b, prods_b, snaps_b = run(False)
    # iteration 1 was repeated on bf16x3 (its batch appears twice in a row among the rollout passes), then two held iterations, then f16x2 again
    assert prods_a == ['bf16x3', 'bf16x3', 'f16x2', 'f16x2'] and a._safe_hold == 4, (prods_a, a._safe_hold)
    assert ms_a[0]['xp_rollout_pass_pipelined'] == 0 and any(m['xp_rollout_pass_pipelined'] == 1 for m in ms_a[1:])
    # rollout passes in order: batch 1, (batch 2: pipelined behind iteration 1's epochs, then dropped), batch 1 again on bf16x3, batch 2 again, ...
    dedup = []
    for x in seen_a:
        if x not in dedup:
            dedup.append(x)
    assert seen_a.count(seen_a[0]) == 2, 'the poisoned iteration was not repeated'
    flat = [gid for batch in dedup[:n_it] for gid in batch]
    assert flat == [r['game_id'] for r in stream[:len(flat)]], 'a rollout was lost, duplicated or reordered'
    assert dedup[:n_it] == seen_b[:n_it]
    for it, (ma, mb) in enumerate(zip(ms_a, ms_b)):
        # iterations 1-2: the same arithmetic (bf16x3) in both loops -> tight; 3-4: f16x2 against bf16x3, both f32-grade, but Adam turns
        # rounding noise on near-zero gradients into +-lr steps, so only the metrics are compared there
        tol = 2e-5 if it < 2 else 2e-3
        for k in ('loss/sum', 'loss/policy', 'loss/value', 'entropy', 'grad_norm/unclipped', 'avg_rollout_len'):
            assert abs(float(ma[k]) - float(mb[k])) <= tol * max(1.0, abs(float(mb[k]))), (it, k, float(ma[k]), float(mb[k]))
    for it in range(2):
        assert torch.allclose(snaps_a[it], snaps_b[it], rtol=0, atol=2e-6), it
    assert int(a.engine.status.item()) == 0 and torch.isfinite(a.engine.params).all()


def test_consumer_loop_leaves_the_team_kernels_after_a_recorded_timeout(tmp_path):
    # The team kernels' workgroups wait for each other and assume a compute unit each; when that does not hold (CU mask, co-tenant) a wait
    # times out, the launch records it in DC_WS_FAULT and poisons its outputs with NaN.  The consumer loop then repeats the iteration ONCE
    # on the launch-per-step recurrent kernels (Engine.use_safe_recurrent) instead of dying of a ValueError on every batch.  A timeout
    # cannot be provoked on a free GPU, so the record and the NaN status are planted here - what is tested is the recovery path.
    from dotaclient_amd.engine import DC_DIMS_RNN_PER_STEP, DC_FAULT_TEAM_TIMEOUT
    g, _ = util.load_case('ragged_s16')
    stream = synth.make_rollouts(33, [40, 64, 21, 33, 50, 16, 64, 48, 17, 80, 30, 64, 25, 70, 41, 64, 18, 52])
    opt = make_opt([dict(r) for r in stream], g, tmp_path)
    opt.min_seq_per_epoch = 9
    opt.pipeline_rollout_pass = False
    m1 = opt.run_iteration(1)                                             # builds the workspace; default kernels
    assert opt.engine.fault() is None and not (opt.engine.kernel_flags & DC_DIMS_RNN_PER_STEP)
    opt.engine._ws[:32].view(torch.int32).copy_(torch.tensor([DC_FAULT_TEAM_TIMEOUT + 1, 0, 3, 1, 7, 12, 8, 0], dtype=torch.int32))
    opt.engine.status.fill_(1)                                            # what the poisoned launch leads to (sticky: no update happens)
    before = opt.engine.params.clone()
    products = opt.engine.products
    m2 = opt.run_iteration(2)
    assert opt.engine.kernel_flags & DC_DIMS_RNN_PER_STEP and opt.engine.fault() is None and int(opt.engine.status.item()) == 0
    assert opt.engine.products == products                                # the timeout was the cause on record: the products stay
    assert np.isfinite(float(m2['loss/sum'])) and not torch.equal(before, opt.engine.params)
    # same stream, per-step kernels from the start: the second iteration agrees (f32 rounding of a different kernel aside)
    ref = make_opt([dict(r) for r in stream], g, tmp_path)
    ref.min_seq_per_epoch = 9
    ref.pipeline_rollout_pass = False
    ref.run_iteration(1)
    ref.engine.kernel_flags |= DC_DIMS_RNN_PER_STEP
    r2 = ref.run_iteration(2)
    assert abs(float(m2['loss/sum']) - float(r2['loss/sum'])) <= 1e-4 * max(1.0, abs(float(r2['loss/sum'])))
    # a further NaN without a record: the products are what is left to try; after that it raises like the reference
    opt.engine.status.fill_(1)
    assert np.isfinite(float(opt.run_iteration(3)['loss/sum'])) and opt.engine.products == 'bf16x3'
    opt.engine.status.fill_(1)
    with pytest.raises(ValueError):
        opt.run_iteration(4)


def test_incremental_packer_on_the_gpu_equals_pack_rollouts():
    # the GPU form of tests/test_host_logic.py::test_incremental_packer_equals_pack_rollouts: page-locked staging, the H2D copies going
    # out in pieces WHILE the batch is packed (IncrementalPacker.COPY_ROWS) - also when the staging set (and with it the device
    # buffers) has to grow mid-batch after copies went out, when a batch is abandoned half-way, and when the sets are reused
    from dotaclient_amd.engine import IncrementalPacker, pack_rollouts
    dev = torch.device('cuda:0')
    pk = IncrementalPacker(16, dev, expected_rows=32)
    pk.COPY_ROWS = 64                                # several flushes per batch at these sizes
    for seed, lens in [(5, [40, 64, 7, 300, 16]), (6, [16]), (7, [500, 3, 3, 90]), (8, [70, 70, 70, 70, 1200, 33, 16])]:
        rollouts = synth.make_rollouts(seed, lens)
        want = pack_rollouts(rollouts, 16, dev)
        if seed == 7:                                # an abandoned batch: rows packed and copied, then dropped
            pk.add(rollouts[0]); pk.add(rollouts[3]); pk._begin()
        for d in rollouts:
            pk.add(d)
        got = pk.finish()
        torch.cuda.current_stream().wait_event(got.ready)
        assert got.rows == want.rows and got.max_len == want.max_len
        for k in ('obs', 'act', 'mask', 'rew', 'seq_off', 'seq_len'):
            assert torch.equal(getattr(got, k), getattr(want, k)), (seed, k)
        # the chunk view's layout tables come with the batch (host-built); pack_rollouts' batch builds them with torch kernels
        assert 16 in got._chunk_meta and 16 not in want._chunk_meta
        cg, cw = got.as_chunks(16), want.as_chunks(16)
        for k in ('seq_off', 'seq_len', 'is_first', 'prev_row'):
            a, b = getattr(cg, k), getattr(cw, k)
            assert a.dtype == b.dtype and torch.equal(a, b), (seed, k)
