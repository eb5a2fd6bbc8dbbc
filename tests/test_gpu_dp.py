"""GPU: the data-parallel step (/root/reference/distributed.py:16-79) on the real HIP path.

  * dc_dp_average_grads on real device buffers against the oracle's distributed.py:24-57 emulation;
  * two ranks, each a REAL Engine, against the fixture produced by the reference's own
    DistributedDataParallelSparseParamCPU under two gloo ranks (tests/golden/dp2_s16.npz).  On a one-GPU box the two
    ranks share cuda:0 and talk gloo (RCCL refuses two ranks on one device); with >= 2 GPUs the same test also runs
    over RCCL, one device per rank - the path bench.py takes under torch.distributed.run.
"""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dotaclient_amd import layout as L
from dotaclient_amd import synth
from oracle import ref_optimizer as RO
from tests import util

pytestmark = pytest.mark.gpu


def test_dp_average_grads_kernel_matches_oracle():
    from dotaclient_amd import _lib
    from dotaclient_amd.engine import Engine
    dev = torch.device('cuda:0')
    eng = Engine('gru', 256, 1, dev)
    names = eng.seg_names
    world = 3
    # which rank's heads acted: head 4 (ability) on rank 0 only, head 3 (target_unit) on ranks 0 and 2, head 1 on none
    head_on = np.array([[1, 0, 1, 1, 1], [1, 0, 1, 0, 0], [1, 0, 1, 1, 0]], np.float32)
    gen = torch.Generator().manual_seed(5)
    local = [torch.randn(eng.total, generator=gen) for _ in range(world)]
    per_rank = []
    for r in range(world):
        gs = []
        for n, gate in zip(names, eng.seg_gate_host):
            o, ln, _ = eng.layout[n]
            has = gate < 0 or gate == 5 or head_on[r, gate] != 0
            if not has:
                local[r][o:o + ln] = 0                      # a head that never acted leaves zeros in the flat buffer
            gs.append(local[r][o:o + ln].clone() if has else None)
        per_rank.append(gs)
    want = RO.dp_average_grads(per_rank)
    # what the SUM all-reduce of [flat grads | head flags | 1] leaves on every rank
    summed = torch.stack(local).sum(0).to(dev)
    counts = torch.tensor(list(head_on.sum(0)) + [world, 0, 0], dtype=torch.float32, device=dev)
    _lib.check(eng.lib.dc_dp_average_grads(_lib.ptr(eng.seg_off), _lib.ptr(eng.seg_len), _lib.ptr(eng.seg_gate), len(names),
                                           eng.max_seg_len, _lib.ptr(summed), _lib.ptr(counts), 0.5, _lib.stream_ptr()),
               'dc_dp_average_grads')
    got = summed.cpu()
    for j, n in enumerate(names):
        o, ln, _ = eng.layout[n]
        ref = next((want[r][j] for r in range(world) if want[r][j] is not None), None)
        if ref is None:                                       # nobody had a gradient: the zeros stay zeros
            assert torch.all(got[o:o + ln] == 0), n
        else:
            assert torch.allclose(got[o:o + ln], ref, rtol=1e-6, atol=1e-7), n


def _dp_engine_worker(rank, world, port, tmp, backend, overlap):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev_index = rank if backend == 'nccl' else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from dotaclient_amd.distributed import FlatGradAllReducer
    from dotaclient_amd.engine import Engine, pack_rollouts
    g, shards = util.load_dp_case()
    S = int(g['seq_len'])
    eng = Engine('gru', 256, 1, dev)
    sd = synth.init_state_dict(7)
    if rank != 0:                                             # must be overwritten by rank 0's weights (distributed.py:71-74)
        sd = {k: v + 0.01 for k, v in sd.items()}
    eng.load_state_dict(sd)
    hook = FlatGradAllReducer(eng, overlap=overlap)
    hook.sync_parameters()
    batch = pack_rollouts(shards[rank], S, dev)
    chunks = eng.rollout_pass(batch, S)
    B = chunks.n_seq
    out = {'advantages': batch.adv.view(B, S).cpu().numpy(), 'returns': batch.ret.view(B, S).cpu().numpy()}
    names = list(L.param_shapes().keys())
    assert names == eng.seg_names
    prev_steps = np.zeros(len(names), np.int64)
    for ep in range(int(g['epochs'])):
        res, status = eng.train_epoch(chunks, float(g['lr']), float(g['entropy_coef']), float(g['vf_coef']), grad_hook=hook)
        assert int(status.item()) == 0
        r = res.cpu().numpy().astype(np.float64)
        out['ep%d_losses' % ep], out['ep%d_entropies' % ep], out['ep%d_grad_norms' % ep] = r[0:4], r[4:9], r[9:11]
        steps = eng.seg_step.cpu().numpy().astype(np.int64)
        stepped = steps > prev_steps
        prev_steps = steps
        # a parameter this rank has no gradient for (grad None in the reference: the reduced value is discarded there,
        # distributed.py:50-57; here the flat bucket holds the other ranks' average, which Adam skips) compares as zeros
        gs = [util.tensor_summary(eng.param_view(n, eng.grads))[1] for n in names]
        out['ep%d_grad_samples' % ep] = np.concatenate([x if on else np.zeros_like(x) for x, on in zip(gs, stepped)])
        out['ep%d_param_samples' % ep] = np.concatenate([util.tensor_summary(eng.param_view(n))[1] for n in names])
        out['ep%d_steps' % ep] = steps
    np.savez(os.path.join(tmp, 'rank%d.npz' % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(tmp_path, backend, overlap):
    world = 2
    port = 29600 + (os.getpid() * 7 + int(overlap) + (3 if backend == 'nccl' else 0)) % 1500
    mp.spawn(_dp_engine_worker, args=(world, port, str(tmp_path), backend, overlap), nprocs=world, join=True)
    g, _ = util.load_dp_case()
    tol = 1e-4
    for r in range(world):
        out = np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r))
        for key in ('advantages', 'returns'):
            assert util.scaled_err(out[key], g['r%d_%s' % (r, key)]) < tol, (r, key)
        for ep in range(int(g['epochs'])):
            pre = 'r%d_ep%d_' % (r, ep)
            for key in ('losses', 'entropies', 'grad_norms'):
                a, b = out['ep%d_%s' % (ep, key)], g[pre + key]
                err = util.loss_rel_err(a, b) if key == 'losses' else util.rel_err(a, b)
                assert err < tol, (pre + key, a, b)
            # has-grad pattern: rank 1 never used the ability head -> no gradient there -> not stepped (distributed.py:50-57)
            assert np.array_equal(out['ep%d_steps' % ep] > 0, g[pre + 'has_grad']), pre
            # averaged + clipped gradients (zeros where the reference has grad None) and post-step parameters
            assert util.scaled_err(out['ep%d_grad_samples' % ep], g[pre + 'grad_samples']) < 5 * tol, pre
            assert util.scaled_err(out['ep%d_param_samples' % ep], g[pre + 'param_samples']) < tol, pre
    assert not g['r1_ep0_has_grad'].all()


@pytest.mark.parametrize('overlap', [False, True])
def test_two_rank_engines_match_reference_wrapper_gloo_one_device(tmp_path, overlap):
    _run_two_ranks(tmp_path, 'gloo', overlap)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one device per rank')
@pytest.mark.parametrize('overlap', [False, True])
def test_two_rank_engines_match_reference_wrapper_rccl(tmp_path, overlap):
    _run_two_ranks(tmp_path, 'nccl', overlap)


def test_overlapped_reduce_call_order_and_stream_ordering(monkeypatch):
    # the overlapped data-parallel epoch (Engine.train_epoch with a FlatGradAllReducer, dotaclient_amd/distributed.py:53-68)
    # has only ever met gloo, where "async" is synchronous.  What must hold on RCCL - checked here with a recording stand-in for
    # torch.distributed on ONE process, real engine, real kernels:
    #   1. order of the calls: backward(UPPER) -> async all-reduce of bucket[embed_floats:] (incl. the head flags) -> backward(EMBED) ->
    #      all-reduce of bucket[:embed_floats] -> wait() of the first -> average kernel -> Adam;
    #   2. stream ordering: at the moment the asynchronous all-reduce is CALLED, everything that writes its slice (the upper backward)
    #      is already enqueued on the calling stream - RCCL's collective stream waits on exactly that - and nothing that writes the slice
    #      is enqueued between the call and its wait(): an event recorded at the call is complete before the slice's checksum, taken on
    #      the stream at wait() time, is read back, and that checksum equals the one taken right after the upper backward;
    #   3. the two slices are disjoint and cover the bucket.
    from dotaclient_amd import distributed as D
    from dotaclient_amd import engine as E
    dev = torch.device('cuda:0')
    log = []

    class Work:
        def __init__(self, t):
            self.t = t
        def wait(self):
            log.append(('wait', float(self.t.double().sum().item())))
            return True

    class FakeDist:
        ReduceOp = dist.ReduceOp
        def is_available(self): return True
        def is_initialized(self): return True
        def get_world_size(self, group=None): return 2
        def broadcast(self, t, src=0, group=None): log.append(('broadcast', t.numel()))
        def all_reduce(self, t, op=None, group=None, async_op=False):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            log.append(('all_reduce', t.data_ptr(), t.numel(), bool(async_op), float(t.double().sum().item()), ev))
            t.mul_(2.0)                                                   # "sum over two identical ranks"
            return Work(t) if async_op else None

    monkeypatch.setattr(D, 'dist', FakeDist())
    eng = E.Engine('gru', 256, 1, dev)
    eng.load_state_dict(synth.init_state_dict(7))
    hook = D.FlatGradAllReducer(eng, overlap=True)
    hook.sync_parameters()
    orig_backward = eng.backward
    eng.backward = lambda d, b, part=0: (log.append(('backward', part)), orig_backward(d, b, part))[1]
    orig_avg = hook._average
    hook._average = lambda e: (log.append(('average',)), orig_avg(e))[1]
    orig_adam = eng.adam
    eng.adam = lambda lr, vf: (log.append(('adam',)), orig_adam(lr, vf))[1]
    rollouts = synth.make_rollouts(3, [48, 64, 32])
    chunks = eng.rollout_pass(E.pack_rollouts(rollouts, 16, dev), 16)
    eng.train_epoch(chunks, 1e-4, 5e-4, 0.5, grad_hook=hook)
    torch.cuda.synchronize()
    kinds = [x[0] for x in log]
    assert kinds == ['broadcast', 'backward', 'all_reduce', 'backward', 'all_reduce', 'wait', 'average', 'adam'], kinds
    (_, part0), ar0, (_, part1), ar1, wait = log[1], log[2], log[3], log[4], log[5]
    assert part0 == E.DC_DIMS_BWD_UPPER and part1 == E.DC_DIMS_BWD_EMBED
    base = hook.bucket.data_ptr()
    assert ar0[3] is True and ar0[1] == base + 4 * eng.embed_floats and ar0[2] == eng.total + 8 - eng.embed_floats     # upper slice + flags, async
    assert ar1[3] is False and ar1[1] == base and ar1[2] == eng.embed_floats                                            # embedding slice, blocking
    assert ar0[5].query()                                                  # the event behind the upper backward has completed ...
    assert abs(wait[1] - 2.0 * ar0[4]) <= 1e-9 * abs(ar0[4]) and ar0[4] != 0.0     # ... and nothing touched that slice between the call and its wait()
    assert int(eng.status.item()) == 0 and torch.isfinite(eng.params).all()
    # the non-overlapped form: one collective over the whole bucket
    log.clear()
    hook2 = D.FlatGradAllReducer(eng, overlap=False)
    eng.backward, eng.adam = orig_backward, orig_adam
    eng.train_epoch(chunks, 1e-4, 5e-4, 0.5, grad_hook=hook2)
    ars = [x for x in log if x[0] == 'all_reduce']
    assert len(ars) == 1 and ars[0][2] == eng.total + 8 and ars[0][3] is False


def test_bench_eight_ranks_on_one_device_prints_one_dp8_line():
    # no multi-GPU box was offered in any round, so the driver's N = 8 command has never run.  A dry run of exactly that
    # flow on ONE GPU: `python bench.py --gpus 8` starts its own eight ranks (torch.distributed.run), every rank on cuda:0 over gloo
    # (DC_BENCH_ONE_DEVICE=1 - RCCL refuses two ranks on one device), each with configs[3]'s per-GPU shard (128 trajectories x 256 steps): the
    # broadcast, eight real engines, the flat-bucket all-reduce every epoch, the barriers, the max over ranks and rank 0's ONE JSON line.
    # Not a performance number (eight ranks time-share the GPU) - the line's keys and the workload it names are what is checked.
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(DC_BENCH_ONE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '8', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-weak-unit'],
                       cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, (len(lines), r.stdout[-2000:])
    j = json.loads(lines[0])
    assert j['n_gpus'] == 8 and j['steps'] == 1 and j['scaling'] == 'weak' and j['higher_is_better'] is True
    assert j['config']['parallelism'] == 'dp8' and 'configs[3]' in j['config']['workload'] and j['config']['batch_per_gpu'] == 128
    assert j['unit'] == 'env-steps/s' and j['value'] > 0 and abs(j['value'] - 8 * 128 * 256 / (j['ms_per_step'] * 1e-3)) < 1e-3 * j['value']
    assert j['nan_status'] == 0 and np.isfinite(j['final_loss'])
