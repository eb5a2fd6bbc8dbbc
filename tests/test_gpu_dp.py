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
