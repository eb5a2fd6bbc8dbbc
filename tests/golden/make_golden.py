"""Generates tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Run only where /root/reference exists (the build container):   python tests/golden/make_golden.py
Nothing here is imported by tests or the product; the tests read the .npz files it wrote.

Recipe: stub the reference's missing third-party imports, build
`DotaOptimizer` with `__new__` (its __init__ needs RabbitMQ/GCS, optimizer.py:278-284), feed
torch.bool masks/actions (uint8 masks stopped working in torch>=1.2), load the deterministic weights
of dotaclient_amd.synth.init_state_dict, then call the reference's own
`experiences_from_rollout` (optimizer.py:328-430) and `train` (optimizer.py:581-689).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = '/root/reference'


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    stub('google'); stub('google.cloud'); stub('google.cloud.storage')
    sys.modules['google'].cloud = sys.modules['google.cloud']
    sys.modules['google.cloud'].storage = sys.modules['google.cloud.storage']
    stub('tensorboardX', SummaryWriter=object)
    stub('pika')
    stub('dotaservice'); stub('dotaservice.protos')
    stub('dotaservice.protos.DotaService_pb2', TEAM_DIRE=3, TEAM_RADIANT=2)
    sys.path.insert(0, REF)
    import optimizer as ref_optimizer   # noqa
    import policy as ref_policy         # noqa
    return ref_optimizer, ref_policy


from dotaclient_amd import synth, layout as L   # noqa: E402

SAMPLE_STRIDE = 251


def boolify(data):
    d = dict(data)
    d['masks'] = {k: v.bool() for k, v in data['masks'].items()}
    d['actions'] = {k: v.bool() for k, v in data['actions'].items()}
    return d


def tensor_summary(t):
    f = t.detach().double().flatten()
    return np.array([f.sum().item(), f.abs().sum().item(), f.norm().item()]), \
        t.detach().flatten()[::SAMPLE_STRIDE][:1000].float().numpy().copy()


def masked_argmax(ref_policy, policy, experiences):
    pol = ref_policy.Policy
    with torch.no_grad():
        obs = {k: torch.stack([e.observations[k] for e in experiences]) for k in pol.INPUT_KEYS}
        hidden = torch.cat([e.hidden for e in experiences], dim=1)
        logits, _, _ = policy(**obs, hidden=hidden)
        cols = []
        for k in pol.OUTPUT_KEYS:
            m = torch.stack([e.masks[k] for e in experiences])
            lp = pol.masked_softmax(logits=logits[k], mask=m)
            lp = torch.where(m, lp, torch.full_like(lp, -float('inf')))
            idx = lp.argmax(dim=-1)
            idx[~m.any(dim=-1)] = -1
            cols.append(idx)
    return torch.stack(cols, dim=-1).numpy().astype(np.int16)


def run_case(ref_optimizer, ref_policy, name, lengths, seq_len, epochs, lr, entropy_coef, vf_coef,
             data_seed, forbid_enum=()):
    torch.manual_seed(0)
    opt = ref_optimizer.DotaOptimizer.__new__(ref_optimizer.DotaOptimizer)
    opt.policy_base = ref_policy.Policy()
    opt.policy_base.load_state_dict(synth.init_state_dict(seed=7), strict=True)
    opt.policy = opt.policy_base
    opt.seq_len = seq_len
    opt.e_clip = 0.1
    opt.entropy_coef = entropy_coef
    opt.vf_coef = vf_coef
    opt.optimizer = torch.optim.Adam(opt.policy.parameters(), lr=lr)

    rollouts = synth.make_rollouts(data_seed, lengths, forbid_enum=forbid_enum)
    out = {'lengths': np.array(lengths), 'seq_len': np.array(seq_len), 'epochs': np.array(epochs),
           'lr': np.array(lr), 'entropy_coef': np.array(entropy_coef), 'vf_coef': np.array(vf_coef),
           'data_seed': np.array(data_seed), 'forbid_enum': np.array(list(forbid_enum), dtype=np.int64)}
    experiences = []
    with torch.no_grad():
        for r in rollouts:
            experiences.extend(opt.experiences_from_rollout(data=boolify(r)))
    out['advantages'] = torch.stack([e.advantages for e in experiences]).numpy()
    out['returns'] = torch.stack([e.returns for e in experiences]).numpy()
    out['values'] = torch.cat([e.values for e in experiences], dim=0).squeeze(-1).numpy()
    out['hidden'] = torch.cat([e.hidden for e in experiences], dim=1)[0].numpy()
    for k in ref_policy.Policy.OUTPUT_KEYS:
        out['old_logp_' + k] = torch.cat([e.log_probs_sel[k] for e in experiences]).numpy()
    out['argmax'] = masked_argmax(ref_policy, opt.policy, experiences)

    names = [n for n, _ in opt.policy.named_parameters()]
    for ep in range(epochs):
        losses, entropies, norms = opt.train(experiences=experiences)
        out['ep%d_losses' % ep] = np.array([float(losses[k]) for k in
                                            ('loss', 'policy_loss', 'entropy_loss', 'value_loss')], np.float64)
        out['ep%d_entropies' % ep] = np.array([float(entropies[k]) for k in ref_policy.Policy.OUTPUT_KEYS], np.float64)
        out['ep%d_grad_norms' % ep] = np.array([float(norms['unclipped']), float(norms['clipped'])], np.float64)
        gs, gv, ps, pv = [], [], [], []
        has_grad = []
        for n, p in opt.policy.named_parameters():
            has_grad.append(p.grad is not None)
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            s, v = tensor_summary(g); gs.append(s); gv.append(v)
            s, v = tensor_summary(p); ps.append(s); pv.append(v)
        out['ep%d_grad_summary' % ep] = np.stack(gs)          # clipped grads, [34,3]
        out['ep%d_grad_samples' % ep] = np.concatenate(gv)
        out['ep%d_param_summary' % ep] = np.stack(ps)
        out['ep%d_param_samples' % ep] = np.concatenate(pv)
        out['ep%d_has_grad' % ep] = np.array(has_grad)
    out['param_names'] = np.array(names)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(name, 'B=%d' % len(experiences), {k: v for k, v in out.items() if k.startswith('ep0_l')},
          '%.1f KB' % (os.path.getsize(path) / 1024))


# ---- data parallel: the reference's own wrapper (distributed.py:16-79) under two gloo ranks ---------------------------
DP_SHARDS = [dict(lengths=[48, 64], data_seed=131, forbid_enum=()),
             dict(lengths=[40, 32], data_seed=132, forbid_enum=(3,))]      # rank 1 never uses the ability head
DP_CFG = dict(seq_len=16, epochs=2, lr=1e-3, entropy_coef=5e-4, vf_coef=0.5)


def _dp_worker(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ref_optimizer, ref_policy = import_reference()
    import distributed as ref_distributed
    torch.set_num_threads(4)
    torch.manual_seed(0)
    opt = ref_optimizer.DotaOptimizer.__new__(ref_optimizer.DotaOptimizer)
    opt.policy_base = ref_policy.Policy()
    sd = synth.init_state_dict(seed=7)
    if rank != 0:                     # sync_parameters (distributed.py:71-74) must overwrite this with rank 0's weights
        sd = {k: v + 0.01 for k, v in sd.items()}
    opt.policy_base.load_state_dict(sd, strict=True)
    wrapper = ref_distributed.DistributedDataParallelSparseParamCPU(opt.policy_base)   # optimizer.py:269-270
    opt.seq_len, opt.e_clip = DP_CFG['seq_len'], 0.1
    opt.entropy_coef, opt.vf_coef = DP_CFG['entropy_coef'], DP_CFG['vf_coef']
    opt.optimizer = torch.optim.Adam(wrapper.parameters(), lr=DP_CFG['lr'])
    sh = DP_SHARDS[rank]
    rollouts = synth.make_rollouts(sh['data_seed'], sh['lengths'], forbid_enum=sh['forbid_enum'])
    # the rollout pass goes to policy_base: the wrapper has no init_hidden / sequence (the reference's own bug at
    # optimizer.py:340,385 - SURVEY.md 8(c))
    opt.policy = opt.policy_base
    experiences = []
    with torch.no_grad():
        for r in rollouts:
            experiences.extend(opt.experiences_from_rollout(data=boolify(r)))
    opt.policy = wrapper
    out = {'advantages': torch.stack([e.advantages for e in experiences]).numpy(),
           'returns': torch.stack([e.returns for e in experiences]).numpy()}
    for ep in range(DP_CFG['epochs']):
        losses, entropies, norms = opt.train(experiences=experiences)
        out['ep%d_losses' % ep] = np.array([float(losses[k]) for k in ('loss', 'policy_loss', 'entropy_loss', 'value_loss')], np.float64)
        out['ep%d_entropies' % ep] = np.array([float(entropies[k]) for k in ref_policy.Policy.OUTPUT_KEYS], np.float64)
        out['ep%d_grad_norms' % ep] = np.array([float(norms['unclipped']), float(norms['clipped'])], np.float64)
        gs, gv, ps, pv, hg = [], [], [], [], []
        for n, p in opt.policy_base.named_parameters():
            hg.append(p.grad is not None)
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            s_, v_ = tensor_summary(g); gs.append(s_); gv.append(v_)
            s_, v_ = tensor_summary(p); ps.append(s_); pv.append(v_)
        out['ep%d_grad_summary' % ep] = np.stack(gs)
        out['ep%d_grad_samples' % ep] = np.concatenate(gv)
        out['ep%d_param_summary' % ep] = np.stack(ps)
        out['ep%d_param_samples' % ep] = np.concatenate(pv)
        out['ep%d_has_grad' % ep] = np.array(hg)
    out['param_names'] = np.array([n for n, _ in opt.policy_base.named_parameters()])
    np.savez(os.path.join(tmp, 'rank%d.npz' % rank), **out)
    dist.destroy_process_group()


def run_dp_case(name):
    import tempfile
    import torch.multiprocessing as mp
    world = len(DP_SHARDS)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_dp_worker, args=(world, 29400 + os.getpid() % 500, tmp), nprocs=world, join=True)
        out = {'world': np.array(world), 'seq_len': np.array(DP_CFG['seq_len']), 'epochs': np.array(DP_CFG['epochs']),
               'lr': np.array(DP_CFG['lr']), 'entropy_coef': np.array(DP_CFG['entropy_coef']), 'vf_coef': np.array(DP_CFG['vf_coef'])}
        for r in range(world):
            sh = DP_SHARDS[r]
            out['r%d_lengths' % r] = np.array(sh['lengths'])
            out['r%d_data_seed' % r] = np.array(sh['data_seed'])
            out['r%d_forbid_enum' % r] = np.array(list(sh['forbid_enum']), dtype=np.int64)
            for k, v in np.load(os.path.join(tmp, 'rank%d.npz' % r)).items():
                out['r%d_%s' % (r, k)] = v
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(name, {k: out[k] for k in ('r0_ep0_losses', 'r1_ep0_losses', 'r0_ep0_has_grad', 'r1_ep0_has_grad')},
          '%.1f KB' % (os.path.getsize(path) / 1024))


def gae_long_fixture(ref_optimizer):
    import hashlib
    from tests.util import gae_long_inputs        # the tests regenerate the same inputs
    r, v, x = gae_long_inputs()
    out = {}
    for tag, rr, vv in (('zero_terminal', np.concatenate([r[:-1], [0]]).astype(np.float32), np.concatenate([v[:-1], [0]]).astype(np.float32)),
                        ('any_terminal', r, v)):
        adv, ret = ref_optimizer.advantage_returns(rr, vv, 0.98, 0.97)
        adv, ret = np.ascontiguousarray(adv, np.float32), np.ascontiguousarray(ret, np.float32)
        out[tag + '_adv_sha256'] = np.frombuffer(hashlib.sha256(adv.tobytes()).digest(), np.uint8)
        out[tag + '_ret_sha256'] = np.frombuffer(hashlib.sha256(ret.tobytes()).digest(), np.uint8)
        out[tag + '_adv_samples'], out[tag + '_ret_samples'] = adv[::997], ret[::997]
    d = np.ascontiguousarray(ref_optimizer.discount(x, 0.98 * 0.97), np.float32)
    out['discount_sha256'] = np.frombuffer(hashlib.sha256(d.tobytes()).digest(), np.uint8)
    out['discount_samples'] = d[::997]
    np.savez_compressed(os.path.join(HERE, 'gae_long.npz'), n=np.int64(50000), seed=np.int64(50), **out)
    print('gae_long', out['zero_terminal_adv_samples'][:3], '%.1f KB' % (os.path.getsize(os.path.join(HERE, 'gae_long.npz')) / 1024))


def main():
    ref_optimizer, ref_policy = import_reference()
    torch.set_num_threads(8)
    # known-answer vector for advantage_returns
    adv, ret = ref_optimizer.advantage_returns(np.array([1, 2, 3, 0], np.float32),
                                               np.array([.5, .4, .3, 0], np.float32), 0.98, 0.97)
    rng = np.random.Generator(np.random.PCG64(5))
    r = (0.3 * rng.standard_normal(1025)).astype(np.float32); r[-1] = 0
    v = rng.standard_normal(1025).astype(np.float32); v[-1] = 0
    adv2, ret2 = ref_optimizer.advantage_returns(r, v, 0.98, 0.97)
    # non-zero terminal reward / bootstrap value (the function accepts any (L+1)-vectors, optimizer.py:57-64) and
    # `discount` on its own (optimizer.py:53-54)
    r3 = (0.3 * rng.standard_normal(778)).astype(np.float32)
    v3 = rng.standard_normal(778).astype(np.float32)
    adv3, ret3 = ref_optimizer.advantage_returns(r3, v3, 0.98, 0.97)
    x4 = rng.standard_normal(3001).astype(np.float32)
    disc4 = ref_optimizer.discount(x4, 0.98).astype(np.float32)
    disc4b = ref_optimizer.discount(x4[:65], 0.98 * 0.97).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'gae_kat.npz'), adv=adv, ret=ret, r2=r, v2=v, adv2=adv2, ret2=ret2,
                        r3=r3, v3=v3, adv3=adv3, ret3=ret3, x4=x4, disc4=disc4, disc4b=disc4b)
    print('gae_kat', adv, ret)
    # rollouts longer than one LDS block of the HIP scan (20 480 steps): 50 000 steps through the real
    # reference.  Inputs are regenerated from the seed by the tests; the fixture holds a SHA-256 of the reference's full float32
    # outputs (bit-exactness over all 50 000 entries from 64 bytes) plus every 997th value for a readable diff.
    gae_long_fixture(ref_optimizer)

    common = dict(entropy_coef=5e-4, vf_coef=0.5)
    run_case(ref_optimizer, ref_policy, 'ragged_s16', [50, 64, 33], 16, 2, 5e-5, data_seed=123, **common)
    run_case(ref_optimizer, ref_policy, 'clip_s16', [48, 64, 40, 16], 16, 3, 3e-3, data_seed=124, **common)
    run_case(ref_optimizer, ref_policy, 'cfg1_4x128', [128, 128, 128, 128], 128, 1, 5e-5, data_seed=125, **common)
    run_case(ref_optimizer, ref_policy, 'emptyhead_s16', [40, 32], 16, 2, 5e-5, data_seed=126,
             forbid_enum=(3,), **common)
    # the loss-coefficient branches of optimizer.py:652-663 (entropy_coef == 0 / vf_coef == 0: the term is a constant 0
    # and - for the value head - affine_value gets no gradient at all)
    run_case(ref_optimizer, ref_policy, 'noent_s16', [50, 64, 33], 16, 2, 5e-5, data_seed=123, entropy_coef=0.0, vf_coef=0.5)
    run_case(ref_optimizer, ref_policy, 'novf_s16', [50, 64, 33], 16, 2, 5e-5, data_seed=123, entropy_coef=5e-4, vf_coef=0.0)
    # BASELINE.json configs[1]'s batch (64 trajectories x 256 steps) on the reference's own network (GRU-256): the real
    # reference at a bench-sized batch, one epoch
    run_case(ref_optimizer, ref_policy, 'cfg2_gru_64x256', [256] * 64, 256, 1, 5e-5, data_seed=1000, **common)
    run_dp_case('dp2_s16')


if __name__ == '__main__':
    main()
