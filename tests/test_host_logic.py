"""CPU: host-side logic that needs no GPU - packing, layout, the DP bucket protocol over gloo."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dotaclient_amd import layout as L
from dotaclient_amd import synth
from oracle import ref_optimizer as RO


def test_flat_layout_is_dense_for_the_head_block_and_aligned():
    for cell, h, layers in [('gru', 256, 1), ('lstm', 128, 1), ('lstm', 512, 2)]:
        lay, total = L.flat_layout(cell, h, layers)
        shapes = L.param_shapes(cell, h, layers)
        assert set(lay) == set(shapes)
        spans = sorted((o, o + n) for o, n, _ in lay.values())
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 <= b0                       # no overlap
        hw = ['affine_unit_attention', 'affine_head_enum', 'affine_move_x', 'affine_move_y', 'affine_head_ability', 'affine_value']
        for suffix in ('.weight', '.bias'):
            off = lay[hw[0] + suffix][0]
            assert off % 4 == 0
            for n in hw:
                assert lay[n + suffix][0] == off   # contiguous [154,H] / [154] block
                off += lay[n + suffix][1]
        ref_total = {('gru', 256, 1): 765210, ('lstm', 128, 1): 548378, ('lstm', 512, 2): 4088090}[(cell, h, layers)]
        assert sum(n for _, n, _ in lay.values()) == ref_total      # SURVEY.md 8(d) parameter counts


def test_flatten_rollout_layout_and_actor_invariants():
    r = synth.make_rollout(3, 37)
    obs, act, msk, rew = synth.flatten_rollout(r)
    assert obs.shape == (37, 483) and act.shape == (37, 65) and msk.shape == (37, 65) and rew.shape == (37, 10)
    assert np.array_equal(obs[:, 3 + 12 * 6:3 + 12 * 7], r['observations']['allied_nonheroes'][:, 0].numpy())
    assert np.array_equal(obs[:, 3 + 12 * 39:], r['observations']['enemy_towers'][:, 0].numpy())
    # exactly one enum action, sub-head actions only inside their masks, unselected heads fully masked out
    assert (act[:, :4].sum(1) == 1).all()
    assert ((act & ~msk.astype(bool)) == 0).all()
    e = act[:, :4].argmax(1)
    assert (act[e != 2, 22:62] == 0).all() and (msk[e != 2, 22:62] == 0).all()
    assert (act[e == 1, 4:13].sum(1) == 1).all() and (act[e == 1, 13:22].sum(1) == 1).all()


def test_pack_rollouts_pads_to_seq_len_multiples():
    from dotaclient_amd.engine import pack_rollouts
    rollouts = synth.make_rollouts(4, [50, 16, 33])
    b = pack_rollouts(rollouts, 16, torch.device('cpu'))          # packing itself is device-agnostic plumbing
    assert b.seq_len.tolist() == [64, 16, 48] and b.seq_off.tolist() == [0, 64, 80] and b.rows == 128
    assert torch.all(b.obs[50:64] == 0) and torch.all(b.mask[50:64] == 0) and torch.all(b.rew[113:128] == 0)
    c = b.as_chunks(16)
    assert c.n_seq == 8 and c.seq_off.tolist() == list(range(0, 128, 16))


def test_pack_rollouts_matches_flatten_and_reuses_staging():
    # the direct-to-staging ingest against the per-rollout flattening, twice (the staging buffers are reused, a shorter
    # rollout in the same slot must not inherit rows of the previous batch)
    from dotaclient_amd.engine import pack_rollouts
    for seed, lens in [(5, [40, 64, 7]), (6, [33, 50, 16]), (7, [40, 64, 7])]:
        rollouts = synth.make_rollouts(seed, lens)
        b = pack_rollouts(rollouts, 16, torch.device('cpu'))
        r0 = 0
        for d, T in zip(rollouts, lens):
            o, a, m, r = synth.flatten_rollout(d)
            lp = (T + 15) // 16 * 16
            assert np.array_equal(b.obs[r0:r0 + T].numpy(), o) and np.array_equal(b.act[r0:r0 + T].numpy(), a)
            assert np.array_equal(b.mask[r0:r0 + T].numpy(), m) and np.array_equal(b.rew[r0:r0 + T].numpy(), r)
            for x in (b.obs, b.act, b.mask, b.rew):
                assert torch.all(x[r0 + T:r0 + lp] == 0)
            r0 += lp
        assert b.rows == r0


def test_incremental_packer_equals_pack_rollouts():
    # the consumer loop's form (one rollout at a time, as they come off the queue) must build byte-for-byte the batch pack_rollouts
    # builds from the same rollouts - also when the staging set has to grow mid-batch and when it is reused for the next batch
    from dotaclient_amd.engine import IncrementalPacker, pack_rollouts
    pk = IncrementalPacker(16, torch.device('cpu'), expected_rows=32)
    for seed, lens in [(5, [40, 64, 7, 300, 16]), (6, [16]), (7, [500, 3, 3, 90])]:
        rollouts = synth.make_rollouts(seed, lens)
        want = pack_rollouts(rollouts, 16, torch.device('cpu'))
        for i, d in enumerate(rollouts):
            pk.add(d)
            assert len(pk) == i + 1 and pk.n_seq == sum((t + 15) // 16 for t in lens[:i + 1])
        got = pk.finish()
        assert len(pk) == 0 and got.max_len == want.max_len and got.rows == want.rows
        for k in ('obs', 'act', 'mask', 'rew', 'seq_off', 'seq_len'):
            assert torch.equal(getattr(got, k), getattr(want, k)), k
    with pytest.raises(ValueError):
        pk.finish()


def test_incremental_packer_drops_a_malformed_rollout_whole():
    # a rollout that fails validation half-way through its keys must leave NO copy descriptor queued (its rows are not
    # committed, so the next good rollout targets the same staging rows: a stale descriptor would race it in dc_pack_rows)
    import copy
    from dotaclient_amd.engine import IncrementalPacker, pack_rollouts
    good = synth.make_rollouts(11, [40, 64, 7, 300])
    bad = copy.deepcopy(synth.make_rollouts(12, [500])[0])       # longer than its successors: stale copies would spill into later rows
    bad['actions']['ability'] = bad['actions']['ability'][:-3]    # a late key: the observations and most heads have queued their copies by then
    worse = copy.deepcopy(synth.make_rollouts(13, [90])[0])
    worse['masks']['ability'] = worse['masks']['ability'][:10]
    pk = IncrementalPacker(16, torch.device('cpu'), expected_rows=32)
    pk.add(good[0])
    for b in (bad, worse):
        n_items, n_keep, rows = len(pk._items), len(pk._keep), pk.rows
        with pytest.raises(ValueError):
            pk.add(b)
        # (a rollout that needs a larger staging set packs what is pending before it is validated: then nothing is queued at all)
        assert (len(pk._items), len(pk._keep)) in ((n_items, n_keep), (0, 0)) and (pk.rows, len(pk)) == (rows, 1)
    for d in good[1:]:
        pk.add(d)
    got, want = pk.finish(), pack_rollouts(good, 16, torch.device('cpu'))
    for k in ('obs', 'act', 'mask', 'rew', 'seq_off', 'seq_len'):
        assert torch.equal(getattr(got, k), getattr(want, k)), k


def test_staging_is_one_growing_pair_not_one_per_batch_size():
    # the consumer loop's row count changes almost every iteration: the page-locked staging must not accumulate a pair of
    # buffers per distinct size - one double-buffered pair, grown geometrically, a batch uses its leading rows
    from dotaclient_amd import engine as E
    dev = torch.device('cpu')
    E._DEFAULT_STAGING.pop(False, None)
    caps = []
    for seed, lens in enumerate([[16], [48, 16], [32], [200, 40], [64], [16, 16, 16], [500], [16]]):
        rollouts = synth.make_rollouts(40 + seed, lens)
        b = E.pack_rollouts(rollouts, 16, dev)
        assert b.rows == sum((t + 15) // 16 * 16 for t in lens) and b.obs.shape[0] == b.rows
        o, a, m, r = synth.flatten_rollout(rollouts[0])
        assert np.array_equal(b.obs[:lens[0]].numpy(), o) and np.array_equal(b.mask[:lens[0]].numpy(), m)
        sets = E._DEFAULT_STAGING[False].sets
        assert len(sets) == 2
        caps.append(sorted(st.capacity for st in sets if st is not None))
    assert max(caps[-1]) >= 512 and len(E._DEFAULT_STAGING) == 1        # still exactly one pair
    assert all(c >= p for p, c in zip(caps[1:-1:2], caps[3::2]))          # capacities only ever grow


def test_gather_cache_is_keyed_on_content_not_on_list_identity():
    # a cache keyed on id(list) returns the PREVIOUS iteration's batch when CPython recycles the address of a
    # freed same-length list.  _gather's key is the (batch, chunk index) sequence, and the keyed batches are kept alive.
    from dotaclient_amd.optimizer import DotaOptimizer, Sequence
    from dotaclient_amd.engine import PackedBatch

    def fake_batch(fill):
        t = lambda w, dt=torch.float32: torch.full((32, w), fill, dtype=dt)
        b = PackedBatch(t(483), t(65, torch.uint8), t(65, torch.uint8), t(10), torch.tensor([0, 16]), torch.tensor([16, 16], dtype=torch.int32), 16)
        b.old_logp, b.values, b.adv, b.ret = t(5), t(1)[:, 0], t(1)[:, 0], t(1)[:, 0]
        return b

    opt = DotaOptimizer.__new__(DotaOptimizer)
    opt.seq_len, opt.device = 16, torch.device('cpu')
    h = torch.zeros(1, 1, 256)
    outs = []
    for fill in (1.0, 2.0, 3.0):                    # a fresh same-length list each "iteration", the old one freed
        b = fake_batch(fill)
        exps = [Sequence('g', 1, 2, b, 1, 16, h), Sequence('g', 1, 2, b, 0, 16, h)]     # reordered: takes the slow path
        outs.append(float(opt._gather(exps).obs[0, 0]))
        again = opt._gather(list(exps))              # same content, another list object: served from the cache
        assert again is opt._gather_val
        del exps, b
    assert outs == [1.0, 2.0, 3.0]


def test_pack_rollouts_accepts_what_the_wire_may_carry():
    # the actors send torch tensors (agent.py:406-416); numpy arrays, bool masks (the oracle's form), non-contiguous
    # views and float64 observations must come out the same bytes - dc_pack_rows only sees contiguous f32 / u8
    from dotaclient_amd.engine import pack_rollouts
    base = synth.make_rollouts(21, [40, 64, 7, 300])
    want = pack_rollouts(base, 16, torch.device('cpu'))
    odd = []
    for i, d in enumerate(base):
        e = {'observations': dict(d['observations']), 'actions': dict(d['actions']), 'masks': dict(d['masks']),
             'rewards': d['rewards']}
        if i == 0:
            e['masks'] = {k: v.bool() for k, v in d['masks'].items()}
            e['actions'] = {k: v.numpy().astype(bool) for k, v in d['actions'].items()}
        if i == 1:
            e['observations'] = {k: v.double() for k, v in d['observations'].items()}
            e['rewards'] = np.asfortranarray(d['rewards'])
        if i == 2:
            e['observations'] = {k: torch.stack([v, v], 1)[:, 0] for k, v in d['observations'].items()}     # strided views
        if i == 3:
            e['observations'] = {k: v.numpy() for k, v in d['observations'].items()}
            e['rewards'] = torch.from_numpy(np.ascontiguousarray(d['rewards']))
        odd.append(e)
    got = pack_rollouts(odd, 16, torch.device('cpu'))
    for a, b in ((want.obs, got.obs), (want.act, got.act), (want.mask, got.mask), (want.rew, got.rew)):
        assert torch.equal(a, b)
    bad = dict(base[0]); bad['rewards'] = base[0]['rewards'][:-1]       # a key with one step less than the others
    with pytest.raises(ValueError):
        pack_rollouts([bad], 16, torch.device('cpu'))


def test_dc_pack_rows_threads_and_bad_items():
    import ctypes
    from dotaclient_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    n, rows, w, stride = 200, 256, 192, 1932
    src = [rng.integers(0, 255, size=(rows, w), dtype=np.uint8) for _ in range(n)]
    dst = np.zeros((n, rows, stride), dtype=np.uint8)
    tab = np.zeros((5, n + 1), dtype=np.int64)
    for i in range(n):
        tab[:, i] = (src[i].ctypes.data, dst[i].ctypes.data + 7 * (i % 5), rows, w, stride)
    pad = np.full((64, 40), 9, dtype=np.uint8)
    tab[:, n] = (0, pad.ctypes.data, 64, 40, 40)                        # zero-fill item
    col = lambda k: ctypes.c_void_p(tab[k].ctypes.data)
    for threads in (1, 3, 8, 64):
        dst[:] = 0; pad[:] = 9
        assert lib.dc_pack_rows(col(0), col(1), col(2), col(3), col(4), n + 1, threads) == 0
        for i in range(0, n, 17):
            o = 7 * (i % 5)
            assert np.array_equal(dst[i].reshape(-1)[o:o + rows * stride - stride + w].reshape(-1)[:w], src[i][0])
            assert np.array_equal(np.lib.stride_tricks.as_strided(dst[i].reshape(-1)[o:], (rows, w), (stride, 1)), src[i])
        assert not pad.any()
    tab[4, 0] = w - 1                                                   # destination stride smaller than the row
    assert lib.dc_pack_rows(col(0), col(1), col(2), col(3), col(4), n + 1, 4) != 0
    assert b'dc_pack_rows' in lib.dc_last_error()


def test_pack_rollouts_rejects_an_empty_batch():
    from dotaclient_amd.engine import pack_rollouts
    with pytest.raises(ValueError):
        pack_rollouts([], 16, torch.device('cpu'))


def test_checkpointing_into_a_used_log_dir_without_a_model_is_refused(tmp_path):
    # the reference resumes from the newest model_*.pt of log_dir by itself (optimizer.py:243-253); this class leaves the scan to
    # the launcher, so it must not silently start at version 1 over an existing run (model files overwritten, versions going backwards to
    # the actors).  The check comes before any device work: no GPU needed.
    from dotaclient_amd.optimizer import DotaOptimizer
    kw = dict(rmq_host='x', rmq_port=0, epochs=1, min_seq_per_epoch=1, seq_len=16, learning_rate=1e-4, mq_prefetch_count=1,
              entropy_coef=5e-4, vf_coef=0.5, run_local=True)
    (tmp_path / 'model_000000041.pt').write_bytes(b'x')
    (tmp_path / 'model_000000007.pt').write_bytes(b'x')
    with pytest.raises(ValueError, match='model_000000041.pt'):
        DotaOptimizer(checkpoint=True, pretrained_model=None, log_dir=str(tmp_path), mq=object(), **kw)


def test_product_path_refuses_cpu():
    from dotaclient_amd import _lib
    from dotaclient_amd.engine import Engine
    with pytest.raises(_lib.DotaHipError):
        Engine('gru', 256, 1, 'cpu')


# ---- data parallel: flat bucket + has-grad counts over gloo, world_size 2 --------------------------------
class _FakeLib:
    """CPU stand-in for the scaling kernel only (the collective protocol is what is under test)."""

    def __init__(self, eng):
        self.eng = eng

    def dc_dp_average_grads(self, seg_off, seg_len, seg_gate, n_seg, max_len, grads, counts, vf, stream):
        e = self.eng
        c = e.reducer.tail
        for off, ln, gate in zip(e.seg_off.tolist(), e.seg_len.tolist(), e.seg_gate.tolist()):
            cnt = c[gate].item() if 0 <= gate < 5 else c[5].item()
            if cnt > 0:
                e.grads[off:off + ln] /= cnt
        return 0


class _FakeEngine:
    def params_changed(self):
        self.params_changed_calls = getattr(self, 'params_changed_calls', 0) + 1

    def __init__(self, head_on):
        lay, total = L.flat_layout()
        names = list(L.param_shapes().keys())
        self.total, self.device = total, torch.device('cpu')
        self.seg_names = names
        self.seg_off = torch.tensor([lay[n][0] for n in names])
        self.seg_len = torch.tensor([lay[n][1] for n in names], dtype=torch.int32)
        gate = []
        for n in names:
            g = -1
            for pre, k in (('affine_head_enum', 0), ('affine_move_x', 1), ('affine_move_y', 2),
                           ('affine_unit_attention', 3), ('affine_unit_eth', 3), ('affine_head_ability', 4), ('affine_value', 5)):
                if n.startswith(pre):
                    g = k
            gate.append(g)
        self.seg_gate = torch.tensor(gate, dtype=torch.int32)
        self.max_seg_len = int(self.seg_len.max())
        self.params = torch.zeros(total)
        self.grads = torch.zeros(total)
        self.head_on = torch.tensor(head_on + [0, 0, 0], dtype=torch.int32)
        self.lib = _FakeLib(self)
        self.layout = lay
        self.embed_floats = lay['affine_pre_rnn.weight'][0]


def _dp_worker(rank, world, port, tmp, overlap):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dotaclient_amd import distributed as D
    D._lib.ptr = lambda t: t                         # no device pointers on CPU
    D._lib.stream_ptr = lambda: None
    D._lib.check = lambda code, what='': None
    # rank 1 never used the ability head (index 4): its ability grads are "None"
    eng = _FakeEngine([1, 1, 1, 1, 1 if rank == 0 else 0])
    red = D.FlatGradAllReducer(eng, overlap=overlap)
    eng.reducer = red
    eng.params.fill_(float(rank + 1))
    red.sync_parameters()
    assert torch.all(eng.params == 1.0)              # broadcast from rank 0 (distributed.py:71-74)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(eng.total, generator=g)
    if rank == 1:
        for nm in ('affine_head_ability.weight', 'affine_head_ability.bias'):
            o, n, _ = eng.layout[nm]
            local[o:o + n] = 0                       # no gradient on this rank
    eng.grads.copy_(local)
    if overlap:                                       # the two-collective form Engine.train_epoch drives
        red.start_upper(eng)
        red.finish(eng)
    else:
        red(eng)
    torch.save({'local': local, 'out': eng.grads.clone()}, os.path.join(tmp, 'r%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_dp_flat_bucket_matches_reference_semantics(tmp_path, overlap):
    world, port = 2, 29500 + (os.getpid() + int(overlap)) % 2000
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path), overlap), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in range(world)]
    lay, total = L.flat_layout()
    names = list(L.param_shapes().keys())
    # oracle: distributed.py:24-57 emulation - average over the ranks that have a grad
    per_rank = []
    for r in range(world):
        gs = []
        for n in names:
            o, ln, _ = lay[n]
            has = not (r == 1 and n.startswith('affine_head_ability'))
            gs.append(res[r]['local'][o:o + ln].clone() if has else None)
        per_rank.append(gs)
    want = RO.dp_average_grads(per_rank)
    for r in range(world):
        for j, n in enumerate(names):
            o, ln, _ = lay[n]
            if want[r][j] is None:
                continue                                # grad None in the reference: Adam skips it on this rank
            assert torch.allclose(res[r]['out'][o:o + ln], want[r][j], rtol=1e-6, atol=1e-7), (r, n)


# world 8 (BASELINE.json configs[3] / configs[4]: DP = 8): the same protocol with DIFFERENT heads missing on several ranks - the divisor of
# a parameter is the number of ranks that had a gradient for it (distributed.py:36-57), 8 for the trunk, 5 / 3 / 1 for three of the heads,
# and the all-reduced fault flag of the consumer loop's NaN recovery (dotaclient_amd/optimizer.py run_iteration).  gloo on CPU; RCCL has
# never run on more than one rank (no multi-GPU box was offered in any round: DESIGN.md).
_W8_HEAD_ON = [[1, 1, 1, 1, 1], [1, 1, 1, 0, 1], [1, 0, 1, 1, 0], [1, 1, 1, 0, 0], [1, 1, 1, 0, 1], [1, 0, 1, 1, 0], [1, 1, 1, 0, 0], [1, 1, 1, 0, 0]]
_W8_HEAD_PARAMS = {1: ('affine_move_x',), 3: ('affine_unit_attention', 'affine_unit_eth'), 4: ('affine_head_ability',)}


def _w8_has_grad(rank, name):
    for k, pres in _W8_HEAD_PARAMS.items():
        if any(name.startswith(p) for p in pres):
            return bool(_W8_HEAD_ON[rank][k])
    return True


def _dp8_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dotaclient_amd import distributed as D
    D._lib.ptr = lambda t: t
    D._lib.stream_ptr = lambda: None
    D._lib.check = lambda code, what='': None
    eng = _FakeEngine(list(_W8_HEAD_ON[rank]))
    red = D.FlatGradAllReducer(eng, overlap=(rank >= 0))          # the two-collective form on every rank
    eng.reducer = red
    eng.params.fill_(float(rank + 1))
    red.sync_parameters()
    assert torch.all(eng.params == 1.0)
    local = torch.randn(eng.total, generator=torch.Generator().manual_seed(500 + rank))
    for n in eng.seg_names:
        if not _w8_has_grad(rank, n):
            o, ln, _ = eng.layout[n]
            local[o:o + ln] = 0
    eng.grads.copy_(local)
    red.start_upper(eng)
    red.finish(eng)
    # the NaN recovery's agreement step: ranks 2 and 6 hold a team-kernel fault record, everybody learns of it
    flag = torch.tensor([1.0 if rank in (2, 6) else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    torch.save({'local': local, 'out': eng.grads.clone(), 'any_fault': bool(flag.item() > 0)}, os.path.join(tmp, 'r%d.pt' % rank))
    dist.destroy_process_group()


def test_dp_flat_bucket_world_8_with_different_heads_missing_per_rank(tmp_path):
    world, port = 8, 31500 + os.getpid() % 2000
    mp.spawn(_dp8_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in range(world)]
    lay, total = L.flat_layout()
    names = list(L.param_shapes().keys())
    per_rank = [[res[r]['local'][lay[n][0]:lay[n][0] + lay[n][1]].clone() if _w8_has_grad(r, n) else None for n in names] for r in range(world)]
    want = RO.dp_average_grads(per_rank)            # distributed.py:24-57 restated: average over the ranks that have a gradient
    counts = {n: sum(_w8_has_grad(r, n) for r in range(world)) for n in names}
    assert counts['affine_pre_rnn.weight'] == 8 and counts['affine_move_x.weight'] == 6 and counts['affine_unit_attention.weight'] == 3 \
        and counts['affine_head_ability.bias'] == 3
    for r in range(world):
        assert res[r]['any_fault'] is True
        for j, n in enumerate(names):
            o, ln, _ = lay[n]
            if want[r][j] is None:
                continue                                # grad None in the reference: Adam skips it on this rank
            assert torch.allclose(res[r]['out'][o:o + ln], want[r][j], rtol=1e-6, atol=1e-7), (r, n)
    # the averaged gradient of a parameter is the same on every rank that has one
    for j, n in enumerate(names):
        o, ln, _ = lay[n]
        have = [r for r in range(world) if _w8_has_grad(r, n)]
        for r in have[1:]:
            assert torch.equal(res[r]['out'][o:o + ln], res[have[0]]['out'][o:o + ln]), n


def test_nan_recovery_takes_the_same_branch_on_every_rank():
    # Engine.recover_from_nan under data parallelism: the decision comes from the all-reduced flag, not from the rank-local
    # fault record - a rank with a STALE record (it is on the launch-per-step kernels already) must go on to the products like its peers,
    # and a rank without a record must leave the team kernels when another rank's timed out.  Host logic only: engines without a device.
    from dotaclient_amd import engine as E

    def fake(kernel_flags, record):
        e = E.Engine.__new__(E.Engine)
        e.kernel_flags, e.hidden, e.products = kernel_flags, 256, 'f16x2'
        e.status = torch.ones(1, dtype=torch.int32)
        e._ws = torch.zeros(256, dtype=torch.uint8)
        if record:
            e._ws[:32].view(torch.int32).copy_(torch.tensor([E.DC_FAULT_TEAM_TIMEOUT + 3, 0, 2, 1, 9, 4, 8, 0], dtype=torch.int32))
        e._ws_holds, e._graphs = None, {}
        return e
    # (1) one rank's team kernel timed out, nobody is on the per-step kernels yet: all eight switch, the record is consumed
    ranks = [fake(0, r == 5) for r in range(8)]
    hows = [e.recover_from_nan(True) for e in ranks]
    assert all('launch-per-step' in h for h in hows) and all(e.kernel_flags & E.DC_DIMS_RNN_PER_STEP for e in ranks)
    assert all(e.fault() is None and int(e.status.item()) == 0 and e.products == 'f16x2' for e in ranks)
    # (2) every rank is on the per-step kernels, rank 3 still holds a stale record: the flag is up, and ALL ranks fall back to the bf16x3 products
    ranks = [fake(E.DC_DIMS_RNN_PER_STEP, r == 3) for r in range(8)]
    hows = [e.recover_from_nan(True) for e in ranks]
    assert len(set(hows)) == 1 and 'bf16x3' in hows[0] and all(e.products == 'bf16x3' and e.fault() is None for e in ranks)
    # (3) no fault anywhere: products; a second NaN: nothing left
    e = fake(0, False)
    assert 'bf16x3' in e.recover_from_nan(False) and e.recover_from_nan(False) == ''
    # (4) single process (no flag passed): the local record decides, as before
    e = fake(0, True)
    assert 'layer 0, team 2' in e.recover_from_nan() and e.kernel_flags & E.DC_DIMS_RNN_PER_STEP and e.fault() is None


# ---- lane-permutation logic of the one-sequence-per-workgroup LSTM kernels (rnn_persist_valu.hip) ------------------
# numpy emulation of the reduce-scatter trees: which register of which lane holds which column, and the DPP /
# permlane-swap data movement.  The kernels compute the same index expressions; a mistake there shows up here as a
# wrong matrix-vector product (the GPU tests then check the real instructions against the oracle).
def _half_mirror16(l):
    return (l & 8) | (7 - (l & 7))


def _colmap(l, r):
    if r & 8: l ^= 8
    if r & 4: l = _half_mirror16(l)
    if r & 2: l ^= 2
    if r & 1: l ^= 1
    return l


def _row_dpp(vals, kind):
    src = {'ror8': lambda i: (i + 8) % 16, 'hm': _half_mirror16, 'x2': lambda i: i ^ 2, 'x1': lambda i: i ^ 1}[kind]
    return np.array([vals[src(i)] for i in range(16)])


@pytest.mark.parametrize('H', [64, 128])
def test_lstm_valu_forward_reduce_scatter_tree(H):
    rng = np.random.default_rng(H)
    kpl = H // 16
    W, h = rng.standard_normal((4 * H, H)), rng.standard_normal(H)
    got = np.zeros(4 * H)
    for jg in range(H // 4):
        a = np.zeros((16, 16))                       # [row lane kg][register]
        for kg in range(16):
            for r in range(16):
                c = _colmap(kg, r)
                col = (c & 3) * H + 4 * jg + (c >> 2)
                ks = [(kk >> 2) * 64 + 4 * kg + (kk & 3) for kk in range(kpl)]
                a[kg, r] = W[col, ks] @ h[ks]
        for c in range(8): a[:, c] += _row_dpp(a[:, 8 + c].copy(), 'ror8')
        for c in range(4): a[:, c] += _row_dpp(a[:, 4 + c].copy(), 'hm')
        for c in range(2): a[:, c] += _row_dpp(a[:, 2 + c].copy(), 'x2')
        a[:, 0] += _row_dpp(a[:, 1].copy(), 'x1')
        for kg in range(16):
            tid = jg * 16 + kg
            got[(tid & 3) * H + (tid >> 2)] = a[kg, 0]          # thread tid ends with gate tid & 3 of unit tid >> 2
    assert np.allclose(got, W @ h, atol=1e-12)


@pytest.mark.parametrize('H', [64, 128])
def test_lstm_valu_backward_reduce_scatter_tree(H):
    rng = np.random.default_rng(H + 1)
    kpl = H // 16
    W, dg = rng.standard_normal((4 * H, H)), rng.standard_normal(4 * H)
    lds = np.zeros(4 * H)
    for u in range(H):
        for q in range(4): lds[4 * u + q] = dg[q * H + u]       # LDS position 4*unit + gate
    got = np.zeros(H)
    for wave in range(H // 16):
        a = np.zeros((64, 16))
        for l in range(64):
            Q = (l >> 2) & 3
            for r in range(16):
                uo = 16 * wave + ((r & 12) | ((r & 3) ^ Q))
                for kk in range(kpl):
                    pos = (kk >> 2) * 256 + 4 * l + (kk & 3)
                    a[l, r] += W[(pos & 3) * H + (pos >> 2), uo] * lds[pos]
        s8 = np.zeros((64, 8))
        for c in range(8):                                        # v_permlane32_swap(a[c], a[8+c]) then add
            x, y = a[:, c].copy(), a[:, 8 + c].copy()
            nx, ny = x.copy(), y.copy()
            nx[32:], ny[:32] = y[:32], x[32:]
            s8[:, c] = nx + ny
        s4 = np.zeros((64, 4))
        for c in range(4):                                        # v_permlane16_swap: odd rows of x <-> even rows of y
            x, y = s8[:, c].copy(), s8[:, 4 + c].copy()
            nx, ny = x.copy(), y.copy()
            for row in (1, 3):
                nx[16 * row:16 * row + 16] = y[16 * (row - 1):16 * row]
                ny[16 * (row - 1):16 * row] = x[16 * row:16 * row + 16]
            s4[:, c] = nx + ny
        for row in range(4):
            tr = s4[16 * row:16 * row + 16]
            for c in range(2): tr[:, c] += _row_dpp(tr[:, 2 + c].copy(), 'ror8')
            tr[:, 0] += _row_dpp(tr[:, 1].copy(), 'hm')
            tr[:, 0] += _row_dpp(tr[:, 0].copy(), 'x1')
            tr[:, 0] += _row_dpp(tr[:, 0].copy(), 'x2')
        for l in range(64):
            got[(64 * wave + l) >> 2] = s4[l, 0]                  # all four lanes of a quad hold the unit's sum
    assert np.allclose(got, dg @ W, atol=1e-12)


# ---- the four-workgroup team kernels for H = 256 (rnn_team.hip): same emulation, plus the split over members ---------
def _swap32_sum(x, y):
    nx, ny = x.copy(), y.copy()
    nx[32:], ny[:32] = y[:32], x[32:]
    return nx + ny


def _swap16_sum(x, y):
    nx, ny = x.copy(), y.copy()
    for row in (1, 3):
        nx[16 * row:16 * row + 16] = y[16 * (row - 1):16 * row]
        ny[16 * (row - 1):16 * row] = x[16 * row:16 * row + 16]
    return nx + ny


@pytest.mark.parametrize('G', [3, 4])
def test_rnn_team_forward_tree(G):
    H, US = 256, 64
    rng = np.random.default_rng(G)
    W, h = rng.standard_normal((G * H, H)), rng.standard_normal(H)
    got = np.full(G * H, np.nan)
    for member in range(4):
        U0 = member * US
        for row in range(32):
            a = np.zeros((16, 8))
            for kg in range(16):
                for r in range(8):
                    c = _colmap(kg, r) & 7
                    gate, unit = c & 3, U0 + 2 * row + (c >> 2)
                    ks = [(kk >> 2) * 64 + 4 * kg + (kk & 3) for kk in range(16)]
                    a[kg, r] = W[gate * H + unit, ks] @ h[ks] if gate < G else 0.0
            for c in range(4): a[:, c] += _row_dpp(a[:, 4 + c].copy(), 'hm')
            for c in range(2): a[:, c] += _row_dpp(a[:, 2 + c].copy(), 'x2')
            a[:, 0] += _row_dpp(a[:, 1].copy(), 'x1')
            a[:, 0] += _row_dpp(a[:, 0].copy(), 'ror8')
            for kg in range(16):
                tid = row * 16 + kg
                q, u = tid & 3, U0 + 2 * row + ((tid >> 2) & 1)      # both "dup" halves of the row hold the same column
                if q < G:
                    assert np.isnan(got[q * H + u]) or got[q * H + u] == a[kg, 0]
                    got[q * H + u] = a[kg, 0]
    assert np.allclose(got, W @ h, atol=1e-12)


@pytest.mark.parametrize('G', [3, 4])
def test_rnn_team_backward_tree(G):
    H, US = 256, 64
    rng = np.random.default_rng(10 + G)
    W, dg = rng.standard_normal((G * H, H)), rng.standard_normal(G * H)
    lds = np.zeros(4 * H)                                            # position 4*unit + gate slot (slot 3 of the GRU: 0)
    for u in range(H):
        for q in range(G): lds[4 * u + q] = dg[q * H + u]
    got = np.zeros(H)
    for member in range(4):
        U0 = member * US
        for wave in range(8):
            a = np.zeros((64, 8))
            for l in range(64):
                b3 = (l >> 3) & 1
                for r in range(8):
                    uo = U0 + 8 * wave + ((r & 6) | ((r & 1) ^ b3))
                    for kk in range(16):
                        pos = (kk >> 2) * 256 + 4 * l + (kk & 3)
                        gate, unit = pos & 3, pos >> 2
                        if gate < G: a[l, r] += W[gate * H + unit, uo] * lds[pos]
            s4 = np.stack([_swap32_sum(a[:, c], a[:, 4 + c]) for c in range(4)], axis=1)
            s2 = np.stack([_swap16_sum(s4[:, c], s4[:, 2 + c]) for c in range(2)], axis=1)
            for row in range(4):
                tr = s2[16 * row:16 * row + 16]
                tr[:, 0] += _row_dpp(tr[:, 1].copy(), 'ror8')
                tr[:, 0] += _row_dpp(tr[:, 0].copy(), 'hm')
                tr[:, 0] += _row_dpp(tr[:, 0].copy(), 'x1')
                tr[:, 0] += _row_dpp(tr[:, 0].copy(), 'x2')
            for l in range(64):
                u = U0 + 8 * wave + (l >> 3)
                assert l & 7 == 0 or np.isclose(got[u], s2[l, 0])    # all eight lanes of a group agree
                got[u] = s2[l, 0]
    assert np.allclose(got, dg @ W, atol=1e-12)


# ---- the team kernels' exchange protocol (rnn_team.hip): tagged granules in a ring of four slots ---------------------
# A model of what the four members of a team do, run under random interleavings: a member's stream publishes its
# step output with tag T (running counter over the sequences it walks through) into slot T & 3 and, at every step
# but a sequence's first, waits for the peers' granules of tag T-1.  Claims checked: nobody overwrites a granule a
# peer still needs (the slot's old tag T-4 has been consumed by all peers or never will be read), a reader never finds a
# tag NEWER than the one it waits for (it would wait forever), and every schedule terminates.
def _run_team_model(rng, ns, seq_lens, members=4, slots=4):
    # stream s owns sequences s, s + ns, ... (one team); program of a member: round-robin over its live streams
    streams = [[l for l in seq_lens[s::ns] if l > 0] for s in range(ns)]
    ring = [[[0] * members for _ in range(slots)] for _ in range(ns)]           # ring[s][slot][member] = tag
    consumed = [[[0] * members for _ in range(members)] for _ in range(ns)]     # consumed[s][reader][writer] = last tag read

    def member_program(m):
        state = [{'seq': 0, 't': 0, 'tag': 0} for _ in range(ns)]
        live = [bool(streams[s]) for s in range(ns)]
        while any(live):
            for s in range(ns):
                if not live[s]: continue
                st = state[s]
                if st['t'] > 0:                                     # poll the peers' previous step
                    want = st['tag']
                    for w in range(members):
                        if w == m: continue
                        while True:
                            have = ring[s][want % slots][w]
                            assert have <= want, 'granule overwritten before it was read'
                            if have == want: break
                            yield                                    # spin
                        consumed[s][m][w] = want
                yield                                                # arithmetic (any delay)
                st['tag'] += 1
                T = st['tag']
                old = ring[s][T % slots][m]
                # the granule being replaced: every peer has moved past it, or it was a sequence's last step (never read)
                for r in range(members):
                    if r != m and old > 0: assert consumed[s][r][m] >= old or old in last_tags[s], (old, consumed[s][r][m])
                ring[s][T % slots][m] = T
                st['t'] += 1
                if st['t'] == streams[s][st['seq']]:
                    st['seq'] += 1; st['t'] = 0
                    if st['seq'] == len(streams[s]): live[s] = False
                yield

    last_tags = []
    for s in range(ns):
        acc, ends = 0, set()
        for l in streams[s]:
            acc += l; ends.add(acc)
        last_tags.append(ends)
    progs = [member_program(m) for m in range(members)]
    alive = list(range(members))
    steps = 0
    while alive:
        m = alive[rng.integers(len(alive))] if rng.random() < 0.7 else alive[0]   # mostly random, sometimes starve the others
        try:
            next(progs[m])
        except StopIteration:
            alive.remove(m)
        steps += 1
        assert steps < 2_000_000, 'schedule does not terminate'


@pytest.mark.parametrize('ns', [1, 2, 4])
def test_rnn_team_ring_protocol_is_safe_under_any_interleaving(ns):
    rng = np.random.default_rng(100 + ns)
    for trial in range(12):
        n = int(rng.integers(1, 9))
        lens = [int(x) for x in rng.choice([1, 1, 2, 3, 4, 5, 7, 9, 16], size=n)]
        _run_team_model(rng, ns, lens)


def test_rnn_team_ring_model_detects_a_ring_that_is_too_small():
    # the model has teeth: with two slots a fast member overwrites a granule a slow peer still waits for
    rng = np.random.default_rng(5)
    failures = 0
    for trial in range(20):
        n = int(rng.integers(2, 9))
        lens = [int(x) for x in rng.choice([3, 4, 5, 7, 9, 16], size=n)]
        try:
            _run_team_model(rng, 2, lens, slots=2)
        except AssertionError:
            failures += 1
    assert failures > 0


def test_default_device_follows_local_rank(monkeypatch):
    # rank r of a torch.distributed.run launch must not land on GPU 0 by default (optimizer.py:726-734 reads the
    # same environment); without LOCAL_RANK it is torch's current device
    from dotaclient_amd.engine import default_device
    monkeypatch.delenv('LOCAL_RANK', raising=False)
    assert default_device().type == 'cuda' and default_device().index is not None
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    monkeypatch.setenv('LOCAL_RANK', '3')
    want = 3 if (n == 0 or n > 3) else torch.cuda.current_device()      # a rank beyond the visible devices keeps the current one
    assert default_device() == torch.device('cuda', want)
    import inspect
    from dotaclient_amd import optimizer as O, policy as P, engine as E
    for fn in (O.DotaOptimizer.__init__, P.Policy.__init__, E.Engine.__init__):
        assert inspect.signature(fn).parameters['device'].default is None
    assert 'cuda:0' not in inspect.getsource(O) and 'cuda:0' not in inspect.getsource(E)


def test_pool16m_lane_level_index_model():
    # tools/pool16m_sim.py models csrc/embed_pool16m.hip lane by lane (which lane builds which MFMA operand element, which accumulator
    # register holds which output, K slot <-> unit mapping, the rank-one attention term through two K slots, the R fix-up) against a dense
    # float64 evaluation of the same gradient; an odd number of env-steps exercises the half-empty last pair
    from tools import pool16m_sim as sim
    rng = np.random.default_rng(11)
    n = 3
    x = rng.standard_normal((n, 16, 12)); W1 = rng.standard_normal((128, 12)) * 0.3; b1 = rng.standard_normal(128) * 0.3
    W2 = rng.standard_normal((128, 128)) * 0.1; amax = rng.integers(0, 16, (n, 128)); d = rng.standard_normal((n, 128))
    dtu = rng.standard_normal((n, 16)) * np.array([[1.0], [0.0], [1.0]]); q = rng.standard_normal((n, 128))
    got, ref = sim.kernel(x, W1, b1, W2, amax, d, dtu, q), sim.reference(x, W1, b1, W2, amax, d, dtu, q)
    for name, a, b in zip(('dW2', 'part1', 'db2'), got, ref):
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-12, name


def _x3_lrow(p):
    # csrc/gemm_x3.hip X3Loader::lrow: the tile row a thread slot stages
    m = p >> 2
    return 8 * (m >> 1) + (m & 1) + 2 * (p & 3)


@pytest.mark.parametrize('slots,lanes_per_slot,dwords_per_lane', [(128, 2, 4), (64, 4, 2)])
def test_gemm_x3_store_rows_tile_the_lds_banks(slots, lanes_per_slot, dwords_per_lane):
    # csrc/gemm_x3.hip (round 5): the row-major operand images have 48-byte rows (12 dwords: conflict-free fragment READS); a
    # ds_write_b128 / ds_write_b64 is serviced in contiguous groups of 8 / 16 lanes = four row slots over 32 banks.  Four CONSECUTIVE rows
    # put rows 0 and 3 on the same banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.32); lrow() makes every group's rows tile the banks.
    rows = [_x3_lrow(p) for p in range(slots)]
    assert sorted(rows) == list(range(slots))                      # every row staged exactly once
    for g0 in range(0, slots, 4):                                  # one lane group = four consecutive slots
        banks = []
        for p in range(g0, g0 + 4):
            for lane in range(lanes_per_slot):
                a = _x3_lrow(p) * 12 + lane * dwords_per_lane      # dword address of the lane's store inside the plane
                banks += [(a + i) % 32 for i in range(dwords_per_lane)]
        assert sorted(banks) == list(range(32)), (g0, sorted(banks))
    consecutive = [(p * 12 + i) % 32 for p in range(4) for i in range(8)]
    assert len(set(consecutive)) < 32                               # what the kernel did before: rows 0 and 3 collide


def _x3_decode(block, it, grid, mt, nt, splits):
    # csrc/gemm_x3.hip gemm_x3_kernel::decode: work item `it` of workgroup `block` -> (m tile, n tile, split), or None
    n_items = mt * nt * splits
    z_map = splits >= 8 and splits % 8 == 0 and grid % 8 == 0
    xcd_map = (not z_map) and mt % 8 == 0 and grid % 8 == 0
    if z_map:
        per, wi = n_items >> 3, it * (grid >> 3) + (block >> 3)
        if wi >= per:
            return None
        tiles = mt * nt
        zl, t = divmod(wi, tiles)
        return t // nt, t % nt, zl * 8 + (block & 7)
    if xcd_map:
        per, wi = n_items >> 3, it * (grid >> 3) + (block >> 3)
        if wi >= per:
            return None
        per_m = nt * splits
        ml, rest = divmod(wi, per_m)
        return ml * 8 + (block & 7), rest % nt, rest // nt
    w = it * grid + block
    if w >= n_items:
        return None
    z, t = divmod(w, mt * nt)
    return t // nt, t % nt, z


@pytest.mark.parametrize('mt,nt,splits', [(8, 4, 24), (2, 7, 48), (2, 2, 128), (16, 8, 6), (512, 8, 1), (1024, 16, 1), (3, 5, 9), (8, 1, 8)])
def test_gemm_x3_work_items_are_covered_once_and_splits_stay_in_one_xcd(mt, nt, splits):
    # every (m tile, n tile, split) exactly once whatever the mapping; with a multiple of 8 splits (round 5, the weight gradients) all
    # tiles of a split run on workgroups of ONE XCD (blockIdx & 7), so both operands of its K range are fetched from HBM once
    n_items = mt * nt * splits
    grid = min(n_items, 768)
    seen, xcd_of_split = {}, {}
    for block in range(grid):
        it = 0
        while True:
            d = _x3_decode(block, it, grid, mt, nt, splits)
            if d is None:
                break
            assert d not in seen, d
            assert 0 <= d[0] < mt and 0 <= d[1] < nt and 0 <= d[2] < splits
            seen[d] = block
            xcd_of_split.setdefault(d[2], set()).add(block & 7)
            it += 1
    assert len(seen) == n_items
    if splits >= 8 and splits % 8 == 0 and grid % 8 == 0:
        assert all(len(x) == 1 for x in xcd_of_split.values())


def test_lstm512_backward_wide_access_chunk_maps():
    # csrc/rnn_team512.hip (round 5, WIDE): a member moves its share of a step - 16 sequences x 32 units - between HBM rows and LDS images in
    # 16-byte chunks: chunk tid of the gates / dgx image = (sequence tid >> 4, gate (tid >> 2) & 3, units 8 (tid & 3) ..), chunk cf = tid & 127
    # of an f32 image = (sequence cf >> 3, units 4 (cf & 7) ..).  Every element exactly once, where the cells look for it.
    H, US, member = 512, 32, 5
    U0 = US * member
    GROW = (4 * US * 2 + 16) // 2                    # T5_GROW in bf16 elements: the gate-gradient tile's row pitch
    image = {}                                       # gates image: element index -> (sequence, column of the [4H] row)
    for tid in range(256):
        s, g, c = tid >> 4, (tid >> 2) & 3, tid & 3
        gcol = g * H + U0 + 8 * c
        for j in range(8):
            assert tid * 8 + j not in image
            image[tid * 8 + j] = (s, gcol + j)
    for s in range(16):                              # what a cell (sequence s, unit u) reads for gate g
        for u in range(US):
            for g in range(4):
                e = s * US + u
                assert image[e + (3 * s + g) * US] == (s, g * H + U0 + u)
    assert len(image) == 16 * 4 * US
    tile = {}                                        # dgx: the LDS tile the next product reads -> the dgx row
    for tid in range(256):
        s, g, c = tid >> 4, (tid >> 2) & 3, tid & 3
        src = s * GROW + g * US + 8 * c              # bf16 element offset inside the tile
        for j in range(8):
            tile[(s, src + j - s * GROW)] = g * H + U0 + 8 * c + j
    for s in range(16):
        for g in range(4):
            for u in range(US):
                assert tile[(s, g * US + u)] == g * H + U0 + u      # the cells wrote d4[g] of (s, u) at column g * 32 + u of row s
    f32 = {}
    for cf in range(128):
        s, c = cf >> 3, cf & 7
        for j in range(4):
            f32[cf * 4 + j] = (s, U0 + 4 * c + j)
    for s in range(16):
        for u in range(US):
            assert f32[s * US + u] == (s, U0 + u)


@pytest.mark.parametrize('nr', [128, 256, 384, 24 * 16, 65536, 100 * 128])
def test_fused_forward_five_unit_tiles_cover_every_unit_row_once(nr):
    # host model of embed_fused.hip's forward tiling of the five-unit type since round 6 (ef_frow / the epilogue's pool and store guards): a tile
    # holds 24 whole env-steps, block b (32 accumulator rows) the steps 24 tile + 6 b .. + 5, block row 5 s + u = unit u of its step s, rows 30 / 31
    # are padding; the block's 30 emb rows are contiguous in the type-major emb block (row 5 n + u); steps past the end are neither pooled nor stored
    n_tiles = (nr + 23) // 24
    seen = np.zeros((nr, 5), np.int32)
    pooled = np.zeros(nr, np.int32)
    for tile in range(n_tiles):
        row0 = 5 * 24 * tile                                            # ef_frow(ty, 1, tile, 0) - row_begin[1]
        for b in range(4):
            st0 = 24 * tile + 6 * b
            for r in range(32):
                q = min(r, 29)                                          # rows 30, 31 repeat the block's last unit (computed, never stored)
                local = 5 * (24 * tile + 6 * b) + q                     # the record the row is computed from
                n, u = local // 5, local % 5
                assert (n, u) == (st0 + q // 5, q % 5)
                sl = (r * 13) >> 6                                      # the kernel's r / 5 for r < 32
                if r < 30:
                    assert sl == r // 5
                    if st0 + sl < nr:                                   # the store guard (mask bit of the row)
                        assert row0 + 30 * b + r == 5 * n + u           # store address: the block's rows are contiguous
                        seen[n, u] += 1
            for sl in range(6):                                         # the pool: six steps per block, five consecutive image rows each
                if st0 + sl < nr:
                    pooled[st0 + sl] += 1
    assert (seen == 1).all() and (pooled == 1).all()
    # the types behind it start `shift` tiles later than in the 128-row tiling (ftile_begin)
    assert n_tiles - nr * 5 // 128 >= 0


def test_policy_single_wave_items_cover_every_output_once():
    # host model of csrc/policy_single.hip's work split (64 workgroups x 8 waves): stage A's (unit type, column) items publish every emb[u][c]
    # and every pooled xcat slot exactly once (slot 6 = the enh max again, policy.py:127; eth is never pooled), stage B / C rows one wave
    # each, stage D: workgroup 0 the 128 query rows (16 per wave) and the 40 target-unit logits (units wave + 8 i), workgroup 1 rows 128 .. 159
    WG, WPB = 64, 8
    u0_of = {0: 0, 1: 1, 2: 6, 3: 22, 4: 38, 5: 39}
    nu_of = {0: 1, 1: 5, 2: 16, 3: 16, 4: 1, 5: 1}
    emb = np.zeros((40, 128), np.int32)
    xcat = np.zeros(896, np.int32)
    for gw in range(WG * WPB):
        c = gw & 127
        if gw < 256:
            items = [(2 + (gw >> 7), [(2 + (gw >> 7)) + 1] + ([6] if gw >> 7 == 1 else []))]
        elif gw < 384:
            items = [(1, [2]), (0, [1])]
        else:
            items = [(4, [5]), (5, [])]
            xcat[c] += 1                                   # the env embedding: slot 0
        for t, slots in items:
            for u in range(nu_of[t]):
                emb[u0_of[t] + u, c] += 1
            for s in slots:
                xcat[s * 128 + c] += 1
    assert (emb == 1).all() and (xcat == 1).all()
    for H in (64, 128, 256, 512):                          # B: a row per wave gw < 256; C: a hidden unit per wave gw < H
        assert H <= WG * WPB and 256 <= WG * WPB
    out = np.zeros(200, np.int32)
    for wave in range(WPB):
        for i in range(16):
            out[wave * 16 + i] += 1                        # workgroup 0: query rows
        for i in range(5):
            if wave + 8 * i < 40:
                out[160 + wave + 8 * i] += 1               # ... and the target-unit logits
        for i in range(4):
            out[128 + i * 8 + wave] += 1                   # workgroup 1: rows 128 .. 159 (154 .. 159: zero weights, zero bias)
    assert (out == 1).all()
    # scratch: granules (8 bytes) of the launch generation, emb, xcat, pre, h of every layer - inside what the header promises
    words = 8 + 40 * 128 + 896 + 256 + 4 * 512
    from dotaclient_amd import _lib
    assert 2 * words <= _lib.DC_SINGLE_SCRATCH_FLOATS
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'dotaclient_hip.h')).read()
    assert '#define DC_SINGLE_SCRATCH_FLOATS %d' % _lib.DC_SINGLE_SCRATCH_FLOATS in hdr
