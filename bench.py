#!/usr/bin/env python
"""Benchmark of the PPO optimizer hot path (BASELINE.json metric: env-steps/sec through the PPO
optimizer) on N MI355X of one node, one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one optimizer iteration over one batch of synthetic trajectories already resident in
HBM: the rollout pass (no-grad forward for old log-probs/values + GAE scan, optimizer.py:328-430)
followed by E = 4 full-batch epochs of train (optimizer.py:581-689; `--epochs` default of the
reference, optimizer.py:781), each = forward + PPO loss + backward + (RCCL all-reduce if N > 1) +
clip + Adam.  Default workload = BASELINE.json configs[1]: LSTM hidden=128, 64 trajectories x 256
steps PER GPU (weak scaling).  Padded steps would count as steps as in the reference
(optimizer.py:486); the throughput runs have none.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the launch stream in one
extra, untimed iteration right after the timed region; `cpu_baseline` times the CPU oracle
(oracle/, restatement of the reference - kind "port") on the host cores, rank 0, N = 1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from dotaclient_amd import synth                      # noqa: E402
from dotaclient_amd.engine import Engine, pack_rollouts  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense fp32
PEAK_HBM_GBS = 8000.0


def fwd_flops_per_step(cell, hidden, layers):
    """algorithmic forward FLOPs (2*MACs) per env-step."""
    g = 3 if cell == 'gru' else 4
    f = 768 + 122880 + 1310720 + 458752
    inp = 256
    for _ in range(layers):
        f += 2 * g * hidden * (inp + hidden)
        inp = hidden
    return f + 2 * hidden * 154 + 10240


def profile_report(lib):
    n = 32
    names = ctypes.create_string_buffer(64 * n)
    launches = (ctypes.c_int64 * n)()
    ms = (ctypes.c_double * n)()
    fl = (ctypes.c_double * n)()
    by = (ctypes.c_double * n)()
    k = lib.dc_profile_report(names, launches, ms, fl, by, n)
    out = []
    for i in range(k):
        nm = names.raw[64 * i:64 * (i + 1)].split(b'\0')[0].decode()
        out.append({'kernel': nm, 'launches': int(launches[i]), 'total_ms': ms[i], 'flops': fl[i], 'bytes': by[i]})
    return out


def pmc_traffic(path, kernel, workload_key):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate passes - MI355X_MICROARCH.md); None when the summary is missing, was taken on a
    different workload, or does not hold the kernel."""
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None
    if j.get('meta', {}).get('workload') not in (None, workload_key):
        return None
    region_kernels = {'gru_fwd_team': 'rnn_team_fwd', 'lstm_fwd_team': 'rnn_team_fwd', 'gru_bwd_team': 'rnn_team_bwd',
                      'lstm_bwd_team': 'rnn_team_bwd', 'lstm_fwd_persist': 'lstm_fwd_valu', 'lstm_bwd_persist': 'lstm_bwd_valu'}
    k = j.get('kernels', {}).get(region_kernels.get(kernel, kernel) + '_kernel')
    return int(k['bytes']) if k else None


def cpu_baseline(cell, hidden, layers, rollouts, seq_len, epochs, lr, ent, vf):
    """Times the CPU oracle (kind "port") on the same synthetic workload: one rollout pass + `epochs`
    epochs = one bench step.  Thread count: the best of a short sweep (torch CPU ops of this size get
    slower, not faster, when spread over all 256 host threads of the GPU box)."""
    from oracle import ref_optimizer as RO
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    sd = synth.init_state_dict(7, cell, hidden, layers)

    def one_iteration(rs, n_ep):
        pol = RO.make_policy(sd, cell, hidden, layers)
        opt = torch.optim.Adam(pol.parameters(), lr=lr)
        t0 = time.time()
        chunks = [c for r in rs for c in RO.rollout_pass(pol, r, seq_len)]
        t1 = time.time()
        for _ in range(n_ep):
            RO.train_step(pol, opt, chunks, ent, vf)
        return t1 - t0, time.time() - t1, len(chunks)

    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)})
    best, best_t = cands[0], None
    for t in cands:                                   # short sweep on 4 trajectories, 1 epoch
        torch.set_num_threads(t)
        one_iteration(rollouts[:2], 1)                # warm-up
        a, b, _ = one_iteration(rollouts[:4], 1)
        if best_t is None or a + b < best_t:
            best, best_t = t, a + b
    torch.set_num_threads(best)
    t_roll, t_train, n_chunks = one_iteration(rollouts, epochs)
    n_steps = n_chunks * seq_len
    return {
        'value': round(n_steps / (t_roll + t_train), 1), 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
        'sample': '1 full bench step (rollout pass %.2fs + %d epochs %.2fs) of the same %dx%d workload; '
                  'oracle/ref_optimizer.py, torch CPU fp32, %d of %d host threads (best of sweep %s)'
                  % (t_roll, epochs, t_train, len(rollouts), seq_len, best, ncpu, cands),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--cell', default='lstm')
    ap.add_argument('--hidden', type=int, default=128)
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='trajectories per GPU')
    ap.add_argument('--seq-len', type=int, default=256)
    ap.add_argument('--epochs', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-host-extras', action='store_true',
                    help='skip the ingest / publish timings (hundreds of small copies that would pollute a kernel trace)')
    ap.add_argument('--traffic-json', default=os.path.join(REPO, 'profiles', 'pmc_traffic_latest.json'),
                    help='per-kernel HBM bytes from the rocprofv3 PMC passes (tools/gpu_round.sh + tools/pmc_traffic.py); '
                         'PMC counters cannot be read from inside the process, so `traffic` is taken from this file')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Test aid (never set by the driver): DC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo, so that the whole
    # multi-rank flow of this script (broadcast, flat-bucket all-reduce, barriers, max over ranks, rank-0 JSON) can be
    # exercised on a one-GPU box; RCCL refuses two ranks on one device.
    one_device = os.environ.get('DC_BENCH_ONE_DEVICE') == '1'
    dev_index = 0 if (one_device or world == 1) else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_device:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', dev_index))
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node = --gpus'
    dev = torch.device('cuda', dev_index)
    lr, ent, vf = 5e-5, 5e-4, 0.5
    B, S, E = args.batch, args.seq_len, args.epochs

    eng = Engine(args.cell, args.hidden, args.layers, dev)
    eng.load_state_dict(synth.init_state_dict(7, args.cell, args.hidden, args.layers))
    hook = None
    if world > 1:
        from dotaclient_amd.distributed import FlatGradAllReducer
        hook = FlatGradAllReducer(eng)
        hook.sync_parameters()
    # every rank gets its own shard of trajectories (the reference's ranks pull from a shared queue)
    rollouts = synth.make_rollouts(1000 + rank, [S] * B)
    batch = pack_rollouts(rollouts, S, dev)

    # host -> device ingest of one batch (pack_rollouts: page-locked staging + 4 H2D copies), steady state; reported
    # beside the headline, never inside it (inputs are resident in HBM when the timed region starts)
    ingest_ms = None
    if rank == 0 and not args.no_host_extras:
        for _ in range(2):
            pack_rollouts(rollouts, S, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pack_rollouts(rollouts, S, dev)
        torch.cuda.synchronize()
        ingest_ms = (time.perf_counter() - t0) / 3 * 1e3

    # model publish (optimizer.py:697-716, once per iteration): flat snapshot D2H + host views vs per-tensor .cpu() copies;
    # reported beside the headline, not inside it
    publish_ms = None
    if rank == 0 and not args.no_host_extras:
        import io

        def pub_flat():
            i = eng.start_param_snapshot()
            buf = io.BytesIO()
            torch.save(eng.snapshot_state_dict(i), buf)

        def pub_per_tensor():
            buf = io.BytesIO()
            torch.save({k: v.cpu() for k, v in eng.state_dict().items()}, buf)

        samples = {'flat_snapshot': [], 'per_tensor_copies': []}
        for it in range(17):                      # interleaved, median: single samples of host-side work scatter by 10x
            for name, fn in (('flat_snapshot', pub_flat), ('per_tensor_copies', pub_per_tensor)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                if it >= 2:
                    samples[name].append((time.perf_counter() - t0) * 1e3)
        publish_ms = {k: round(float(np.median(v)), 3) for k, v in samples.items()}

    def step():
        chunks = eng.rollout_pass(batch, S)
        for _ in range(E):
            eng.train_epoch(chunks, lr, ent, vf, grad_hook=hook)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    status = int(eng.status.item())
    losses = eng.out.cpu().numpy()

    # ---- split of a step ("also report per-epoch train-only steps/s"): one extra untimed iteration with
    # events on the launch stream around the rollout pass and around the E epochs
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    chunks_x = eng.rollout_pass(batch, S)
    ev[1].record()
    for _ in range(E):
        eng.train_epoch(chunks_x, lr, ent, vf, grad_hook=hook)
    ev[2].record()
    torch.cuda.synchronize()
    rollout_ms, epochs_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])

    # ---- roofline: one extra untimed iteration with per-launch HIP events --------------------------------
    eng.lib.dc_profile_enable(1)
    step()
    torch.cuda.synchronize()
    regions = profile_report(eng.lib)
    eng.lib.dc_profile_enable(0)

    if rank == 0:
        n_steps = world * B * S * args.steps
        value = n_steps / elapsed
        ffwd = fwd_flops_per_step(args.cell, args.hidden, args.layers)
        regions.sort(key=lambda r: -r['total_ms'])
        kernels = []
        for r in regions:
            avg_us = r['total_ms'] * 1e3 / r['launches']
            kernels.append({'kernel': r['kernel'], 'launches_per_step': r['launches'], 'avg_us': round(avg_us, 3),
                            'ms_per_step': round(r['total_ms'], 3),
                            'achieved_tflops': round(r['flops'] / (r['total_ms'] * 1e-3) / 1e12, 3),
                            # algorithmic bytes / time; HBM peak 8 TB/s (MI355X_MICROARCH.md) - the yardstick of the streaming kernels
                            'achieved_gbs': round(r['bytes'] / (r['total_ms'] * 1e-3) / 1e9, 1)})
        # the HBM-bound side (GAE / loss / Adam / pooling stream their operands once): largest by time
        # (the mask-aware attention kernels are left out: their byte count depends on the masks, the host-side figure is the dense one)
        hbm_names = ('pool_env_fwd', 'embed_scatter_bwd(+reduce)', 'ppo_loss(stats+loss+finalize)', 'gradnorm_clip_adam', 'gae_scan',
                     'select_logp')
        hbm = [r for r in regions if r['kernel'] in hbm_names]
        roofline_hbm = None
        if hbm:
            hd = hbm[0]
            gbs = hd['bytes'] / (hd['total_ms'] * 1e-3) / 1e9
            roofline_hbm = {'bound': 'hbm', 'kernel': hd['kernel'], 'achieved': round(gbs, 1), 'peak': 8000.0, 'unit': 'GB/s',
                            'frac': round(gbs / 8000.0, 4), 'avg_launch_us': round(hd['total_ms'] * 1e3 / hd['launches'], 3),
                            'algorithmic_bytes_per_launch': hd['bytes'] / hd['launches'],
                            'traffic': pmc_traffic(args.traffic_json, hd['kernel'].split('(')[0], '%s-%d-%dx%d' % (args.cell, args.hidden, B, S))}
        dom = regions[0]
        achieved = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
        # every region on this list is priced against the dense f32 rate of the chip, 157.3 TF: it is both the f32 MFMA
        # peak (the GEMM-shaped kernels) and the packed-f32 VALU peak (the sparse max-pool backward and the recurrent
        # kernels, whose algorithmic flop counts are the sparse / per-sequence ones)
        notes = {'embed_bwd_pool16': 'f32 VALU kernel (gathers of 512-byte W2 / basic rows scaled per channel): 1/16 of the '
                                     'dense MACs; its own limits are VALU issue and LDS bandwidth, not the matrix pipes',
                 'lstm_fwd_persist': 'recurrence, serial in time: latency-bound', 'lstm_bwd_persist': 'recurrence, serial in time: latency-bound'}
        for k in ('gru_fwd_team', 'gru_bwd_team', 'lstm_fwd_team', 'lstm_bwd_team'):
            notes[k] = ('recurrence, serial in time, a sequence spread over four CUs: bound by the per-step hand-off latency '
                        'between them (rnn_team.hip), not by arithmetic')
        roofline = {'bound': 'mfma', 'bound_note': notes.get(dom['kernel'], 'f32 MFMA (v_mfma_f32_32x32x2_f32)'), 'kernel': dom['kernel'], 'achieved': round(achieved, 3), 'peak': PEAK_F32_MFMA_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    'traffic': pmc_traffic(args.traffic_json, dom['kernel'], '%s-%d-%dx%d' % (args.cell, args.hidden, B, S)),
                    'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, %s)' % os.path.relpath(args.traffic_json, REPO),
                    'algorithmic_bytes_per_launch': dom['bytes'] / dom['launches'],
                    'avg_launch_us': round(dom['total_ms'] * 1e3 / dom['launches'], 3),
                    'flops_per_launch': dom['flops'] / dom['launches'],
                    'whole_step': {'flops_per_env_step': ffwd * (1 + 3 * E),
                                   'achieved_tflops': round(value / world * ffwd * (1 + 3 * E) / 1e12, 3),
                                   # steps/s x F_iter / peak, per GPU, dense (reference) flop count
                                   'frac_of_f32_peak': round(value / world * ffwd * (1 + 3 * E) / 1e12 / 157.3, 4)},
                    'kernels': kernels}
        key = (args.cell, args.hidden, args.layers, B, S)
        which = {('lstm', 128, 1, 64, 256): 'BASELINE.json configs[1] (1v1-mid, the configuration the metric is quoted on)',
                 ('lstm', 256, 1, 256, 256): 'BASELINE.json configs[2] (5v5: 256 trajectories, LSTM-256)',
                 ('gru', 256, 1, 64, 256): "configs[1]'s batch with the reference's own cell (GRU-256, policy.py:66)"}.get(
                     key, 'other configuration (not a BASELINE.json bench line)')
        line = {
            'metric': 'env-steps/sec through PPO optimizer', 'value': round(value, 1), 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: synthetic trajectories, %s hidden=%d x%d layer, '
                                   'batch=%d trajectories x %d steps per GPU, %d epochs + rollout pass per step'
                                   % (which, args.cell.upper(), args.hidden, args.layers, B, S, E),
                       'cell': args.cell, 'hidden': args.hidden, 'layers': args.layers, 'batch_per_gpu': B,
                       'seq_len': S, 'epochs': E, 'parallelism': 'dp%d' % world},
            'phases': {'rollout_pass_ms': round(rollout_ms, 3), 'epoch_ms': round(epochs_ms / E, 3),
                       'train_only_env_steps_per_s_per_gpu': round(B * S / (epochs_ms / E * 1e-3), 1),
                       'note': 'one untimed iteration on rank 0: no-grad forward + old log-probs + GAE, then the mean of the %d '
                               'full-batch epochs (forward, loss, backward, clip + Adam); train-only = B*S / epoch time' % E},
            'roofline': roofline,
            'roofline_hbm': roofline_hbm,
            'nan_status': status, 'final_loss': float(losses[0]),
            'ingest': None if ingest_ms is None else {
                       'pack_h2d_ms_per_batch': round(ingest_ms, 3),
                       'env_steps_per_s_with_ingest_serialised': round(B * S / (elapsed / args.steps + ingest_ms * 1e-3), 1),
                       'note': 'wire-format dicts -> page-locked staging (dc_pack_rows, DC_PACK_THREADS host threads) -> HBM (engine.pack_rollouts); '
                               'not part of `value`'},
            'publish': {'ms_per_publish': publish_ms,
                        'note': 'model publish once per iteration (optimizer.py:697-716): D2H + torch.save of the 34-tensor state_dict; '
                                'flat_snapshot = one asynchronous copy of the flat buffer into page-locked memory (Engine.start_param_snapshot), '
                                'per_tensor_copies = the reference\'s form; not part of `value`'},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.cell, args.hidden, args.layers, rollouts, S, E, lr, ent, vf)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
