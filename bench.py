#!/usr/bin/env python
"""Benchmark of the PPO optimizer hot path (BASELINE.json metric: env-steps/sec through the PPO
optimizer) on N MI355X of one node, one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one optimizer iteration over one batch of synthetic trajectories already resident in
HBM: the rollout pass (no-grad forward for old log-probs/values + GAE scan, optimizer.py:328-430)
followed by E = 4 full-batch epochs of train (optimizer.py:581-689; `--epochs` default of the
reference, optimizer.py:781), each = forward + PPO loss + backward + (RCCL all-reduce if N > 1) +
clip + Adam.  Default workload = the 5v5 LSTM-256 family of BASELINE.json: N = 1 runs configs[2] (LSTM
hidden=256, 256 trajectories x 256 steps on the one GPU), N > 1 runs configs[3]'s geometry (128 trajectories x
256 steps PER GPU = 1024 x 256 at DP=8, RCCL all-reduce of the flat gradient bucket every epoch; weak scaling:
the per-GPU shard is fixed for every N > 1, and the N = 1 line also carries the one-GPU rate of that 128-trajectory
shard as `weak_scaling_unit` so that a scaling efficiency can be formed from like with like).  configs[1]
(LSTM-128, 64 x 256) and the reference's own GRU-256 on the same batch are measured as `secondary` lines at N = 1.
Padded steps would count as steps as in the reference (optimizer.py:486); the throughput runs have none.

Prints ONE JSON line (rank 0) and flushes it BEFORE anything optional runs: timed loop -> status check -> profile iteration
-> cpu_baseline / parity -> print.  Side measurements (ingest, publish, hipGraph replay, reuse-forward, the other
single-GPU configurations) only run with `--extras`, in a SUBPROCESS, after the line is out; their JSON goes to
`--extras-out` (default gpurun_out/bench_extras.json) and to stderr - a failure there cannot cost the headline (round 2's
BENCH_r02 died in such a side workload with the headline still unprinted; profiles/r03/crash_bisect.md).
`roofline` is measured live with HIP events on the launch stream in one
extra, untimed iteration right after the timed region; `cpu_baseline` times the CPU oracle
(oracle/, restatement of the reference - kind "port") on the host cores, rank 0, N = 1 only, on one full step of the
same workload - and the same oracle run is the checker of `parity`: advantages, returns, old log-probs, the losses /
entropies / gradient norms of all E epochs and the post-step parameters of the HIP path's first iteration (from the
same initial weights) against the oracle's, masked argmax indices bit-exact (BASELINE.json north_star: 1e-4).
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

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense fp32 (= the f32 VALU peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
# f32-grade products by exact 3-way bf16 splitting execute SIX bf16 MFMAs per f32 product (gemm_tiles.h): the matrix pipes'
# ceiling for ALGORITHMIC f32 flops on that path
PEAK_X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# ... and with TWO f16 pieces per operand (Engine.products 'f16x2', the default: csrc/gemm_x3.hip PREC = 4) THREE f16 MFMAs (same rate as bf16):
# hh, hm, mh - the m*m term is below what the two pieces can represent anyway (csrc/gemm_tiles.h, DC_X2H_MM)
PEAK_X2H_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0
PRODUCTS = 'f16x2'        # --products
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
    # profiling region -> kernel(s) that run inside it (the first one present in the summary wins)
    region_kernels = {'embed_bwd_pool16': ['embed_bwd_pool16w', 'embed_bwd_pool16'], 'embed_bwd_pool16m': [('embed_pool16m_dw2', 'embed_pool16m_dw1')], 'gru_fwd_team': ['team_mfma_fwd_of', 'team_mfma_fwd_col', 'team_mfma_fwd', 'team8_fwd', 'rnn_team_fwd'],
                      'gru_bwd_team': ['team_mfma_bwd', 'team8_bwd', 'rnn_team_bwd'],
                      'lstm_fwd_team': ['team_mfma_fwd_of', 'team_mfma_fwd_col', 'team_mfma_fwd', 'team8_fwd', 'rnn_team_fwd'], 'lstm_bwd_team': ['team_mfma_bwd', 'team8_bwd', 'rnn_team_bwd'],
                      'embed_bwd_small': ['embed_small_bwd'], 'gradnorm_clip_adam': ['gradnorm_clip_adam'],
                      'embed_scatter_bwd': ['embed_env_bwd', 'embed_scatter_bwd'], 'attn_logits': ['attn_logits_masked', 'attn_logits'],
                      'lstm_fwd_persist': ['lstm_fwd_valu'], 'lstm_bwd_persist': ['lstm_bwd_valu'],
                      'gemm_f32_dW': ['gemm_x3'], 'gemm_f32_fwd': ['gemm_fast', 'gemm_x3'], 'gemm_f32_dX': ['gemm_fast', 'gemm_x3']}
    if PRODUCTS == 'f16x2':       # x W^T / dy W: the row-streaming kernel (gemm_x3s.hip, round 6; one PMC row: the average over its launches), else the split-on-load kernel
        region_kernels['gemm_f32_fwd'] = region_kernels['gemm_f32_dX'] = ['gemm_x3s', 'gemm_x3', 'gemm_fast']
    for cand in region_kernels.get(kernel, [kernel]):
        if isinstance(cand, tuple):                   # a region of several kernels, each launched once per pass: the sum
            ks = [j.get('kernels', {}).get(c + '_kernel') for c in cand]
            if all(ks):
                return int(sum(k['bytes'] for k in ks))
            continue
        k = j.get('kernels', {}).get(cand + '_kernel')
        if k:
            return int(k['bytes'])
    return None


def pmc_whole_step(path, workload_key, passes_per_step):
    """Sum over the library's kernels of (HBM bytes per launch x launches) / iterations in the PMC summary: the HBM traffic of one
    bench step.  The number of iterations the profiled run made is recovered from the fused embedding forward, which runs once per pass."""
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None
    if j.get('meta', {}).get('workload') not in (None, workload_key):
        return None
    ks = j.get('kernels', {})
    ref = ks.get('embed_fwd_fused_kernel')
    if not ref or not ref.get('launches'):
        return None
    iters = ref['launches'] / float(passes_per_step)
    total = sum(v['bytes'] * v['launches'] for k, v in ks.items() if k.endswith('_kernel') and not k.startswith(('void', 'at::', '__amd')))
    return total / iters


def _tensor_samples(t, stride=251):
    return t.detach().flatten()[::stride][:1000].float().cpu().numpy().copy()


def hip_parity_iteration(eng, batch, S, E, lr, ent, vf, hook=None):
    """The HIP path's FIRST iteration (from the freshly loaded weights), with everything `parity` compares read back."""
    from dotaclient_amd import layout as L
    chunks = eng.rollout_pass(batch, S)
    out = {'advantages': batch.adv.cpu().numpy(), 'returns': batch.ret.cpu().numpy(), 'values': batch.values.cpu().numpy(),
           'argmax': batch.argmax.cpu().numpy()}
    act = batch.act.cpu().numpy()
    lp = batch.old_logp.cpu().numpy()
    for k, name in enumerate(L.OUTPUT_KEYS):
        o = L.HEAD_OFFSETS[name]
        out['old_logp_' + name] = lp[act[:, o:o + L.HEAD_COUNTS[name]].any(axis=1), k]
    per_epoch = []
    for _ in range(E):
        res, status = eng.train_epoch(chunks, lr, ent, vf, grad_hook=hook)
        per_epoch.append(res.cpu().numpy().astype(np.float64)[:11])
    out['epochs'] = np.stack(per_epoch)
    names = list(L.param_shapes(eng.cell, eng.hidden, eng.layers).keys())
    out['param_samples'] = np.concatenate([_tensor_samples(eng.param_view(n)) for n in names])
    return out


def oracle_iteration(cell, hidden, layers, rs, seq_len, n_ep, lr, ent, vf, keep=False):
    """ONE optimizer iteration of the CPU oracle (oracle/ref_optimizer.py: rollout pass + n_ep epochs from the seed-7 weights) on
    rollouts `rs`; returns (rollout-pass seconds, epoch seconds, chunks, read-outs for `parity` or None)."""
    from oracle import ref_optimizer as RO
    sd = synth.init_state_dict(7, cell, hidden, layers)
    pol = RO.make_policy(sd, cell, hidden, layers)
    opt = torch.optim.Adam(pol.parameters(), lr=lr)
    t0 = time.time()
    chunks = [c for r in rs for c in RO.rollout_pass(pol, r, seq_len)]
    t1 = time.time()
    ref = None
    if keep:                                       # read-outs of the rollout pass: outside the timed spans
        ref = {'advantages': torch.stack([c.advantages for c in chunks]).numpy().ravel(),
               'returns': torch.stack([c.returns for c in chunks]).numpy().ravel(),
               'values': torch.stack([c.values for c in chunks]).numpy().ravel(),
               'argmax': RO.masked_argmax(pol, chunks).numpy().reshape(-1, 5)}
        for k in RO.HEADS:
            ref['old_logp_' + k] = torch.cat([c.old_logp[k] for c in chunks]).numpy()
    t_train = 0.0
    per_epoch = []
    for _ in range(n_ep):
        t2 = time.time()
        parts, entr, norms = RO.train_step(pol, opt, chunks, ent, vf)
        t_train += time.time() - t2
        per_epoch.append([float(parts[k]) for k in ('loss', 'policy_loss', 'entropy_loss', 'value_loss')] +
                         [float(entr[k]) for k in RO.HEADS] + [float(norms['unclipped']), float(norms['clipped'])])
    if keep:
        ref['epochs'] = np.array(per_epoch, dtype=np.float64)
        ref['param_samples'] = np.concatenate([_tensor_samples(p) for _, p in pol.named_parameters()])
    return t1 - t0, t_train, len(chunks), ref


def cpu_baseline(cell, hidden, layers, rollouts, seq_len, epochs, lr, ent, vf):
    """Times the CPU oracle (kind "port") on the same synthetic workload: after a thread sweep and two warm-up steps on a bounded
    sample (the first 64 trajectories), ONE full bench step (rollout pass + `epochs` epochs) over ALL trajectories of the GPU's own
    batch - `value`, so that any GPU / CPU ratio compares the same workload, ~16 s of CPU work at configs[2] - which is
    also the checker side of `parity`.  Thread count: the best of a short sweep (torch CPU ops of this size get slower, not
    faster, when spread over all host threads of the GPU box).

    kind is "port": the reference itself (Python, /root/reference) cannot travel to the GPU box, and this script cannot run where the
    reference is (no GPU there) - tools/reference_cpu_step.py times the REAL reference's functions (optimizer.py:57-64,328-430,581-689) next
    to this port in the build container (profiles/r05/reference_vs_port_cpu.json: 489.2 against 486.9 env-steps/s on the same cores, the same final loss
    to the last digit - the port is a faithful stand-in for the reference's CPU speed)."""
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)

    def one_iteration(rs, n_ep, keep=False):
        return oracle_iteration(cell, hidden, layers, rs, seq_len, n_ep, lr, ent, vf, keep)

    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)})
    best, best_t = cands[0], None
    for t in cands:                                   # short sweep: 8 trajectories, 1 epoch, after a warm-up of the same size
        torch.set_num_threads(t)
        one_iteration(rollouts[:8], 1)
        a, b, _, _ = one_iteration(rollouts[:8], 1)
        if best_t is None or a + b < best_t:
            best, best_t = t, a + b
    torch.set_num_threads(best)
    # warm-up of the timed run: one step on the bounded sample (the sweep above ran 8 trajectories only)
    sample = rollouts[:min(64, max(8, len(rollouts) // 4))]
    rates = []
    for _ in range(2):
        a, b, n, _ = one_iteration(sample, epochs)
        rates.append(n * seq_len / (a + b))
    t_roll, t_train, n_chunks, ref = one_iteration(rollouts, epochs, keep=True)      # the timed run = `parity`'s checker
    full_rate = n_chunks * seq_len / (t_roll + t_train)
    return {
        'value': round(full_rate, 1), 'unit': 'env-steps/s', 'cores': best, 'kind': 'port',
        'sample': 'ONE full bench step (rollout pass %.2fs + %d epochs %.2fs) on ALL %d trajectories x %d steps of the GPU\'s own batch - the same '
                  'run is `parity`\'s checker - after two warm-up steps on the first %d trajectories (%s env-steps/s, reported as '
                  '`sample_rates`); oracle/ref_optimizer.py, torch CPU fp32, %d of %d host threads (best of sweep %s)'
                  % (t_roll, epochs, t_train, len(rollouts), seq_len, len(sample), [round(r, 1) for r in rates], best, ncpu, cands),
        'sample_rates': [round(r, 1) for r in rates], 'sample_trajectories': len(sample),
        'reference_check': _reference_check(),
    }, ref


def _reference_check():
    """Why a "port" may stand in for the reference's CPU speed: the committed measurement of the REAL reference's functions next to this port
    (tools/reference_cpu_step.py in the build container, where /root/reference is - it does not exist on the GPU box).  Quoted, with its
    provenance, inside `cpu_baseline`; None if the file is missing."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r05', 'reference_vs_port_cpu.json')
    try:
        with open(path) as f:
            j = json.load(f)
    except (OSError, ValueError):
        return None
    return {'source': 'profiles/r05/reference_vs_port_cpu.json (tools/reference_cpu_step.py, build container, 8-core host, no GPU)',
            'workload': j.get('workload'),
            'reference_env_steps_per_s': j.get('reference', {}).get('env_steps_per_s'), 'port_env_steps_per_s': j.get('port', {}).get('env_steps_per_s'),
            'port_over_reference_rate': j.get('port_over_reference_rate'), 'same_final_loss': j.get('same_final_loss'),
            'note': 'the reference is Python under /root/reference and cannot travel to the GPU box; the port is pinned to its golden outputs (tests/test_oracle.py)'}


def parity_report(got, ref, tol=1e-4, argmax_min_equal=None, sub_batch=None, checker=None):
    """HIP path vs oracle on the bench workload itself.  Vectors: max |a-b| / max |b|; per-epoch scalars: relative, a loss
    part measured against max(|itself|, 1 % of the largest part), the total against the sum of |parts| (the same yardsticks as tests/test_gpu_parity.py)."""
    def scaled(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if a.size else 0.0
    def elementwise(a, b, floor=1e-3):
        a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
        keep = np.abs(b) > floor * (np.max(np.abs(b)) if b.size else 0.0)
        return float(np.max(np.abs(a[keep] - b[keep]) / np.abs(b[keep]))) if keep.any() else 0.0
    rep, elem = {}, {}
    for k in ['advantages', 'returns', 'values', 'param_samples'] + [k for k in ref if k.startswith('old_logp_')]:
        rep[k] = scaled(got[k], ref[k])
        elem[k] = elementwise(got[k], ref[k])
    ge, re_ = got['epochs'], ref['epochs']
    den = np.abs(re_) + 1e-30
    den[:, 1:4] = np.maximum(den[:, 1:4], 0.01 * np.abs(re_[:, 1:4]).max(axis=1, keepdims=True))
    den[:, 0] = np.abs(re_[:, 1:4]).sum(axis=1)          # the loss is the sum of the three parts and cancels (tests/util.py)
    err = np.abs(ge - re_) / den
    rep['losses'] = float(err[:, :4].max())
    rep['entropies'] = float(err[:, 4:9].max())
    rep['grad_norms'] = float(err[:, 9:11].max())
    # the plain relative error of the TOTAL loss, |a - b| / |b| per epoch, beside the sum-of-|parts| yardstick above (reported,
    # not part of `ok` - the total is a sum of signed parts and may cancel)
    total_plain = float((np.abs(ge[:, 0] - re_[:, 0]) / (np.abs(re_[:, 0]) + 1e-30)).max())
    if argmax_min_equal is not None:
        rep.pop('returns', None)       # do not depend on the network: bit-exact either way
    worst = max(rep.values())
    argmax_equal = bool(np.array_equal(got['argmax'], ref['argmax'].astype(got['argmax'].dtype)))
    argmax_frac = float((got['argmax'] == ref['argmax'].astype(got['argmax'].dtype)).mean())
    argmax_ok = argmax_equal if argmax_min_equal is None else argmax_frac >= argmax_min_equal
    return {'checker': checker or 'oracle/ref_optimizer.py (fp32) on the same %d env-steps%s, first iteration from the same initial weights, all %d epochs'
                       % (ref['advantages'].size, '' if sub_batch is None else ' (the first %d trajectories of the timed batch, run as a batch of their own)' % sub_batch,
                          re_.shape[0]),
            'parity_rel_err': worst, 'tolerance': tol, 'argmax_bit_exact': argmax_equal, 'argmax_equal_fraction': argmax_frac,
            'argmax_rule': 'bit-exact' if argmax_min_equal is None else 'bf16 path: >= %.2f of the masked argmax indices equal (near-ties flip under a 2^-9 perturbation; tests/test_gpu_bf16.py)' % argmax_min_equal,
            'ok': bool(worst < tol and argmax_ok), 'per_quantity': dict({k: float('%.3g' % v) for k, v in rep.items()}, total_loss_plain_relative=float('%.3g' % total_plain)),
            'yardstick': 'vectors: max|a-b| / max|b|; loss parts: relative (floor 1 % of the largest part), total loss against the sum of |parts|',
            'elementwise_rel_err': {k: float('%.3g' % v) for k, v in elem.items()},
            'elementwise_note': 'reported beside the scaled figures, not part of `ok`: max over entries with |ref| > 1e-3 * max|ref| of '
                                '|a-b| / |ref| (the literal per-entry reading of "1e-4 relative"; an entry of 1e-3 * max that is a sum of '
                                'O(max) f32 terms is only defined to ~1e-4 of itself)',
            'final_epoch_losses_hip': [float(x) for x in ge[-1, :4]], 'final_epoch_losses_oracle': [float(x) for x in re_[-1, :4]]}


# what bounds each timed region (the rates in `kernels` are priced against that resource's peak)
# peak (TFLOP/s of algorithmic f32 flops) of the MFMA-bound regions.  The embedding kernels regenerate their first layer
# (K = 12 of the 12 + 128, resp. 24 of 24 + 128 flops per output) with f32 MFMAs and run the 128 x 128 layer as x3 products:
# time-weighted harmonic peak.  With the f16x2 products the first layer runs on the f16 MFMAs as well: the plain f16x2 ceiling.
def _mixed_peak(f32_share):
    if PRODUCTS == 'f16x2':
        return PEAK_X2H_TFLOPS
    grade = PEAK_X3_TFLOPS
    return 1.0 / (f32_share / PEAK_F32_MFMA_TFLOPS + (1.0 - f32_share) / grade)


def mfma_peak(region, prec_bf16):
    if prec_bf16 and region.startswith('gemm_'):
        return PEAK_BF16_MFMA_TFLOPS
    return {'embed_fwd_fused': _mixed_peak(12.0 / 140.0), 'embed_bwd_dw2': _mixed_peak(12.0 / 140.0),
            'embed_bwd_dw1': _mixed_peak(24.0 / 152.0)}.get(region, PEAK_X2H_TFLOPS if PRODUCTS == 'f16x2' else PEAK_X3_TFLOPS)


def latency_peak(region, prec_bf16, hidden):
    """Peak for the VALU / latency-bound regions.  f32 kernels: the f32 vector peak (= the f32-input MFMA peak).  In bf16 mode the
    H >= 512 recurrences run on v_mfma_f32_32x32x16_bf16 (rnn_team512.hip, rnn_step_bf16.hip): priced against the dense bf16 MFMA
    peak (against the f32 peak their `frac` came out above 1)."""
    if prec_bf16 and hidden >= 512 and region in ('lstm_fwd_team', 'lstm_bwd_team', 'rnn_fwd_steps', 'rnn_bwd_steps'):
        return PEAK_BF16_MFMA_TFLOPS
    return PEAK_F32_MFMA_TFLOPS


REGION_BOUND = {
    'embed_fwd_fused': 'mfma', 'embed_bwd_dw2': 'mfma', 'embed_bwd_dw1': 'mfma', 'gemm_f32_fwd(NT)': 'mfma', 'gemm_f32_dX(NN)': 'mfma',
    'gemm_f32_dW(TN,split-K)': 'mfma',
    'embed_bwd_pool16': 'valu', 'embed_bwd_pool16m': 'mfma', 'embed_bwd_small': 'mfma',
    'lstm_fwd_persist': 'latency', 'lstm_bwd_persist': 'latency', 'gru_fwd_team': 'latency', 'gru_bwd_team': 'latency',
    'lstm_fwd_team': 'latency', 'lstm_bwd_team': 'latency', 'rnn_fwd_steps': 'latency', 'rnn_bwd_steps': 'latency',
    'pool_env_fwd': 'hbm', 'embed_scatter_bwd(+reduce)': 'hbm', 'ppo_loss(stats+loss+finalize)': 'hbm', 'gradnorm_clip_adam': 'hbm',
    'gae_scan': 'hbm', 'select_logp': 'hbm', 'attn_logits': 'hbm', 'attn_bwd_q': 'hbm', 'colsum': 'hbm',
}
BOUND_NOTES = {
    'mfma': 'f32-grade products on the 16-bit matrix cores: every f32 operand is split into two f16 pieces (x 2^s = h + m, 23 of the 24 '
            'significand bits; fixed power-of-two pre-scales) and a product is three v_mfma_f32_32x32x16_f16 (hh, hm, mh; Engine.products f16x2, the '
            'default; gemm_x3.hip PREC 4) - or into three bf16 pieces and six MFMAs (--products bf16x3); `achieved` = ALGORITHMIC f32 '
            'flops / time, `peak` = the dense 16-bit MFMA peak / 3 = 833 TF (/ 6 = 416.7 TF for bf16x3, where the embedding kernels\' K = 12 first layer '
            'is regenerated with f32 MFMAs and priced so), so `frac` is the share of the matrix pipes\' capacity on this path; '
            '`frac_of_f32_mfma_peak` prices the same rate against the 157.3 TF of the f32-input MFMA the products would otherwise run on',
    'valu': 'packed-f32 VALU kernel (per-channel-scaled gathers of 512-byte W2 / basic rows; 1/16 of the dense MACs): priced against '
            'the f32 VALU peak, which equals the f32 MFMA peak (157.3 TF at 2.4 GHz); what limits it is VALU issue and LDS '
            'bandwidth at 2 waves/SIMD, not the matrix pipes',
    'latency': 'recurrence, serial in time: bound by the per-step dependency chain (and, for the H=256 team kernels, the hand-off '
               'between the four CUs that share a sequence), not by arithmetic or HBM',
    'hbm': 'streams its operands once: priced against the 8 TB/s HBM peak',
}
REGION_NOTES = {
    'embed_bwd_small': 'embedding backward of the four small unit types (8 of the 40 units; policy.py:100-105,118-127 under optimizer.py:672) as one '
                       'kernel with every operand formed on chip (csrc/embed_small.hip, round 6: d(emb) exists for no type): flops = 2 x rows x 128 x '
                       '(2 x 128 + 24); rounds 1-5: embed_scatter_bwd wrote d(emb), embed_bwd_dw2 + embed_bwd_dw1 read it back (0.16 / 0.12 of the ceiling)',
    'embed_bwd_pool16m': 'max-pool backward of the two 16-unit types as the DENSE products of the reference\'s autograd (policy.py:102-136 under '
                         'optimizer.py:672) on the f16 matrix cores, every operand generated on chip (csrc/embed_pool16m.hip: two kernels, dW2 and '
                         'd(basic) -> dW1): flops = the dense count SURVEY.md 8(d) uses; round 4 ran the sparse form on the vector unit '
                         '(embed_bwd_pool16: 21.5 GFLOP per launch in 683-719 us + a 110 us prepare pass, 0.19-0.20 of the 157 TF vector peak)',
}
PARITY_FULL_MAX = 65536   # env-steps per GPU up to which `parity` / `cpu_baseline` run the oracle on the whole timed batch
KERNEL_FLAGS = 0      # --kernel-flags: DC_DIMS_* kernel-selection overrides for A/B runs (include/dotaclient_hip.h)
USE_GRAPHS = False    # --epoch-graph: replay every epoch as one hipGraph launch (Engine.train_epoch(graph=True))
REUSE_FORWARD = False  # side measurement only: Engine.reuse_rollout_forward
MASK_DEPENDENT_BYTES = ('attn_logits', 'attn_bwd_q')   # mask-aware: bytes moved depend on the masks; the host-side figure is the dense form


def run_workload(cell, hidden, layers, B, S, E, steps, warmup, dev, rank, world, hook_factory=None, want_parity=False,
                 want_profile=False, lengths=None, data_seed=1000, parity_epochs=None):
    """Builds an engine + a resident batch, runs warmup + `steps` timed iterations; returns a dict of raw results."""
    lr, ent, vf = 5e-5, 5e-4, 0.5
    eng = Engine(cell, hidden, layers, dev)
    eng.kernel_flags = KERNEL_FLAGS
    eng.products = PRODUCTS
    eng.use_graphs = USE_GRAPHS
    eng.reuse_rollout_forward = REUSE_FORWARD
    eng.load_state_dict(synth.init_state_dict(7, cell, hidden, layers))
    hook = hook_factory(eng) if hook_factory is not None else None
    if hook is not None:
        hook.sync_parameters()
    # every rank gets its own shard of trajectories (the reference's ranks pull from a shared queue)
    rollouts = synth.make_rollouts(data_seed + rank, [S] * B if lengths is None else lengths)
    batch = pack_rollouts(rollouts, S, dev)
    res = {'eng': eng, 'rollouts': rollouts, 'batch': batch, 'hook': hook, 'lr': lr, 'ent': ent, 'vf': vf}
    if want_parity:
        res['first_iteration'] = hip_parity_iteration(eng, batch, S, E if parity_epochs is None else parity_epochs, lr, ent, vf, hook)

    def step():
        chunks = eng.rollout_pass(batch, S)
        for _ in range(E):
            eng.train_epoch(chunks, lr, ent, vf, grad_hook=hook)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    res['elapsed'] = elapsed
    res['status'] = int(eng.status.item())
    res['losses'] = eng.out.cpu().numpy()
    res['step'] = step
    if want_profile:
        # split of a step ("also report per-epoch train-only steps/s"): one extra untimed iteration with
        # events on the launch stream around the rollout pass and around the E epochs
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        chunks_x = eng.rollout_pass(batch, S)
        ev[1].record()
        for _ in range(E):
            eng.train_epoch(chunks_x, lr, ent, vf, grad_hook=hook)
        ev[2].record()
        torch.cuda.synchronize()
        res['rollout_ms'], res['epochs_ms'] = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        # roofline: one extra untimed iteration with per-launch HIP events
        eng.lib.dc_profile_enable(1)
        step()
        torch.cuda.synchronize()
        res['regions'] = profile_report(eng.lib)
        eng.lib.dc_profile_enable(0)
    return res


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--cell', default='lstm')
    ap.add_argument('--hidden', type=int, default=256)
    ap.add_argument('--layers', type=int, default=1)
    ap.add_argument('--batch', type=int, default=0,
                    help='trajectories per GPU; 0 = BASELINE.json: 256 at N = 1 (configs[2]), 128 at N > 1 (configs[3]: 1024 at DP=8)')
    ap.add_argument('--seq-len', type=int, default=256)
    ap.add_argument('--epochs', type=int, default=4)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='also skips `parity` (the oracle run is its checker)')
    ap.add_argument('--no-weak-unit', action='store_true', help='skip the 128-trajectory `weak_scaling_unit` run of the default N = 1 line')
    ap.add_argument('--extras', action='store_true',
                    help='after the JSON line is printed and flushed: run the side measurements (ingest, publish, hipGraph replay, '
                         'reuse-forward, the other single-GPU configurations) in a subprocess; result to --extras-out and stderr')
    ap.add_argument('--extras-only', action='store_true', help='(what --extras runs in the subprocess) print only the side measurements')
    ap.add_argument('--no-secondary', action='store_true', help='skip the `secondary` / `products_fallback` blocks of the default N = 1 line')
    ap.add_argument('--secondary-only', action='store_true', help='(what the default line runs in a subprocess) print only those two blocks')
    ap.add_argument('--parity-ref', default='', help='(--secondary-only) .npz with the oracle read-outs of the headline batch, for `products_fallback`')
    ap.add_argument('--extras-out', default=os.path.join(REPO, 'gpurun_out', 'bench_extras.json'))
    ap.add_argument('--kernel-flags', type=int, default=0, help='DC_DIMS_* kernel-selection overrides (A/B measurements)')
    ap.add_argument('--products', default='f16x2', choices=['f16x2', 'bf16x3'],
                    help='form of the f32-grade products (Engine.products): two f16 pieces / three MFMAs (default) or three bf16 pieces / six MFMAs')
    ap.add_argument('--epoch-graph', type=int, default=0, help='1: replay each epoch as ONE hipGraph launch (single GPU), 0: eager launches')
    ap.add_argument('--traffic-json', default=os.path.join(REPO, 'profiles', 'pmc_traffic_latest.json'),
                    help='per-kernel HBM bytes from the rocprofv3 PMC passes (tools/gpu_round.sh + tools/pmc_traffic.py); '
                         'PMC counters cannot be read from inside the process, so `traffic` is taken from this file')
    # accepted and ignored (round-2 command lines): the side measurements are opt-in now
    ap.add_argument('--no-host-extras', action='store_true', help=argparse.SUPPRESS)
    return ap.parse_args()


def check_status(res, what):
    """Stops the bench the moment a workload reports a non-zero status word (1 NaN loss, 2 NaN gradient norm, >= 16 a team kernel
    that timed out: include/dotaclient_hip.h) instead of timing garbage."""
    st = res['status']
    if st != 0:
        from dotaclient_amd.engine import describe_status
        sys.stderr.write('bench.py: %s ended with status %s\n' % (what, describe_status(res['eng'])))
        sys.stderr.flush()
        sys.exit(4)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): re-executes this very command line
    under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU, and passes
    rank 0's JSON line through.  Refuses (exit code 2, nothing on stdout) when fewer than N devices are visible - a one-rank run
    labelled as N GPUs, or N ranks piled on one device, would be a wrong number, not a slow one."""
    import socket
    import subprocess
    one_device = os.environ.get('DC_BENCH_ONE_DEVICE') == '1'       # test aid: every rank on cuda:0 over gloo (see main)
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if one_device else args.gpus):
        sys.stderr.write('bench.py: --gpus %d asked for, %d GPU(s) visible: refusing to run (no line printed)\n' % (args.gpus, have))
        sys.stderr.flush()
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')               # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    sys.stderr.write('bench.py: launching %d ranks: %s\n' % (args.gpus, ' '.join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    global KERNEL_FLAGS, USE_GRAPHS, REUSE_FORWARD, PRODUCTS
    KERNEL_FLAGS = args.kernel_flags
    PRODUCTS = args.products
    USE_GRAPHS = args.epoch_graph == 1

    if args.gpus < 1:
        sys.stderr.write('bench.py: --gpus must be >= 1\n')
        sys.exit(2)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Test aid (never set by the driver): DC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo, so that the whole
    # multi-rank flow of this script (broadcast, flat-bucket all-reduce, barriers, max over ranks, rank-0 JSON) can be
    # exercised on a one-GPU box; RCCL refuses two ranks on one device.
    one_device = os.environ.get('DC_BENCH_ONE_DEVICE') == '1'
    dev_index = 0 if (one_device or world == 1) else local_rank
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s): launch with --nproc-per-node = --gpus\n' % (args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= dev_index:
        sys.stderr.write('bench.py: rank %d needs cuda:%d, %d GPU(s) visible: refusing to run\n'
                         % (rank, dev_index, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        sys.exit(2)
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if one_device:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', dev_index))
    dev = torch.device('cuda', dev_index)
    S, E = args.seq_len, args.epochs
    B = args.batch if args.batch > 0 else (256 if world == 1 else 128)

    if args.secondary_only:
        print(json.dumps(secondary_measurements(args, dev, E, args.parity_ref)))
        sys.stdout.flush()
        return
    if args.extras_only:
        extras = side_measurements(args, dev, B, S, E)
        print(json.dumps(extras))
        sys.stdout.flush()
        return

    hook_factory = None
    if world > 1:
        from dotaclient_amd.distributed import FlatGradAllReducer
        hook_factory = FlatGradAllReducer
    want_cpu = world == 1 and rank == 0 and not args.no_cpu_baseline
    main_run = run_workload(args.cell, args.hidden, args.layers, B, S, E, args.steps, args.warmup, dev, rank, world, hook_factory,
                            want_parity=want_cpu and B * S <= PARITY_FULL_MAX, want_profile=True)
    check_status(main_run, 'the timed workload')
    eng, rollouts = main_run['eng'], main_run['rollouts']
    elapsed, status, losses, regions = main_run['elapsed'], main_run['status'], main_run['losses'], main_run['regions']
    rollout_ms, epochs_ms = main_run['rollout_ms'], main_run['epochs_ms']
    lr, ent, vf = main_run['lr'], main_run['ent'], main_run['vf']

    if rank == 0:
        n_steps = world * B * S * args.steps
        value = n_steps / elapsed
        ffwd = fwd_flops_per_step(args.cell, args.hidden, args.layers)
        workload_key = '%s-%d-%dx%d' % (args.cell, args.hidden, B, S)
        for r in regions:
            if r['kernel'] == 'gradnorm_clip_adam':          # the library does not know the parameter count (segments live on the device): 4 B
                r['bytes'] = 32.0 * eng.total * r['launches']   # read for the norms + 28 B read / written by the update, per parameter
        regions.sort(key=lambda r: -r['total_ms'])
        kernels = []
        for r in regions:
            avg_us = r['total_ms'] * 1e3 / r['launches']
            bound = REGION_BOUND.get(r['kernel'], 'mfma' if r['flops'] > 0 else 'hbm')
            k = {'kernel': r['kernel'], 'bound': bound, 'launches_per_step': r['launches'], 'avg_us': round(avg_us, 3),
                 'ms_per_step': round(r['total_ms'], 3)}
            if bound == 'hbm':
                if r['kernel'] in MASK_DEPENDENT_BYTES:
                    k['achieved_gbs'] = None
                    k['note'] = 'mask-aware: reads only the unmasked units, the byte count depends on the masks (dense form: %.0f MB per launch)' \
                                % (r['bytes'] / r['launches'] / 1e6)
                else:
                    gbs = r['bytes'] / (r['total_ms'] * 1e-3) / 1e9
                    k['achieved_gbs'] = round(gbs, 1)
                    k['frac_of_hbm_peak'] = round(gbs / PEAK_HBM_GBS, 4)
                    # the counters' view of the same launch: a read served by a cache is not HBM traffic, so the HBM rate is
                    # min(algorithmic, counter) bytes / time
                    tr = pmc_traffic(args.traffic_json, r['kernel'].split('(')[0], workload_key)
                    if tr is not None:
                        k['hbm_gbs_by_counter'] = round(tr / (avg_us * 1e-6) / 1e9, 1)
            else:
                tf = r['flops'] / (r['total_ms'] * 1e-3) / 1e12
                k['achieved_tflops'] = round(tf, 3)
                if bound == 'mfma':
                    k['peak_tflops'] = round(mfma_peak(r['kernel'], bool(KERNEL_FLAGS & 4096)), 1)
                    k['frac'] = round(tf / k['peak_tflops'], 4)
                    k['frac_of_f32_mfma_peak'] = round(tf / PEAK_F32_MFMA_TFLOPS, 4)
                else:                      # VALU / latency-bound kernels: against the peak of the pipe their arithmetic runs on
                    k['peak_tflops'] = latency_peak(r['kernel'], bool(KERNEL_FLAGS & 4096), args.hidden)
                    k['frac'] = round(tf / k['peak_tflops'], 4)
            k['traffic'] = pmc_traffic(args.traffic_json, r['kernel'].split('(')[0], workload_key)
            kernels.append(k)
        # the HBM-bound side (GAE / loss / Adam / pooling stream their operands once): largest by time
        hbm = [r for r in regions if REGION_BOUND.get(r['kernel']) == 'hbm' and r['kernel'] not in MASK_DEPENDENT_BYTES]
        roofline_hbm = None
        if hbm:
            hd = hbm[0]
            alg_b = hd['bytes'] / hd['launches']
            tr = pmc_traffic(args.traffic_json, hd['kernel'].split('(')[0], workload_key)
            # bytes that actually crossed the HBM interface: the counters' figure when there is one and it is SMALLER than the
            # algorithmic count (part of the operands came out of a cache: pricing those against the HBM peak would inflate frac)
            used_b = alg_b if tr is None else min(alg_b, tr)
            gbs = used_b / (hd['total_ms'] * 1e-3 / hd['launches']) / 1e9
            roofline_hbm = {'bound': 'hbm', 'kernel': hd['kernel'], 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                            'frac': round(gbs / PEAK_HBM_GBS, 4), 'avg_launch_us': round(hd['total_ms'] * 1e3 / hd['launches'], 3),
                            'algorithmic_bytes_per_launch': alg_b, 'bytes_priced': used_b,
                            'bytes_priced_note': 'min(algorithmic, PMC counter) bytes per launch', 'traffic': tr}
        # `roofline` prices the largest region that HAS a throughput roofline (matrix pipes / vector unit; the HBM-bound kernels have their own
        # object, `roofline_hbm`).  The recurrent team kernels are bound by a serial dependency chain: when one of them is the largest
        # region of all (LSTM-256: lstm_fwd_team and embed_bwd_pool16m are within a few per cent of each other, which one leads depends on the
        # box) it is named in `largest_region` with its time, and priced in `kernels` like every other region.
        priced = [r for r in regions if REGION_BOUND.get(r['kernel'], 'mfma') in ('mfma', 'valu')]
        dom = priced[0] if priced else regions[0]
        dom_bound = REGION_BOUND.get(dom['kernel'], 'mfma')
        achieved = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
        dom_peak = mfma_peak(dom['kernel'], bool(KERNEL_FLAGS & 4096)) if dom_bound == 'mfma' else \
            latency_peak(dom['kernel'], bool(KERNEL_FLAGS & 4096), args.hidden)
        P = eng.total
        alg_step_bytes = 2100.0 * (1 + E) * B * S + 28.0 * P * E            # 2.1 KB per env-step and pass + 28 B per parameter and optimizer step
        step_traffic = pmc_whole_step(args.traffic_json, workload_key, 1 + E)
        bound_note = BOUND_NOTES[dom_bound]
        if dom_bound == 'mfma' and (KERNEL_FLAGS & 4096) and dom['kernel'].startswith('gemm_'):
            bound_note = ('DC_DIMS_BF16: bf16 operands (stored as bf16 on the LSTM-512 path), f32 accumulate, one v_mfma_f32_32x32x16_bf16 per K = 16 and '
                          'tile pair; `peak` = the dense bf16 MFMA peak (2 500 TF)')
        roofline = {'bound': dom_bound, 'bound_note': bound_note, 'kernel': dom['kernel'], 'kernel_note': REGION_NOTES.get(dom['kernel']),
                    'achieved': round(achieved, 3), 'peak': round(dom_peak, 1),
                    'unit': 'TFLOP/s', 'frac': round(achieved / dom_peak, 4),
                    'traffic': pmc_traffic(args.traffic_json, dom['kernel'], workload_key),
                    'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, %s)' % os.path.relpath(args.traffic_json, REPO),
                    'algorithmic_bytes_per_launch': dom['bytes'] / dom['launches'],
                    'avg_launch_us': round(dom['total_ms'] * 1e3 / dom['launches'], 3),
                    'flops_per_launch': dom['flops'] / dom['launches'],
                    'whole_step': {'flops_per_env_step_dense': ffwd * (1 + 3 * E),
                                   'effective_tflops_dense_count': round(value / world * ffwd * (1 + 3 * E) / 1e12, 3),
                                   'hbm_traffic_bytes': None if step_traffic is None else round(step_traffic),
                                   'algorithmic_bytes': round(alg_step_bytes),
                                   'traffic_over_algorithmic': None if step_traffic is None else round(step_traffic / alg_step_bytes, 1),
                                   'note': 'steps/s x the reference\'s DENSE flop count (SURVEY.md 8(d)); the sparse max-pool backward '
                                           'executes 1/16 of the dense MACs of the 16-unit types, so this is an effective rate, not pipe '
                                           'utilisation - per-kernel utilisation is in `kernels`; hbm_traffic_bytes = sum over kernels of '
                                           'PMC bytes x launches per step, algorithmic_bytes = SURVEY.md 8(d)\'s per-unit figures x this batch'},
                    'kernels': kernels}
        if regions[0] is not dom:
            big = regions[0]
            roofline['largest_region'] = {'kernel': big['kernel'], 'bound': REGION_BOUND.get(big['kernel'], 'mfma'),
                                          'bound_note': BOUND_NOTES.get(REGION_BOUND.get(big['kernel'], 'mfma')),
                                          'ms_per_step': round(big['total_ms'], 3), 'avg_launch_us': round(big['total_ms'] * 1e3 / big['launches'], 3),
                                          'priced_region_ms_per_step': round(dom['total_ms'], 3)}
        key = (args.cell, args.hidden, args.layers, B, S, world > 1)
        which = {('lstm', 256, 1, 256, 256, False): 'BASELINE.json configs[2] (5v5 synthetic, LSTM hidden=256, batch=256x256 steps, 1xMI355X)',
                 ('lstm', 256, 1, 128, 256, True): "BASELINE.json configs[3] geometry (5v5 synthetic, LSTM hidden=256, 128 trajectories x 256 "
                                                   'steps per GPU = 1024x256 at DP=8, RCCL gradient all-reduce every epoch)',
                 ('lstm', 256, 1, 128, 256, False): "BASELINE.json configs[3]'s per-GPU shard on one GPU (128 trajectories x 256 steps, no all-reduce)",
                 ('lstm', 128, 1, 64, 256, False): 'BASELINE.json configs[1] (1v1-mid, LSTM hidden=128, batch=64x256 steps)',
                 ('lstm', 512, 2, 256, 512, False): "BASELINE.json configs[4]'s per-GPU shard on one GPU (2-layer LSTM-512, 256 trajectories x 512 steps)",
                 ('gru', 256, 1, 64, 256, False): "configs[1]'s batch with the reference's own cell (GRU-256, policy.py:66)"}.get(
                     key, 'other configuration (not a BASELINE.json bench line)')
        line = {
            'metric': 'env-steps/sec through PPO optimizer', 'value': round(value, 1), 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'bf16' if KERNEL_FLAGS & 4096 else 'f32',
            'dtype_note': ('DC_DIMS_BF16: bf16 operands / f32 accumulate in the dense products and the recurrent products; on the LSTM-512 path the gate '
                           'pre-activations / activations, the gate gradients, `pre`, `hseq` and `hprev` are also STORED as bf16 (round 5; '
                           'DC_DIMS_BF16_F32_STORE keeps them f32); cell state, embeddings, loss, norms, Adam and the master weights stay f32')
                          if KERNEL_FLAGS & 4096 else
                          'f32 end to end: inputs, weights, activations, gradients and optimizer state are f32 and every product is f32-grade '
                          '(matrix products, products=%s: %s; measured against f64 on the network\'s shapes: f16x2 1.2e-7..4.9e-7, bf16x3 1.4e-7..5.5e-7 of '
                          'max |C|, the f32 fma chain 2.1e-7..3.7e-7 - tools/ubench/gemm_x3.hip, profiles/r04/ubench_gemm_f16_pieces.txt); '
                          '`parity` checks the whole step against the fp32 oracle at 1e-4'
                          % (PRODUCTS, 'two f16 pieces per f32 operand (23 of 24 significand bits) with power-of-two pre-scales, three f16 MFMAs (hh, hm, mh), f32 '
                                       'accumulate; an operand outside f16\'s exponent range trips the NaN guard and the consumer loop repeats the '
                                       'iteration with the bf16 pieces' if PRODUCTS == 'f16x2' else
                                       'exact 3-way bf16 splits of both f32 operands, six bf16 MFMAs, f32 accumulate'),
            'data': 'synthetic',
            'config': {'workload': '%s: synthetic trajectories, %s hidden=%d x%d layer, '
                                   'batch=%d trajectories x %d steps per GPU, %d epochs + rollout pass per step'
                                   % (which, args.cell.upper(), args.hidden, args.layers, B, S, E),
                       'cell': args.cell, 'hidden': args.hidden, 'layers': args.layers, 'batch_per_gpu': B,
                       'seq_len': S, 'epochs': E, 'parallelism': 'dp%d' % world, 'epoch_launch': 'hipGraph replay' if USE_GRAPHS else 'eager',
                       'products': PRODUCTS},
            'phases': {'rollout_pass_ms': round(rollout_ms, 3), 'epoch_ms': round(epochs_ms / E, 3),
                       'train_only_env_steps_per_s_per_gpu': round(B * S / (epochs_ms / E * 1e-3), 1),
                       'note': 'one untimed iteration on rank 0: no-grad forward + old log-probs + GAE, then the mean of the %d '
                               'full-batch epochs (forward, loss, backward, clip + Adam); train-only = B*S / epoch time' % E},
            'roofline': roofline,
            'roofline_hbm': roofline_hbm,
            'nan_status': status, 'final_loss': float(losses[0]),
        }
        # The N > 1 lines of this script time 128 trajectories per GPU (configs[3]'s shard); the N = 1 line times configs[2]'s 256.  So
        # that N = 1 and N > 1 compare like with like, the default N = 1 line also carries the one-GPU rate of that 128-trajectory
        # shard (same kernels, no all-reduce), timed the same way (barrier + synchronize around K steps) - a short run, in a try:
        # nothing here may cost the headline.
        line['weak_scaling_unit'] = None
        if world == 1 and args.batch == 0 and not args.no_weak_unit:
            try:
                k_unit, w_unit = min(args.steps, 10), min(args.warmup, 3)
                u = run_workload(args.cell, args.hidden, args.layers, 128, S, E, k_unit, w_unit, dev, 0, 1)
                if u['status'] == 0:
                    ums = u['elapsed'] / k_unit * 1e3
                    line['weak_scaling_unit'] = {
                        'workload': "BASELINE.json configs[3]'s per-GPU shard on ONE GPU (128 trajectories x %d steps, %s-%d, no all-reduce): the "
                                    'like-for-like N = 1 point for the N > 1 lines of this script (which time 128 trajectories per GPU)'
                                    % (S, args.cell.upper(), args.hidden),
                        'value': round(128 * S / (ums * 1e-3), 1), 'unit': 'env-steps/s', 'ms_per_step': round(ums, 3),
                        'steps': k_unit, 'warmup': w_unit, 'batch_per_gpu': 128}
                del u
            except Exception as e:                                  # noqa: BLE001
                sys.stderr.write('bench.py: weak_scaling_unit failed: %r\n' % (e,))
        if want_cpu:
            bf16 = bool(KERNEL_FLAGS & 4096)
            tol, amin = (1e-2, 0.995) if bf16 else (1e-4, None)      # bf16 path: the stated tolerances of tests/test_gpu_bf16.py (round 6: 3e-2 / 0.97 before)
            if B * S > PARITY_FULL_MAX:
                # a batch the oracle would need minutes (and tens of GB) for: checker and CPU baseline run on a bounded sample, the first
                # 64 trajectories, which the HIP path runs again as a batch of its own (first iteration from the same initial weights)
                nsub = min(64, B)
                sub = rollouts[:nsub]
                eng2 = Engine(args.cell, args.hidden, args.layers, dev)
                eng2.kernel_flags = KERNEL_FLAGS
                eng2.products = PRODUCTS
                eng2.load_state_dict(synth.init_state_dict(7, args.cell, args.hidden, args.layers))
                got = hip_parity_iteration(eng2, pack_rollouts(sub, S, dev), S, E, lr, ent, vf)
                st2 = int(eng2.status.item())
                del eng2
                line['cpu_baseline'], ref = cpu_baseline(args.cell, args.hidden, args.layers, sub, S, E, lr, ent, vf)
                line['cpu_baseline']['sample'] += '; BOUNDED SAMPLE: the first %d of the %d trajectories of the timed batch' % (nsub, B)
                line['parity'] = parity_report(got, ref, tol, amin, sub_batch=nsub)
                line['parity']['status_of_the_sub_batch_run'] = st2
            else:
                line['cpu_baseline'], ref = cpu_baseline(args.cell, args.hidden, args.layers, rollouts, S, E, lr, ent, vf)
                line['parity'] = parity_report(main_run['first_iteration'], ref, tol, amin)
        else:
            line['cpu_baseline'] = None
            line['parity'] = None
            ref = None
        # the other configurations + the fallback products, in a subprocess (a crash there costs these two blocks, not the headline)
        line['secondary'] = line['products_fallback'] = None
        default_family = (args.cell, args.hidden, args.layers, S, args.batch, KERNEL_FLAGS, PRODUCTS) == ('lstm', 256, 1, 256, 0, 0, 'f16x2')
        if world == 1 and want_cpu and default_family and not args.no_secondary:      # --no-cpu-baseline = the headline only (A/B and profiled runs)
            sec = run_secondary_subprocess(args, ref if (want_cpu and B * S <= PARITY_FULL_MAX) else None)
            line['secondary'], line['products_fallback'] = sec.get('secondary'), sec.get('products_fallback')
        print(json.dumps(line))
        sys.stdout.flush()                      # the headline is out before anything optional runs
        if line['parity'] is not None and not line['parity']['ok']:
            sys.stderr.write('bench.py: PARITY FAILED against the oracle: %s\n' % json.dumps(line['parity']))
            sys.exit(3)
    if world > 1:
        torch.distributed.destroy_process_group()
    elif args.extras and rank == 0:
        run_extras_subprocess(args)


# ---- the default line's `secondary` and `products_fallback` blocks ---------------------------------------
SECONDARY_STEPS, SECONDARY_WARMUP = 10, 3
GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')


def _slim_parity(p):
    return {k: p[k] for k in ('checker', 'parity_rel_err', 'tolerance', 'argmax_bit_exact', 'argmax_equal_fraction', 'argmax_rule', 'ok',
                              'per_quantity', 'elementwise_rel_err')}


def _golden_as_ref(path):
    """A tests/golden/*.npz of the REAL reference (tests/golden/make_golden.py: one epoch) in the shape `parity_report` compares."""
    g = np.load(path)
    ref = {'advantages': g['advantages'].ravel(), 'returns': g['returns'].ravel(), 'values': g['values'].ravel(),
           'argmax': g['argmax'].reshape(-1, 5), 'param_samples': g['ep0_param_samples'],
           'epochs': np.concatenate([g['ep0_losses'], g['ep0_entropies'], g['ep0_grad_norms']])[None].astype(np.float64)}
    for k in g.files:
        if k.startswith('old_logp_'):
            ref[k] = g[k]
    return ref


def _cfg4_fixture_pair(got, path):
    """(got, ref) restricted to the strided samples tests/golden/cfg4_shard_oracle.npz holds (tests/golden/make_cfg4_fixture.py)."""
    f = np.load(path)
    stride = int(f['stride'])
    ref = {'returns': f['returns'], 'argmax': f['argmax_rows16'].astype(np.int64), 'param_samples': f['ep0_param_samples'],
           'epochs': np.concatenate([f['ep0_losses'], f['ep0_entropies'], f['ep0_grad_norms']])[None].astype(np.float64)}
    sub = {'returns': got['returns'].ravel()[::stride], 'argmax': got['argmax'].reshape(-1, 5)[::16], 'param_samples': got['param_samples'],
           'epochs': got['epochs'][:1]}
    for k in ['advantages', 'values'] + [k for k in f.files if k.startswith('old_logp_') and not k.endswith(('_max', '_n'))]:
        ref[k] = f[k]
        sub[k] = np.asarray(got[k]).ravel()[::stride]
    return sub, ref


def secondary_measurements(args, dev, E, parity_ref_path):
    """The other BASELINE.json configurations and the reference's own cell / default shape, each SECONDARY_STEPS timed steps (barrier +
    synchronize around them, like the headline) with a parity check of the first iteration, plus the headline workload on the fallback
    products (bf16x3).  Runs in a subprocess of the default N = 1 line, before the line is printed: whatever happens here costs these
    two blocks only."""
    global KERNEL_FLAGS, PRODUCTS
    lr, ent, vf = 5e-5, 5e-4, 0.5
    S = 256
    steps, warmup = SECONDARY_STEPS, SECONDARY_WARMUP
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else 16))
    rng = np.random.Generator(np.random.PCG64(99))
    lens, n_chunks = [], 0
    while n_chunks < 1024:                   # the reference's defaults (optimizer.py:776-794): seq_len 16, whole rollouts until >= 1024 chunks
        t = int(rng.integers(100, 900))
        lens.append(t)
        n_chunks += (t + 15) // 16
    work = [
        ('configs[1]', 'BASELINE.json configs[1]: 1v1-mid synthetic, LSTM-128, 64 trajectories x 256 steps',
         dict(cell='lstm', hidden=128, layers=1, b=64, S=S), 'oracle'),
        ('reference_gru256_64x256', "configs[1]'s batch on the reference's OWN cell (GRU-256, policy.py:66)",
         dict(cell='gru', hidden=256, layers=1, b=64, S=S), 'golden'),
        ('reference_defaults_gru256_s16_ragged', "the reference's own defaults (optimizer.py:776-794): GRU-256, seq_len 16, %d ragged rollouts "
         '(100..899 steps) = %d chunks of 16 = %d env-steps incl. padded steps (counted like optimizer.py:486)' % (len(lens), n_chunks, n_chunks * 16),
         dict(cell='gru', hidden=256, layers=1, b=len(lens), S=16, lengths=lens), 'oracle'),
        ('configs[4]_shard_bf16', "BASELINE.json configs[4]'s per-GPU shard on ONE GPU: 2-layer LSTM-512, 256 trajectories x 512 steps, bf16 "
         'path (DC_DIMS_BF16), no all-reduce', dict(cell='lstm', hidden=512, layers=2, b=256, S=512, flags=4096, data_seed=4242), 'cfg4_fixture'),
    ]
    out = {'secondary': {}, 'products_fallback': None}
    for key, what, w, checker in work:
        try:
            KERNEL_FLAGS, PRODUCTS = w.get('flags', 0), 'f16x2'
            bf16 = bool(KERNEL_FLAGS & 4096)
            r = run_workload(w['cell'], w['hidden'], w['layers'], w['b'], w['S'], E, steps, warmup, dev, 0, 1, want_parity=True,
                             lengths=w.get('lengths'), data_seed=w.get('data_seed', 1000), parity_epochs=None if checker == 'oracle' else 1)
            rows = r['batch'].rows
            ms = r['elapsed'] / steps * 1e3
            e = {'workload': what, 'value': round(rows / (ms * 1e-3), 1), 'unit': 'env-steps/s', 'ms_per_step': round(ms, 3), 'steps': steps,
                 'warmup': warmup, 'env_steps_per_step': rows, 'epochs': E, 'dtype': 'bf16' if bf16 else 'f32', 'products': 'bf16' if bf16 else PRODUCTS,
                 'nan_status': r['status']}
            got = r['first_iteration']
            if checker == 'oracle':
                _, _, _, ref = oracle_iteration(w['cell'], w['hidden'], w['layers'], r['rollouts'], w['S'], E, lr, ent, vf, keep=True)
                e['parity'] = _slim_parity(parity_report(got, ref, 1e-4))
            elif checker == 'golden':
                e['parity'] = _slim_parity(parity_report(got, _golden_as_ref(os.path.join(GOLDEN_DIR, 'cfg2_gru_64x256.npz')), 1e-4,
                                                         checker='tests/golden/cfg2_gru_64x256.npz: outputs of the REAL reference (optimizer.py:328-430,581-689 '
                                                                 'imported from /root/reference by tests/golden/make_golden.py) on the same seeded 16 384 env-steps, '
                                                                 'rollout pass + one epoch from the same initial weights'))
            else:
                sub, ref = _cfg4_fixture_pair(got, os.path.join(GOLDEN_DIR, 'cfg4_shard_oracle.npz'))
                e['parity'] = _slim_parity(parity_report(sub, ref, 1e-2, 0.995,
                                                         checker='tests/golden/cfg4_shard_oracle.npz: the fp32 oracle (oracle/ref_optimizer.py; no reference '
                                                                 'exists for this cell, SURVEY.md 8(c)) on the same seeded 131 072 env-steps, strided samples, rollout '
                                                                 'pass + one epoch; bf16 path judged by its STATED tolerance (tests/test_gpu_bf16.py)'))
            out['secondary'][key] = e
            del r
        except Exception as ex:                                  # noqa: BLE001 - one workload's failure is reported, not raised
            out['secondary'][key] = {'workload': what, 'error': repr(ex)}
        torch.cuda.empty_cache()
    # the headline workload on the products the consumer loop falls back to when an operand leaves f16's range (Engine.use_safe_products)
    try:
        KERNEL_FLAGS, PRODUCTS = 0, 'bf16x3'
        B = args.batch if args.batch > 0 else 256
        r = run_workload(args.cell, args.hidden, args.layers, B, args.seq_len, E, steps, warmup, dev, 0, 1, want_parity=bool(parity_ref_path))
        ms = r['elapsed'] / steps * 1e3
        fb = {'products': 'bf16x3', 'what': 'the timed workload of this line on the fallback products (three bf16 pieces per f32 operand, six MFMAs, '
                                            "f32's exponent range): what the consumer loop switches to when an operand leaves f16's range",
              'value': round(r['batch'].rows / (ms * 1e-3), 1), 'unit': 'env-steps/s', 'ms_per_step': round(ms, 3), 'steps': steps, 'warmup': warmup,
              'nan_status': r['status']}
        if parity_ref_path:
            z = np.load(parity_ref_path)
            fb['parity'] = _slim_parity(parity_report(r['first_iteration'], {k: z[k] for k in z.files}, 1e-4))
        out['products_fallback'] = fb
    except Exception as ex:                                      # noqa: BLE001
        out['products_fallback'] = {'products': 'bf16x3', 'error': repr(ex)}
    return out


def run_secondary_subprocess(args, ref):
    """`secondary_measurements` in a process of its own (timeout 600 s); returns its dict, or one that says what went wrong."""
    import subprocess
    import tempfile
    ref_path = ''
    tmp = None
    if ref is not None:
        tmp = tempfile.NamedTemporaryFile(suffix='.npz', delete=False)
        tmp.close()
        np.savez(tmp.name, **ref)
        ref_path = tmp.name
    cmd = [sys.executable, os.path.abspath(__file__), '--secondary-only', '--cell', args.cell, '--hidden', str(args.hidden), '--layers', str(args.layers),
           '--batch', str(args.batch), '--seq-len', str(args.seq_len), '--epochs', str(args.epochs), '--parity-ref', ref_path]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        last = (r.stdout.strip().splitlines() or [''])[-1]
        if r.returncode == 0 and last.startswith('{'):
            return json.loads(last)
        return {'secondary': {'error': 'subprocess rc %d: %s' % (r.returncode, r.stderr[-400:])}, 'products_fallback': None}
    except Exception as ex:                                      # noqa: BLE001 - nothing here may cost the headline
        return {'secondary': {'error': repr(ex)}, 'products_fallback': None}
    finally:
        if tmp is not None:
            try:
                os.unlink(tmp.name)
            except OSError:
                pass


def run_extras_subprocess(args):
    """The side measurements in a process of their own: whatever happens there, this process has already printed its line and exits 0."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--extras-only', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--cell', args.cell, '--hidden', str(args.hidden), '--layers', str(args.layers), '--batch', str(args.batch),
           '--seq-len', str(args.seq_len), '--epochs', str(args.epochs), '--kernel-flags', str(args.kernel_flags), '--products', args.products]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        out = (r.stdout.strip().splitlines() or [''])[-1]
        if r.returncode == 0 and out.startswith('{'):
            os.makedirs(os.path.dirname(args.extras_out), exist_ok=True)
            with open(args.extras_out, 'w') as f:
                f.write(out + '\n')
            sys.stderr.write('bench.py extras: %s\n' % out)
        else:
            sys.stderr.write('bench.py extras: subprocess failed (rc %d): %s\n' % (r.returncode, r.stderr[-600:]))
    except Exception as e:                                  # noqa: BLE001 - nothing here may cost the exit code
        sys.stderr.write('bench.py extras: %r\n' % (e,))


def side_measurements(args, dev, B, S, E):
    """Beside the headline, never inside it (single GPU): host -> device ingest serial and through the consumer loop's own
    prefetcher, model publish, the epochs replayed as hipGraphs, the first epoch on the rollout pass's activations, and the other
    BASELINE.json single-GPU configurations.  Every workload's status word is checked."""
    global USE_GRAPHS, REUSE_FORWARD
    import io
    out = {'steps': args.steps, 'warmup': args.warmup}
    default_family = (args.cell, args.hidden, args.layers, S) == ('lstm', 256, 1, 256) and args.batch == 0

    def timed(cell, hidden, layers, b, lengths=None, seq_len=S):
        r = run_workload(cell, hidden, layers, b, seq_len, E, args.steps, args.warmup, dev, 0, 1, lengths=lengths)
        check_status(r, 'side workload %s-%d x%d B=%d S=%d graphs=%s reuse=%s' % (cell, hidden, layers, b, seq_len, USE_GRAPHS, REUSE_FORWARD))
        ms = r['elapsed'] / args.steps * 1e3
        eng, rollouts = r['eng'], r['rollouts']
        return ms, eng, rollouts

    base_ms, eng, rollouts = timed(args.cell, args.hidden, args.layers, B)
    out['eager_ms_per_step'] = round(base_ms, 3)

    # ---- ingest: wire-format dicts -> pinned staging -> HBM -------------------------------------------------------------------------
    for _ in range(2):
        pack_rollouts(rollouts, S, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        pack_rollouts(rollouts, S, dev)
    torch.cuda.synchronize()
    ingest_ms = (time.perf_counter() - t0) / 3 * 1e3
    # ... and the way the consumer loop (DotaOptimizer.run_iteration, prefetch=True) runs it: while the GPU works on the current batch the
    # host packs the NEXT one rollout by rollout (engine.IncrementalPacker: pinned staging, H2D on its own stream) - the timed steps
    # again with that going on beside them
    from dotaclient_amd.engine import IncrementalPacker
    packer = IncrementalPacker(S, dev, expected_rows=B * S)
    step_fn, lr_, ent_, vf_ = None, 5e-5, 5e-4, 0.5
    batch0 = pack_rollouts(rollouts, S, dev)

    def one_step(batch):
        chunks = eng.rollout_pass(batch, S)
        for _ in range(E):
            eng.train_epoch(chunks, lr_, ent_, vf_)

    def prefetch_next():
        for d in rollouts:
            packer.add(d)
        return packer.finish()
    cur = batch0
    for _ in range(2):
        one_step(cur); cur = prefetch_next()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(cur)                    # enqueued; the GPU is busy for ~a step
        cur = prefetch_next()            # host packs + H2D of the next batch meanwhile
    torch.cuda.synchronize()
    overlap_ms = (time.perf_counter() - t0) / args.steps * 1e3
    out['ingest'] = {'pack_h2d_ms_per_batch': round(ingest_ms, 3),
                     'env_steps_per_s_with_ingest_serialised': round(B * S / ((base_ms + ingest_ms) * 1e-3), 1),
                     'ms_per_step_with_next_batch_prefetched': round(overlap_ms, 3),
                     'env_steps_per_s_with_ingest_prefetched': round(B * S / (overlap_ms * 1e-3), 1),
                     'note': 'wire-format dicts -> page-locked staging (dc_pack_rows) -> HBM; serialised = engine.pack_rollouts then the step; '
                             'prefetched = the consumer loop\'s form (DotaOptimizer.run_iteration with prefetch: engine.IncrementalPacker packs the '
                             'next batch rollout by rollout on its own stream while the GPU works on the current one; a fresh batch every step); '
                             'not part of `value`'}
    del batch0, cur

    # ---- model publish (optimizer.py:697-716, once per iteration) ---------------------------------------------------------------
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
    out['publish'] = {'ms_per_publish': {k: round(float(np.median(v)), 3) for k, v in samples.items()},
                      'note': 'D2H + torch.save of the 34-tensor state_dict; flat_snapshot = one asynchronous copy of the flat buffer into '
                              'page-locked memory (Engine.start_param_snapshot), per_tensor_copies = the reference\'s form'}
    del eng
    torch.cuda.empty_cache()

    # ---- (f4) actor-side inference: one env-step of one hero (agent.py:652 -> policy.py:80-84), CPU observation tensors in, value read back --
    try:
        from dotaclient_amd.policy import Policy
        from dotaclient_amd import layout as L_
        lat = {}
        for cell_, hid_ in (('gru', 256), (args.cell, args.hidden)):
            pol = Policy(cell_, hid_, args.layers if cell_ == args.cell else 1, dev)
            r0 = rollouts[0]
            for mode in ('one_kernel', 'graph_replay', 'eager'):
                pol.single_kernel, pol.single_graph = mode == 'one_kernel', mode == 'graph_replay'
                h = pol.init_hidden()
                for t in range(4):
                    _, _, h = pol.single(**{k: r0['observations'][k][t] for k in L_.INPUT_KEYS}, hidden=h)
                torch.cuda.synchronize()
                ts = []
                for t in range(100):
                    t0 = time.perf_counter()
                    _, v, h = pol.single(**{k: r0['observations'][k][t % S] for k in L_.INPUT_KEYS}, hidden=h)
                    v.cpu()
                    ts.append((time.perf_counter() - t0) * 1e6)
                lat['%s-%d %s' % (cell_, hid_, mode)] = round(float(np.median(ts)), 1)
            del pol
        out['actor_single_step_latency_us'] = dict(lat, note='Policy.single, B = 1, S = 1, CPU observation tensors in, value read back on the host every step '
                                                             '(median of 100): one_kernel = the whole step as ONE kernel reading the pinned observation row in place '
                                                             '(csrc/policy_single.hip, Policy.single_kernel, the default since round 6); graph_replay = the batch path\'s '
                                                             'launches replayed as ONE hipGraph over static buffers (rounds 4-5); eager = the same kernels launched one by one')
    except Exception as e:                                  # noqa: BLE001
        out['actor_single_step_latency_us'] = {'error': repr(e)}
    torch.cuda.empty_cache()

    # ---- epochs as hipGraph replays; first epoch on the rollout pass's activations ---------------------------------------------------
    USE_GRAPHS = True
    g_ms, _, _ = timed(args.cell, args.hidden, args.layers, B)
    USE_GRAPHS = False
    torch.cuda.empty_cache()
    REUSE_FORWARD = True
    r_ms, _, _ = timed(args.cell, args.hidden, args.layers, B)
    REUSE_FORWARD = False
    torch.cuda.empty_cache()
    out['epoch_graph'] = {'eager_ms_per_step': round(base_ms, 3), 'graph_replay_ms_per_step': round(g_ms, 3),
                          'first_epoch_reuses_rollout_forward_ms_per_step': round(r_ms, 3),
                          'note': 'Engine.train_epoch(graph=True): an epoch captured once and replayed as ONE hipGraph launch (kernel nodes only: '
                                  'csrc/fill.hip); Engine.reuse_rollout_forward: epoch 0 back-propagates the rollout pass\'s activations (four '
                                  'forward passes per step instead of five) - opt-in, `value` counts the reference\'s five'}

    # ---- the other BASELINE.json single-GPU configurations ---------------------------------------------------------------------------
    if default_family:
        sec = {}
        for key, (c, h, b, what) in {
                'weak_scaling_unit': ('lstm', 256, 128, "configs[3]'s per-GPU shard (128 trajectories x 256 steps, LSTM-256) on ONE GPU, "
                                                        'no all-reduce: the N = 1 reference point for the N > 1 lines of this script'),
                'configs[1]': ('lstm', 128, 64, 'BASELINE.json configs[1]: 1v1-mid, LSTM-128, 64 trajectories x 256 steps'),
                'reference_gru256_64x256': ('gru', 256, 64, "configs[1]'s batch on the reference's own cell (GRU-256, policy.py:66)")}.items():
            ms, _, _ = timed(c, h, 1, b)
            sec[key] = {'workload': what, 'value': round(b * S / (ms * 1e-3), 1), 'unit': 'env-steps/s', 'ms_per_step': round(ms, 3)}
            torch.cuda.empty_cache()
        # the reference's production shape (optimizer.py:776-794 defaults: seq_len 16, whole rollouts until >= 1024 chunks, its own GRU-256)
        rng = np.random.Generator(np.random.PCG64(99))
        lens, chunks = [], 0
        while chunks < 1024:
            t = int(rng.integers(100, 900))
            lens.append(t)
            chunks += (t + 15) // 16
        ms, _, _ = timed('gru', 256, 1, len(lens), lengths=lens, seq_len=16)
        sec['reference_defaults_gru256_s16_ragged'] = {
            'workload': "the reference's own defaults: GRU-256, seq_len 16, %d ragged rollouts (100..899 steps) = %d chunks of 16 = %d env-steps"
                        % (len(lens), chunks, chunks * 16),
            'value': round(chunks * 16 / (ms * 1e-3), 1), 'unit': 'env-steps/s', 'ms_per_step': round(ms, 3)}
        out['secondary'] = sec
    return out


if __name__ == '__main__':
    main()
