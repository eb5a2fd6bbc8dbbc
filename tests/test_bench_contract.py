"""CPU: the parts of bench.py's contract that do not need a GPU (every builder run used other flags than the driver;
`traffic` was null for the team kernels because the region -> kernel map named kernels that do not exist)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

TRAFFIC = os.path.join(REPO, 'profiles', 'pmc_traffic_latest.json')


def test_defaults_are_the_drivers_command_line(monkeypatch):
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 5)
    assert not a.extras and not a.extras_only          # nothing optional runs unless asked for
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-secondary', '--no-host-extras'])
    bench.parse_args()                                  # round-2 flags are still accepted (and ignored)


def test_every_timed_region_finds_its_kernel_in_the_pmc_summary():
    j = json.load(open(TRAFFIC))
    assert j['meta']['workload'] == 'lstm-256-256x256'
    for region in ['embed_fwd_fused', 'embed_bwd_pool16m', 'lstm_fwd_team', 'lstm_bwd_team', 'gemm_f32_dW', 'gemm_f32_fwd', 'gemm_f32_dX',
                   'embed_bwd_small', 'embed_scatter_bwd', 'attn_logits', 'attn_bwd_q']:      # (pool_env_fwd: folded into embed_fwd_fused in round 6)
        t = bench.pmc_traffic(TRAFFIC, region, 'lstm-256-256x256')
        assert isinstance(t, int) and t > 0, region
    assert bench.pmc_traffic(TRAFFIC, 'embed_fwd_fused', 'gru-256-64x256') is None      # a summary of another workload is not used


def test_whole_step_traffic_is_computed_from_the_summary():
    step = bench.pmc_whole_step(TRAFFIC, 'lstm-256-256x256', 5)
    assert 1.5e10 < step < 1e11                         # 29.7 GB per configs[2] step at the end of round 6 (round 3: ~44)
    assert bench.pmc_whole_step(TRAFFIC, 'other', 5) is None


def test_flop_model_matches_the_survey():
    # forward FLOPs per env-step
    assert bench.fwd_flops_per_step('gru', 256, 1) == 2768640
    assert bench.fwd_flops_per_step('lstm', 128, 1) == 2336000
    assert bench.fwd_flops_per_step('lstm', 256, 1) == 3030784
    assert bench.fwd_flops_per_step('lstm', 512, 2) == 9401088


def test_gpus_n_without_enough_devices_refuses_instead_of_mislabelling():
    """a plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must either start 2 ranks or fail
    loudly - never run one rank and print n_gpus 1.  This container has no GPU: the self-launcher refuses with exit code 2 and
    prints no JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'DC_BENCH_ONE_DEVICE')}
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return                                           # a real multi-GPU box: covered by the GPU tests
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert r.stdout.strip() == ''
    assert 'refusing' in r.stderr


def test_rank_count_must_match_gpus(monkeypatch):
    """Under a launcher whose world size differs from --gpus the script stops (exit 2) before touching a device."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout.strip() == ''
    assert 'launcher started 1 rank' in r.stderr


def test_self_launch_command_line(monkeypatch):
    """What the self-launcher executes: torch.distributed.run, one node, --nproc-per-node = --gpus, loopback rendezvous, this file and
    the caller's own arguments."""
    import subprocess
    seen = {}
    monkeypatch.setattr(subprocess, 'call', lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setenv('DC_BENCH_ONE_DEVICE', '1')
    import torch
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1'])
    a = bench.parse_args()
    assert bench.self_launch(a) == 0
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '2' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '2', '--steps', '3', '--warmup', '1'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # without the test aid two ranks on a one-GPU box are refused
    monkeypatch.delenv('DC_BENCH_ONE_DEVICE')
    assert bench.self_launch(a) == 2


def test_secondary_checkers_read_the_committed_fixtures():
    # the default line's `secondary` block checks the reference's GRU-256 against the REAL reference's golden outputs and configs[4]'s
    # shard against the oracle fixture: both files travel with the repo, and their shapes are what parity_report compares
    import numpy as np
    ref = bench._golden_as_ref(os.path.join(bench.GOLDEN_DIR, 'cfg2_gru_64x256.npz'))
    assert ref['advantages'].shape == (16384,) and ref['argmax'].shape == (16384, 5) and ref['epochs'].shape == (1, 11)
    p = bench.parity_report(ref, ref, 1e-4, checker='self')
    assert p['ok'] and p['parity_rel_err'] == 0.0 and p['checker'] == 'self' and p['argmax_bit_exact']
    f = np.load(os.path.join(bench.GOLDEN_DIR, 'cfg4_shard_oracle.npz'))
    got = {k: np.zeros(int(f[k + '_n']), np.float32) for k in ['advantages', 'returns', 'values'] + ['old_logp_' + h for h in ('enum', 'x', 'y', 'target_unit', 'ability')]}
    got.update(argmax=np.zeros((131072, 5), np.int32), param_samples=np.zeros_like(f['ep0_param_samples']), epochs=np.zeros((4, 11)))
    sub, r2 = bench._cfg4_fixture_pair(got, os.path.join(bench.GOLDEN_DIR, 'cfg4_shard_oracle.npz'))
    for k in r2:
        assert np.asarray(sub[k]).shape == np.asarray(r2[k]).shape, k
    rep = bench.parity_report(sub, r2, 3e-2, 0.97)
    assert not rep['ok'] and 'returns' not in rep['per_quantity']       # zeros are not the oracle's outputs
    assert bench.SECONDARY_STEPS == 10


def test_committed_driver_line_carries_the_contract_keys():
    # the line the driver's command printed at the final HEAD (profiles/r05/v14_bench.json, one MI355X): every key of the bench contract, the
    # two extra objects, the secondary workloads with their parity verdicts, and a roofline priced against a throughput peak
    import json
    path = os.path.join(REPO, 'profiles', 'r05', 'v14_bench.json')
    line = json.loads([l for l in open(path) if l.startswith('{')][0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline', 'parity', 'secondary', 'products_fallback'):
        assert k in line, k
    assert line['n_gpus'] == 1 and line['steps'] == 20 and line['warmup'] == 5 and line['higher_is_better'] is True and line['vs_baseline'] is None
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert abs(line['value'] - 256 * 256 / (line['ms_per_step'] * 1e-3)) / line['value'] < 1e-3
    r = line['roofline']
    assert r['bound'] in ('mfma', 'hbm', 'valu') and 0 < r['frac'] < 1 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['traffic'] is not None and r['whole_step']['hbm_traffic_bytes'] < 40e9          # <= 40 GB per step
    if 'largest_region' in r:                                                               # a latency-bound recurrence may lead by a hair
        assert r['largest_region']['bound'] == 'latency' and r['largest_region']['ms_per_step'] >= r['largest_region']['priced_region_ms_per_step']
    c = line['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    assert line['parity']['ok'] is True and line['parity']['parity_rel_err'] < 1e-4
    assert set(line['secondary']) >= {'configs[1]', 'reference_gru256_64x256', 'reference_defaults_gru256_s16_ragged', 'configs[4]_shard_bf16'}
    for k, v in line['secondary'].items():
        assert v['parity']['ok'] is True, k
    assert line['products_fallback']['parity']['ok'] is True
