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
    for region in ['embed_fwd_fused', 'embed_bwd_pool16', 'lstm_fwd_team', 'lstm_bwd_team', 'gemm_f32_dW', 'gemm_f32_fwd', 'gemm_f32_dX',
                   'embed_bwd_dw1', 'embed_bwd_dw2', 'pool_env_fwd']:
        t = bench.pmc_traffic(TRAFFIC, region, 'lstm-256-256x256')
        assert isinstance(t, int) and t > 0, region
    assert bench.pmc_traffic(TRAFFIC, 'embed_fwd_fused', 'gru-256-64x256') is None      # a summary of another workload is not used


def test_whole_step_traffic_is_computed_from_the_summary():
    step = bench.pmc_whole_step(TRAFFIC, 'lstm-256-256x256', 5)
    assert 3e10 < step < 1e11                           # ~56 GB per configs[2] step
    assert bench.pmc_whole_step(TRAFFIC, 'other', 5) is None


def test_flop_model_matches_the_survey():
    # forward FLOPs per env-step
    assert bench.fwd_flops_per_step('gru', 256, 1) == 2768640
    assert bench.fwd_flops_per_step('lstm', 128, 1) == 2336000
    assert bench.fwd_flops_per_step('lstm', 256, 1) == 3030784
    assert bench.fwd_flops_per_step('lstm', 512, 2) == 9401088
