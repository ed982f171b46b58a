"""The bench contract on a GPU box: one JSON line with the required keys at N = 1, and the one-process-per-GPU path
(sharded chains, max-over-ranks time, summed leapfrogs) exercised with two ranks sharing the box's GPU over gloo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
        'vs_baseline', 'dtype', 'data', 'config', 'roofline')


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '2', '--warmup', '2', '--chains', '512', '--no-cpu-baseline', '--no-configs'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    for k in KEYS:
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 2 and j['warmup'] == 2 and j['value'] > 0 and j['dtype'] == 'f64'
    rf = j['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12
    assert 'workload' in j['config'] and 'model' not in j['config']
    assert 'extras_error' not in j, j.get('extras_error')
    # (executed flops: never more than the algorithm's, except where a kernel runs the decay term's product although it is the bound's)
    assert abs(rf['frac'] - rf['frac_algorithmic'] * rf['executed_share_of_algorithmic_flops']) < 1e-9 * rf['frac']
    assert rf['executed_share_of_algorithmic_flops'] <= 1. + 1e-12 or not rf['kernel'].startswith(('bf_nuts_pipe_kernel', 'bf_lone_kernel'))
    assert 'config 3' in j['config']['workload'] and 'banana' in j['config']['workload']
    # the line is SURVEY 8d's config 3: both rounds, the refit between them, and value = their leapfrogs over their sampling time
    r0, r1 = j['config3_round0'], j['config3_round1']
    assert j['config']['timed_launches_round0'] == 1 and j['config']['timed_launches_round1'] == 1
    lf, t = r0['leapfrogs_timed'] + r1['leapfrogs_timed'], r0['wall_s_timed'] + r1['wall_s_timed']
    assert abs(j['value'] - lf / t) < 1e-9 * j['value'] and abs(j['ms_per_step'] - t / 2 * 1e3) < 1e-6
    assert j['refit']['n_fit_points'] > 4000 and j['refit']['fit_1_ms'] > 0 and j['refit']['select_ms'] > 0
    assert 'Sobol-normal' in r0['workload'] and rf['kernel'] in (r0['roofline']['kernel'], r1['roofline']['kernel'])
    for k in ('gauss64_best_case', 'hetero', 'scaled_inputs', 'refit_cycle', 'full_metric', 'tempered'):   # the side blocks
        assert k in j, k
    assert j['full_metric']['value'] > 0 and j['tempered']['value'] > 0 and j['refit_cycle']['total_ms'] > 0
    assert j['gauss64_best_case']['value'] > 0


def test_bench_two_ranks_on_one_gpu_gloo():
    port = 29700 + os.getpid() % 200
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '2', '--chains', '256',
           '--backend', 'gloo', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = _line(r.stdout)
    assert j['n_gpus'] == 2 and j['scaling'] == 'weak' and j['value'] > 0
    assert j['config']['chains_per_gpu'] == 256
    ex = j['refit_exchange']   # the path's one exchange step, timed on its own: four collectives, the same rows on every rank
    assert ex['collectives'] == 4 and ex['identical_on_all_ranks'] and ex['ms'] > 0 and ex['rows_selected'] == 4290
    assert ex['wire_bytes_per_rank'] < 256 * 100 * 65 * 8 / 2   # below the shard's samples
    assert j['distributed']['backend'] == 'gloo' and j['distributed']['world_size'] == 2
    assert j['layout_vote']['ms_per_launch'] > 0   # (what a sharded sample() pays per launch to agree on the next layout)


def test_bench_eight_ranks_rehearsal_on_one_gpu_gloo():
    """The launch the driver makes on an 8-GPU node, rehearsed on one GPU: eight ranks over gloo, each with its shard of the
    chains (stream = global chain index), the barrier-bracketed timing with the maximum over the ranks, the summed leapfrog
    count, the refit's exchange (the same 2 P rows on all eight ranks) and the cost of the per-launch layout vote."""
    port = 29300 + os.getpid() % 150
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
           '--master-port', str(port), 'bench.py', '--gpus', '8', '--steps', '1', '--warmup', '1', '--chains', '128',
           '--backend', 'gloo', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = _line(r.stdout)
    assert j['n_gpus'] == 8 and j['scaling'] == 'weak' and j['value'] > 0 and j['config']['chains_per_gpu'] == 128
    ex = j['refit_exchange']
    assert ex['collectives'] == 4 and ex['identical_on_all_ranks'] and ex['rows_selected'] == 4290 and ex['rows_per_rank'] == 128 * 100
    assert j['layout_vote']['ms_per_launch'] > 0
    assert 'cpu_baseline' not in j or j['cpu_baseline'] is None


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` without a launcher starts two ranks (never one rank labelled n_gpus 1), and refuses RCCL when the
    box has fewer GPUs than ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--chains', '256', '--backend', 'gloo',
                        '--no-cpu-baseline'], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = _line(r.stdout)
    assert j['n_gpus'] == 2 and j['refit_exchange']['collectives'] == 4 and j['refit_exchange']['identical_on_all_ranks']
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1', '--warmup', '1'], cwd=ROOT, capture_output=True,
                           text=True, timeout=300, env=env)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith('{')]
    # a launcher whose world size disagrees with --gpus is an error, not a mislabelled line
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--steps', '1', '--warmup', '1'], cwd=ROOT, capture_output=True,
                       text=True, timeout=300, env=dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'))
    assert r.returncode != 0


@pytest.mark.parametrize('name', ['gauss32', 'banana_decay', 'funnel', 'cubic128', 'des_pipeline'])
def test_bench_config_blocks(name):
    """The blocks on the BASELINE configs' own targets (bench.py: config_block), at a reduced chain count: each prints its
    rate, tree statistics, divergence rate and its own roofline."""
    r = subprocess.run([sys.executable, 'bench.py', '--workload', name, '--chains', '128', '--no-cpu-baseline'],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j['config_block'] == name and j['value'] > 0 and j['chains'] == 128
    for k in ('mean_tree_size', 'divergence_rate', 'roofline', 'workload', 'ms_per_launch', 'launch_tail', 'work_share_top_2pct_chains'):
        assert k in j, k
    assert 0. <= j['divergence_rate'] <= 1. and j['roofline']['frac'] > 0
    if name == 'banana_decay':
        assert 'round_1' in j and 4200 < j['refit']['n_fit_points'] <= 4290 and j['both_rounds_value'] > 0


def test_bench_config5_evidence_block():
    """BASELINE config 5's "evidence via GBS" at config 5's dimension (bench.py: evidence_block; a reduced chain count here):
    sample() on the 128-d cubic-cross surrogate of the Planck-like target's Gaussian part, GBS on the device path, and the
    closed-form log Z within the estimate's own error."""
    r = subprocess.run([sys.executable, 'bench.py', '--workload', 'evidence128', '--chains', '256', '--no-cpu-baseline'],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j['config_block'] == 'evidence128' and j['dim'] == 128 and j['chains'] == 256
    for k in ('fit_ms', 'sample_ms', 'gbs_ms', 'log_z', 'log_z_err', 'log_z_exact', 'sit_ms_per_iteration'):
        assert k in j and j[k] == j[k], k
    assert 0. < j['log_z_err'] < 0.05
    assert abs(j['log_z'] - j['log_z_exact']) < 4. * j['log_z_err'] + 0.02, j
