#!/usr/bin/env python3
"""Headline benchmark: leapfrog steps/sec (all chains), 4096 chains x 64-d quadratic surrogate, NUTS -- on SURVEY 8d's
"Config 3 (headline)": the 64-d rotated banana, round 0 -> one refit cycle -> round 1.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE launch of the fused NUTS kernel that advances every chain of the rank by --iters NUTS iterations (reference
loop: BaseHMC.run/astep, samplers/hmc_utils/base_hmc.py:62-85,155-156).  The K timed steps are split over the workload's two
sampling rounds: ceil(K / 2) launches on the first fit's surrogate (round 0), then -- untimed, reported beside it -- the refit
(2 P of ALL ranks' round-0 samples picked by their logq, true logp, least-squares fit), and floor(K / 2) launches on the refitted
surrogate (round 1).  Each round: a fixed NUTS adaptation (step size and diagonal metric, untimed, like the fit), W untimed
launches, then its timed launches bracketed by a barrier + torch.cuda.synchronize() on both sides.  value = leapfrog steps of all
K timed launches (sum of tree_size, samplers/sample_trace.py:529-530; the gradient evaluation that opens an iteration is not a
leapfrog step) / the bracketed time of both rounds, max over ranks.  Chains shard over ranks with no data-path collective while
sampling (weak scaling: 4096 chains per GPU, RNG stream = global chain index); the refit's selection is the path's one exchange.

The CPU baseline is the repository's C restatement of the reference path (oracle/, "port") with its tuned density evaluation, run
on the host cores on a bounded sample of the same two rounds; it is a reported baseline, never the measured path.  The side
blocks (tools/benchlib/blocks.py) are never part of `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

N_ADAPT_ROUND = 200   # NUTS adaptation iterations of each round before anything is timed (GPU path; the CPU baseline starts from the device's adapted state)
PEAK_TF = 78.6        # FP64 MFMA, 256 CUs x 4 SIMDs x 2.4 GHz x 2048 flop / 64 cyc; 77.7 measured (profiles/r01_probe_mfma_f64.log)


def _spawn_ranks(n, backend):
    """One process per GPU over torch.distributed.run, as the driver launches them; exits with the launcher's code."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    if backend == 'nccl' and n_dev < n:
        print('bench.py: --gpus %d needs %d GPUs for RCCL, %d visible (plumbing runs on fewer GPUs: --backend gloo)' % (n, n, n_dev),
              file=sys.stderr)
        sys.exit(3)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def _dominant_roofline(rounds):
    """The roofline object of the line: the kernel of the round that takes most of the timed region (round 1: the refitted
    banana's trees run to the depth limit), its flops EXECUTED per launch over its average launch duration (HIP events on the
    launch stream, measured live over that round's timed launches)."""
    r = max(rounds, key=lambda b: b['wall_s_timed'])
    rf = dict(r['roofline'])
    rf['round'] = rounds.index(r)
    rf['share_of_timed_region'] = r['wall_s_timed'] / sum(b['wall_s_timed'] for b in rounds)
    rf['flops_note'] = ('achieved / frac count the flops the kernel EXECUTED per launch -- S x and H (x - mu), 4 d^2 per leapfrog step: the decay '
                        'term\'s matrix and centre are the bound\'s on this density (both come from the fit points), so its product '
                        'H_d (x - mu_d) is the bound\'s and the pipelined kernel\'s two-matrix form runs it once; round 5 executed 6 d^2 -- over '
                        'kernel_ms_per_launch, the HIP-event time of one launch of the dominant round averaged over its timed launches; '
                        '*_algorithmic count the same 4 d^2 whether executed or proven away')
    return rf


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-child':
        from benchlib import cpu
        return cpu.child_main(sys.argv[2])
    from benchlib import blocks
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--chains', type=int, default=4096, help='chains per GPU')
    ap.add_argument('--iters', type=int, default=100, help='NUTS iterations per step = per launch')
    ap.add_argument('--seed', type=int, default=2024)
    ap.add_argument('--backend', default='nccl', help="process-group backend: 'nccl' (= RCCL; default) or 'gloo' (plumbing tests)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fit', action='store_true', help='skip the separately reported fit timing')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary figures (best-case Gaussian, hetero workload, refit cycle, other samplers)')
    ap.add_argument('--no-configs', action='store_true', help="skip the blocks on the other BASELINE configs' targets (config2/4/5, DES pipeline, evidence)")
    ap.add_argument('--workload', default=None, choices=blocks.CONFIG_BLOCKS + ('evidence128', 'gauss64'),
                    help='run ONE block only and print it (profiling: rocprofv3 -- python3 bench.py --workload funnel)')
    a = ap.parse_args()
    if a.no_extras:
        a.no_configs = True
    if a.gpus < 1:
        ap.error('--gpus should be a positive int')
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, BEFORE anything here touches the GPU
        # (fresh children through torch.distributed.run; this process only relays their output and exit code)
        return _spawn_ranks(a.gpus, a.backend)
    if int(os.environ.get('WORLD_SIZE', '1')) != a.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%s: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node '
              '%d ... bench.py --gpus %d), or run `python bench.py --gpus %d` without a launcher'
              % (a.gpus, os.environ.get('WORLD_SIZE'), a.gpus, a.gpus, a.gpus), file=sys.stderr)
        sys.exit(2)

    import torch
    from bayesfast_amd.device import DeviceContext
    from bayesfast_amd.workloads import B_STEP_BYTES

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # one process per GPU; the modulo only matters for plumbing tests on fewer GPUs
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(a.backend)

    d, C = 64, a.chains
    ctx = DeviceContext(dev_index)
    cpu_s = 0. if a.no_cpu_baseline else 4.
    if a.workload:  # one block on its own (the profiles of profiles/*_config_counters.json come from these commands)
        with torch.cuda.device(ctx.device):
            if a.workload == 'evidence128':
                blk = blocks.evidence_block(ctx, a.seed, chains=1024 if a.chains == 4096 else a.chains)
            elif a.workload == 'gauss64':
                blk = blocks.gauss64_best_case(ctx, a.seed, C, cpu_seconds=0. if a.no_cpu_baseline else 8.)
            else:
                blk = blocks.config_block(a.workload, ctx, a.seed, cpu_seconds=cpu_s, chains=None if a.chains == 4096 else a.chains)
        print(json.dumps({'config_block': a.workload, **blk}))
        return

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    red_dev = ctx.device if (dist is None or a.backend == 'nccl') else torch.device('cpu')

    def reduce(v, op):
        t = torch.tensor([float(v)], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t, op=op)
        return float(t.item())

    k0 = (a.steps + 1) // 2          # timed launches of round 0
    k1 = a.steps - k0                # ... of round 1
    with torch.cuda.device(ctx.device):
        t_run = time.perf_counter()
        c3 = blocks.config3_rounds(ctx, a.seed, C, a.iters, N_ADAPT_ROUND, k0, a.warmup, rank=rank, world=world, sync=sync,
                                   cpu_seconds=(14. if (world == 1 and not a.no_cpu_baseline) else 0.), steps_round1=max(k1, 1))
        rounds = c3['rounds']
        if k1 == 0:   # (--steps 1: the line times round 0 alone; round 1 ran one launch for its own block)
            timed_rounds = rounds[:1]
        else:
            timed_rounds = rounds
        # max over ranks of each round's bracketed time, sum over ranks of its leapfrogs
        SUM, MAX = (dist.ReduceOp.SUM, dist.ReduceOp.MAX) if dist is not None else (None, None)
        lf_tot = [reduce(r['leapfrogs_timed'], SUM) for r in timed_rounds]
        t_max = [reduce(r['wall_s_timed'], MAX) for r in timed_rounds]
        elapsed_max, total_lf = sum(t_max), sum(lf_tot)
        chk = reduce(c3['checksum_of_selected_rows'], MAX) if dist is not None else c3['checksum_of_selected_rows']
        chk_min = reduce(-c3['checksum_of_selected_rows'], MAX) if dist is not None else -c3['checksum_of_selected_rows']

        # the layout vote of a sharded run (DeviceChains._note_trees: the histogram of a launch's tree sizes summed over the ranks, so
        # that every rank chooses the same layout for the next launch) is INSIDE the timed launches above; its own cost, for the record
        vote = None
        if dist is not None:
            from bayesfast_amd import parallel, _lib
            from bayesfast_amd.chains import DeviceChains
            st = ctx.zeros((C, a.iters, _lib.STAT_STRIDE))
            st[:, :, _lib.NSTATS.index('tree_size')] = 7.
            from bayesfast_amd.device import DeviceDensity
            from bayesfast_amd.workloads import correlated_gaussian_spec
            vc = DeviceChains(DeviceDensity(correlated_gaussian_spec(d)[0], ctx), np.zeros((C, d)), seed=1)
            vc.hist_reduce = parallel.all_reduce_sum
            tv = []
            for _ in range(5):
                sync()
                t1 = time.perf_counter()
                vc._note_trees(st, 0, a.iters, 'NUTS')
                tv.append((time.perf_counter() - t1) * 1e3)
            tvx = reduce(min(tv[1:]), MAX)
            vote = {'ms_per_launch': tvx, 'share_of_a_launch': tvx / (elapsed_max / max(a.steps, 1) * 1e3),
                    'includes': 'histogram of the last 32 iterations\' tree sizes (device), stream synchronisation, all-reduce of %d int64 '
                                '(%s), max over ranks, best of 4; the timed launches of `value` pay it' % (4096 + 64, a.backend)}

    if rank == 0:
        value = total_lf / elapsed_max
        rf = _dominant_roofline(timed_rounds)
        dom = timed_rounds[rf['round']]
        from bayesfast_amd import _lib
        out = {
            'metric': 'leapfrog steps/sec (all chains), %d chains x %d-d quadratic surrogate' % (C, d),
            'value': value, 'unit': 'leapfrog steps/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed_max / max(a.steps, 1) * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'SURVEY 8d config 3 (headline): %d chains/GPU x 64-d rotated banana (Q = 0.01), PolyModel(\'quadratic\') '
                                   'surrogate (P = 2145) fitted on 2 P Sobol-normal points, bound and decay on, NUTS defaults (diag-adapt metric, '
                                   'target_accept 0.8, max_treedepth 10); round 0 (%d timed launches) -> one refit cycle (2 P of all ranks\' '
                                   'round-0 samples by their logq, true logp, refit: untimed, reported in `refit`) -> round 1 (%d timed launches); '
                                   'value = leapfrogs of both rounds / sampling time of both rounds.  Departure from SURVEY 8d: the decay term '
                                   '(core/density.py:740-746) is on, as in the reference\'s GBS recipes -- without it the chains run away along the '
                                   'first fit\'s indefinite quadratic form.  The benign best case of the path (64-d Gaussian, exact surrogate, every '
                                   'tree 7 leaves) is the gauss64_best_case block' % (C, len(timed_rounds) and k0, k1),
                       'chains_per_gpu': C, 'dim': d, 'nuts_iterations_per_step': a.iters,
                       'nuts_adaptation_iterations_per_round': N_ADAPT_ROUND,
                       'timed_launches_round0': k0, 'timed_launches_round1': k1,
                       'mean_tree_size_round0': rounds[0]['mean_tree_size'], 'mean_tree_size_round1': rounds[-1]['mean_tree_size'],
                       'parallelism': 'chains sharded over %d rank(s), no data-path collective while sampling; the refit selection is '
                                      'the one exchange (4 collectives)' % world},
            'roofline': rf,
            'roofline_hbm_algorithmic': {'bound': 'hbm', 'achieved': total_lf / world * B_STEP_BYTES(d) / elapsed_max / 1e9, 'peak': 8000.,
                                         'unit': 'GB/s', 'frac': total_lf / world * B_STEP_BYTES(d) / elapsed_max / 1e9 / 8000.,
                                         'bytes_per_leapfrog': B_STEP_BYTES(d),
                                         'note': 'SURVEY 8d\'s HBM-side figure per GPU: leapfrogs x (48 d + 32) B over the timed region; the state stays on '
                                                 'chip inside a launch, so the FP64 pipe is the binding roofline'},
            'config3_round0_value': lf_tot[0] / t_max[0],
            'config3_round1_value': (lf_tot[1] / t_max[1]) if len(lf_tot) > 1 else None,
            'config3_round0': {k: v for k, v in rounds[0].items() if not k.startswith('_')},
            'config3_round1': {k: v for k, v in rounds[-1].items() if not k.startswith('_')},
            'refit': dict(c3['refit'], identical_on_all_ranks=bool(chk == -chk_min),
                          note='fit_0 / fit_1: PolyModel.fit on the device (design blocks, MFMA Gram, blocked Cholesky, bound and decay statistics); '
                               'select: device sort of every rank\'s %d logq values + 4 collectives; true_logp: the banana on the host (not part of the path)'
                               % (C * a.iters)),
            'wall_s_whole_workload': time.perf_counter() - t_run,
        }
        if dist is not None:
            out['distributed'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                                  'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()) if a.backend == 'nccl' else None,
                                  'devices_visible': n_dev}
            out['refit_exchange'] = {'ms': c3['refit']['select_ms'], 'wire_bytes_per_rank': c3['refit']['wire_bytes_per_rank'],
                                     'collectives': c3['refit']['collectives'], 'rows_selected': 2 * 2145, 'rows_per_rank': C * a.iters,
                                     'identical_on_all_ranks': bool(chk == -chk_min),
                                     'includes': 'local device sort of %d keys + 4 collectives (%s), rank 0\'s time of the one selection the '
                                                 'workload makes' % (C * a.iters, a.backend)}
        if vote is not None:
            out['layout_vote'] = vote
        if 'cpu_baseline' in c3:
            out['cpu_baseline'] = c3['cpu_baseline']
        elif not a.no_cpu_baseline:
            out['cpu_baseline'] = None
        try:   # the AS-SHIPPED reference's rate: a stored measurement of the build container (tools/time_reference.py), never timed here
            rt = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_timing.json')))
            out['reference_as_shipped'] = {'leapfrog_steps_per_sec_per_core': rt['leapfrog_steps_per_sec_per_core'],
                                           'polymodel_fit_s': rt.get('polymodel_fit_s'), 'host': rt['host']['cpu'],
                                           'source': 'tests/golden/reference_timing.json (build container; the Python reference does not travel)'}
        except Exception:
            pass
        if world == 1:
            from bayesfast_amd.workloads import correlated_gaussian_spec
            _, cov = correlated_gaussian_spec(d)
            with torch.cuda.device(ctx.device):
                if not a.no_fit and not a.no_cpu_baseline:
                    try:
                        out['fit'] = blocks.fit_timing(d, cov)
                    except Exception as ex:  # side measurements; the headline line must still print
                        out['fit'] = {'error': repr(ex)}
                if not a.no_extras:
                    try:
                        out['gauss64_best_case'] = blocks.gauss64_best_case(ctx, a.seed, C, cpu_seconds=0. if a.no_cpu_baseline else 6.)
                        out['hetero'] = blocks.hetero_rate(ctx, d, C, a.seed, 250)
                        out['scaled_inputs'] = blocks.scaled_inputs_rate(ctx, d, C, a.seed, 250)
                        out['refit_cycle'] = blocks.refit_cycle(d, cov, C, a.seed)
                        out.update(blocks.other_samplers(ctx, d, cov, C, a.seed))
                    except Exception as ex:
                        out['extras_error'] = repr(ex)
                if not a.no_configs:
                    for name, key in zip(blocks.CONFIG_BLOCKS, blocks.CONFIG_KEYS):
                        if name == 'banana_decay':
                            continue   # (config 3 IS the line)
                        try:
                            out[key] = blocks.config_block(name, ctx, a.seed, cpu_seconds=cpu_s)
                        except Exception as ex:
                            out[key] = {'error': repr(ex)}
                    try:   # config 5's "evidence via GBS" at config 5's size
                        out['config5_evidence'] = blocks.evidence_block(ctx, a.seed)
                    except Exception as ex:
                        out['config5_evidence'] = {'error': repr(ex)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
