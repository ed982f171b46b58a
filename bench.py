#!/usr/bin/env python3
"""Headline benchmark: leapfrog steps/sec (all chains), 4096 chains x 64-d quadratic surrogate, NUTS.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is ONE launch of the fused NUTS kernel that advances every chain of the rank by --iters NUTS
iterations (reference loop: BaseHMC.run/astep, samplers/hmc_utils/base_hmc.py:62-85,155-156).  The W
untimed steps are the NUTS warm-up (step-size and metric adaptation, n_warmup = W * iters); the K timed steps
are post-warm-up sampling.  Leapfrog steps are counted as the reference counts them: sum of tree_size
(samplers/sample_trace.py:529-530); the extra gradient evaluation that opens every iteration is not a
leapfrog step.  Chains shard over ranks with no data-path collective (weak scaling: 4096 chains per GPU,
RNG stream = global chain index).

The CPU baseline is the repository's C restatement of the reference path (oracle/, "port"), run on the
host cores on a bounded sample of the same workload; it is a reported baseline, never the measured path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


N_ADAPT = 750  # NUTS adaptation iterations before anything is timed (GPU path and CPU baseline alike)


def _host():
    from bayesfast_amd.utils.hostinfo import host_cpu_facts
    h = host_cpu_facts()
    h.pop('one_cpu_per_core', None)
    return h


def _timed_slices(cs, n_warm_iter, n_thr, target_seconds, first_slice=50):
    """Leapfrogs and seconds of post-warm-up slices of the chain set, each slice long enough (>= ~1 s) for the parallel
    region's start-up and the output arrays' page faults not to count."""
    nl, dt, n_it, it = 0, 0., 0, first_slice
    while dt < target_seconds:
        t0 = time.perf_counter()
        _, _, k = cs.run(it, n_warm_iter, n_threads=n_thr)
        t1 = time.perf_counter() - t0
        dt += t1
        nl += k
        n_it += it
        if t1 < 1.:
            it = min(2 * it, 4000)
    return nl, dt, n_it


def _cpu_baseline_here(spec, d, n_warm_iter, seed, target_seconds=12.):
    """Leapfrog steps/sec of the CPU port on the host cores THIS PROCESS MAY USE, post-warm-up, on a bounded sample.  The NUTS
    driver is the oracle's; the density evaluation is its tuned form (oracle/bf_cpu_tuned.c: one symmetrised dense matvec pair
    per gradient, AVX2 + FMA, no allocation -- ~18x the statement-by-statement checker), i.e. the stronger baseline.

    /proc/cpuinfo lists every CPU of the machine; the affinity mask and the cgroup's cpu.max say what the process gets (the
    round-3 line ran 128 threads inside a 16-CPU quota: 34 k steps/s/thread instead of 850 k).  `threads` = min(CPUs in the
    affinity mask, cgroup quota), one chain per thread, threads bound to cores (OMP_PROC_BIND=close, OMP_PLACES=cores); the
    1-thread rate of the same code is measured beside it so that the per-thread efficiency is on the line."""
    from oracle import oracle as orc  # the checker, timed as a baseline only
    host = _host()
    n_thr = max(1, min(host['usable_threads'], orc.max_threads()))
    # one thread, four chains
    x1 = np.random.default_rng(seed).normal(size=(4, d))
    c1 = orc.ChainSet(spec, x1, seed, tuned=True)
    c1.run(n_warm_iter, n_warm_iter, n_threads=1)
    nl1, dt1, _ = _timed_slices(c1, n_warm_iter, 1, min(3., target_seconds / 4))
    c1.close()
    n_chain = 4 * n_thr
    x0 = np.random.default_rng(seed).normal(size=(n_chain, d))
    cs = orc.ChainSet(spec, x0, seed, tuned=True)
    cs.run(n_warm_iter, n_warm_iter, n_threads=n_thr)  # untimed adaptation, same as the GPU path
    nl, dt, n_it = _timed_slices(cs, n_warm_iter, n_thr, target_seconds)
    tuned = cs.tuned
    cs.close()
    one = nl1 / dt1
    return {'value': nl / dt, 'unit': 'leapfrog steps/sec', 'cores': host['usable_cores'], 'threads': n_thr, 'kind': 'port',
            'one_thread_value': one, 'per_thread_efficiency': (nl / dt) / (n_thr * one), 'host': host,
            'omp': {k: os.environ.get(k) for k in ('OMP_PROC_BIND', 'OMP_PLACES', 'OMP_NUM_THREADS')},
            'sample': '%d chains x %d post-warm-up NUTS iterations (%d leapfrogs in %.1f s) of the same %d-d workload, '
                      'one chain per OpenMP thread on the %d CPUs the process may use (affinity %d, cgroup quota %s), %s density '
                      'evaluation, %s' % (n_chain, n_it, nl, dt, d, n_thr, host['affinity_cpus'], host['cgroup_cpu_quota'],
                                          'tuned (dense symmetric matvec, AVX2+FMA)' if tuned else 'statement-by-statement',
                                          host['model'])}


def hetero_rate(ctx, d, C, seed, iters, steps=3, layout='auto'):
    """Secondary figure: the same surrogate family on a target whose trees differ from chain to chain and from iteration
    to iteration (per-dimension scales spread over a decade, identity metric kept fixed: tree sizes 7 .. 63 side by side
    in one workgroup), so that the 16 chains of a group do NOT run in step.  Post-adaptation launches, HIP events."""
    import torch
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    spec, _ = correlated_gaussian_spec(d, scales=np.logspace(-0.5, 0.5, d))
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(seed + 1).normal(size=(C, d))
    ch = DeviceChains(dens, x0, seed=seed + 1)
    kw = dict(n_warmup=N_ADAPT, check=False, adapt_metric=False, target_accept=0.9, layout=layout)
    ch.run(N_ADAPT, 'NUTS', **kw)
    s = ctx.empty((C, iters, d))
    st = ctx.empty((C, iters, _lib.STAT_STRIDE))
    ch.run(iters, 'NUTS', samples=s, stats=st, **kw)
    ch.raise_on_error()
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(ctx.stream)
    for _ in range(steps):
        ch.run(iters, 'NUTS', samples=s, stats=st, **kw)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    ch.raise_on_error()
    ts = st[:, :, _lib.NSTATS.index('tree_size')].cpu().numpy()
    sizes, counts = np.unique(ts, return_counts=True)
    return {'value': (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), 'unit': 'leapfrog steps/sec',
            'mean_tree_size': float(ts.mean()), 'chain_layout': layout,
            'tree_size_share': {str(int(k)): round(float(v) / ts.size, 4) for k, v in zip(sizes, counts)},
            'workload': '%d chains x %d-d Gaussian with per-dimension scales 10^-0.5 .. 10^0.5, identity metric '
                        '(adapt_metric off), target_accept 0.9, %d x %d post-adaptation iterations' % (C, d, steps, iters)}


def scaled_inputs_rate(ctx, d, C, seed, iters, steps=3):
    """Secondary figure: the headline surrogate WITH Surrogate.input_scales (module.py:190-226), as every surrogate of the reference's
    recipes has them: x = lo + diff x_s, the polynomial in x_s.  The scaling is folded into the coefficients and the bound at upload
    (device.density_desc_from_spec), so the launch runs on the same kernels as the headline; the bound's ellipsoid is no longer
    aligned with the proof's sphere, so more trips run its tiles.  Post-adaptation launches, HIP events."""
    import torch
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    rng = np.random.default_rng(seed + 5)
    lo, diff = rng.normal(size=d), rng.uniform(0.5, 3., size=d)
    spec, _ = correlated_gaussian_spec(d)
    spec = dict(spec, su_lo=lo, su_diff=diff)
    ch = DeviceChains(DeviceDensity(spec, ctx), lo + diff * rng.normal(size=(C, d)), seed=seed + 5)
    kw = dict(n_warmup=N_ADAPT, check=False)
    ch.run(N_ADAPT, 'NUTS', **kw)
    ch.run(iters, 'NUTS', **kw)
    ch.raise_on_error()
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(ctx.stream)
    for _ in range(steps):
        s, st = ch.run(iters, 'NUTS', **kw)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    ch.raise_on_error()
    kname = _lib.last_kernel
    return {'value': (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), 'unit': 'leapfrog steps/sec', 'chains': C, 'dim': d,
            'mean_tree_size': float(st[:, :, _lib.NSTATS.index('tree_size')].mean().item()), 'kernel': kname(),
            'note': 'the headline surrogate behind input scales (lo + diff x_s), folded into its coefficients at upload'}


def other_samplers(ctx, d, cov, C, seed):
    """Secondary figures: the two samplers of the path that are not the default -- NUTS with the full-rank metric
    (QuadMetricFull, metrics.py:94-132; every chain streams its own d x d covariance twice per leapfrog step) after adaptation,
    and tempered NUTS (samplers/tnuts.py, integration.py:98-222) with a Gaussian base density.  HIP events, one launch each."""
    import torch
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(seed + 2).normal(size=(C, d))
    out = {}

    def timed(f):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(ctx.stream)
        r = f()
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        return r, e0.elapsed_time(e1) * 1e-3

    ch = DeviceChains(dens, x0, seed=seed + 2, metric='full')
    ch.run(300, 'NUTS', n_warmup=300, check=False)
    lf0 = ch.total_leapfrog
    (_, st), t = timed(lambda: ch.run(100, 'NUTS', n_warmup=300, check=False))
    ch.raise_on_error()
    n_lf = ch.total_leapfrog - lf0
    out['full_metric'] = {'value': n_lf / t, 'unit': 'leapfrog steps/sec', 'chains': C, 'dim': d,
                          'mean_tree_size': float(st[:, :, _lib.NSTATS.index('tree_size')].mean().item()),
                          'covariance_traffic_GBps': n_lf * 2 * 8 * d * d / t / 1e9,
                          'note': 'per-chain adapted covariances (fixed in the timed launch); two cov p products per leapfrog step'}
    ch = DeviceChains(dens, x0, seed=seed + 3)
    ch.run_tempered(120, np.zeros(d), 1.3 * cov, n_warmup=100, check=False)
    lf0 = ch.total_leapfrog
    (_, st, _), t = timed(lambda: ch.run_tempered(60, np.zeros(d), 1.3 * cov, n_warmup=100, check=False))
    ch.raise_on_error()
    ts_t = st[:, :, _lib.NSTATS.index('tree_size')].sum(1)
    out['tempered'] = {'value': (ch.total_leapfrog - lf0) / t, 'unit': 'tempered leapfrog steps/sec', 'chains': C, 'dim': d,
                       'mean_tree_size': float(st[:, :, _lib.NSTATS.index('tree_size')].mean().item()),
                       'launch_tail': float((ts_t.max() / ts_t.mean()).item()),
                       'note': 'TNUTS, Gaussian base density 1.3 x the target covariance; each step evaluates both densities twice'}
    return out


def refit_cycle(d, cov, C, seed):
    """One refit cycle end to end through the package API (BASELINE config 3's shape: sample -> choose 2P points by
    logq -> true logp -> fit -> sample), wall-clock per stage.  The true model is the exactly quadratic target evaluated
    on the host; the banana of config 3 is a parity case (its quadratic surrogate is indefinite)."""
    import torch
    import bayesfast_amd as bfa
    from bayesfast_amd.core.refit import select_fit_points
    prec = np.linalg.inv(cov)

    def logp_true(x):   # (the host's "true model": one matrix product; a three-operand einsum spends 10 ms on 4290 points)
        from bayesfast_amd.utils.threads import blas_single_thread
        with blas_single_thread():
            return -0.5 * np.sum((x @ prec) * x, axis=1)

    # the extrapolation bound at 150 % of the largest Mahalanobis radius of the fit points (PolyModel bound_options,
    # modules/poly.py:232-260): refitted on points drawn FROM the posterior, an ellipsoid through the outermost fit point
    # (alpha_p = 100) cuts into the posterior's own tail in 64 dimensions, and the linear extrapolation outside lets
    # chains leak out
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
    dens = bfa.SurrogateDensity(su)
    n_eval = 2 * su.n_param
    x = 1.5 * np.random.default_rng(seed).normal(size=(n_eval, d))
    t = {}

    def timed(name, f):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        t[name] = (time.perf_counter() - t0) * 1e3
        return r

    timed('fit_0_ms', lambda: dens.fit(x, logp_true(x)))
    kw = dict(n_chain=C, n_iter=1500, n_warmup=500, random_generator=seed)
    tw = bfa.sample(dens, dict(kw), verbose=False)  # untimed, full size: first use of the kernels, and the allocator's blocks
    select_fit_points(tw, None, logp_true, n_eval, logp_cutoff=False)   # (and of the selection path: sort workspace)
    del tw
    t0 = time.perf_counter()
    tt = timed('sample_0_ms', lambda: bfa.sample(dens, dict(kw), verbose=False))
    xf, lf, n_true = timed('select_and_true_logp_ms', lambda: select_fit_points(tt, None, logp_true, n_eval, logp_cutoff=False))
    timed('fit_1_ms', lambda: dens.fit(xf, lf))
    del tt  # (a recipe drops the previous round's trace here; its 3 GB go back to the allocator's cache, not to the driver)
    tt2 = timed('sample_1_ms', lambda: bfa.sample(dens, dict(kw), verbose=False))
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    var_ratio = float(np.mean(tt2.device('samples')[:, 500:].reshape(-1, d).var(0).cpu().numpy() / np.diag(cov)))
    return dict(t, total_ms=total, n_fit_points=int(xf.shape[0]), n_param=int(su.n_param),
                chains=C, iterations_per_round=1500, posterior_variance_ratio_after_refit=var_ratio,
                note='sample_0 -> select (device sort of %d logq values, %d rows to the host) -> true logp on the host -> '
                     'fit_1 -> sample_1; total excludes fit_0' % (C * 1000, int(xf.shape[0])))


def fit_timing(d, cov, seed=7):
    """Device least-squares fit of the same surrogate family (PolyModel.fit, modules/poly.py:505-589): n = 2 P
    points of the exactly quadratic target, timed on the second call (reported beside the headline, never in it)."""
    import torch
    from bayesfast_amd import PolyModel
    su = PolyModel('quadratic', input_size=d, output_size=1)
    n_param = su.n_param
    x = np.random.default_rng(seed).normal(size=(2 * n_param, d))
    prec = np.linalg.inv(cov)
    y = -0.5 * np.einsum('ij,jk,ik->i', x, prec, x)
    dts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        su.fit(x, y[:, None], logp=y)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = min(dts[1:])
    err = float(max(abs(float(np.ravel(su.fun(x[i])[0])[0]) - y[i]) for i in range(8)))
    return {'ms': dt * 1e3, 'n': int(x.shape[0]), 'n_param': int(n_param), 'gram_flops': 2. * x.shape[0] * n_param**2,
            'max_abs_residual_on_fit_points': err,
            'note': 'host arrays in, coefficients out: upload, design blocks, split-K MFMA Gram, blocked Cholesky solve, bound statistics'}


def _cpu_rate_here(spec, x_start, step_size, var, seed, target_accept, target_seconds):
    """The CPU port on the same density: the oracle's NUTS driver, one chain per OpenMP thread, bounded.  The chains start
    where the device chains are after their adaptation, with the device's adapted step size and diagonal metric (means over
    the chains) held fixed -- the CPU pays for sampling, not for a second adaptation (minutes on the deep-tree configs)."""
    from oracle import oracle as orc
    host = _host()
    n_thr = max(1, min(host['usable_threads'], orc.max_threads()))   # (what the process may use: cpu_baseline)
    n_chain = min(n_thr, x_start.shape[0])
    cs = orc.ChainSet(spec, x_start[:n_chain], seed, step_size=step_size, metric=var, adapt_step_size=False, adapt_metric=False,
                      target_accept=target_accept)
    nl, dt, n_it, slice_it = 0, 0., 0, 2
    while dt < target_seconds:
        t0 = time.perf_counter()
        _, _, k = cs.run(slice_it, 0, n_threads=n_thr)
        t1 = time.perf_counter() - t0
        if n_it:   # (the first slice pays the page faults of the threads' stacks: untimed)
            dt += t1
            nl += k
        n_it += slice_it
        if t1 < 0.3:
            slice_it = min(4 * slice_it, 200)
    return {'value': nl / dt, 'unit': 'leapfrog steps/sec', 'cores': host['usable_cores'], 'threads': n_thr, 'kind': 'port',
            'omp': {k: os.environ.get(k) for k in ('OMP_PROC_BIND', 'OMP_PLACES', 'OMP_NUM_THREADS')},
            'sample': '%d chains x %d NUTS iterations (%d leapfrogs in %.1f s) of the same density from the device chains\' '
                      'post-adaptation positions, with their adapted step size and diagonal metric (chain means) held fixed; one '
                      'chain per OpenMP thread' % (n_chain, n_it, nl, dt)}


def _cpu_child(job):
    """The CPU baseline runs in a CHILD process that never touches the GPU: libgomp reads its thread placement when it loads,
    so OMP_PROC_BIND=close / OMP_PLACES=cores / OMP_NUM_THREADS are set for the child only and the benchmark's own process
    (HIP runtime threads, launch path) keeps the scheduler's placement.  The job travels as a pickle, the answer as one JSON
    line."""
    import pickle
    import subprocess
    import tempfile
    host = _host()
    env = dict(os.environ)
    env.setdefault('OMP_PROC_BIND', 'close')
    env.setdefault('OMP_PLACES', 'cores')
    env.setdefault('OMP_NUM_THREADS', str(host['usable_threads']))
    with tempfile.NamedTemporaryFile(suffix='.pkl', delete=False) as f:
        pickle.dump(job, f)
        path = f.name
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-child', path], env=env, capture_output=True, text=True)
    finally:
        os.unlink(path)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not lines:
        return {'error': 'cpu baseline child failed (rc %d): %s' % (r.returncode, r.stderr[-400:])}
    return json.loads(lines[-1])


def cpu_baseline(spec, d, n_warm_iter, seed, target_seconds=12.):
    return _cpu_child(dict(kind='headline', spec=spec, d=d, n_warm_iter=n_warm_iter, seed=seed, target_seconds=target_seconds))


def _cpu_rate(spec, x_start, step_size, var, seed, target_accept, target_seconds):
    return _cpu_child(dict(kind='config', spec=spec, x_start=x_start, step_size=step_size, var=var, seed=seed,
                           target_accept=target_accept, target_seconds=target_seconds))


def _cpu_child_main(path):
    import pickle
    with open(path, 'rb') as f:
        job = pickle.load(f)
    if job.pop('kind') == 'headline':
        out = _cpu_baseline_here(**job)
    else:
        out = _cpu_rate_here(**job)
    print(json.dumps(out))


def _sampler_block(ctx, den, x0, seed, target_accept, n_adapt, iters, steps, cpu_seconds, what, first_stream=0):
    """Adapt n_adapt NUTS iterations on the device, then time `steps` launches of `iters` iterations (HIP events on the
    launch stream).  Returns (block dict, samples (C, iters, d), stats) of the last launch."""
    import torch
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import B_STEP_BYTES, flops_per_leapfrog_spec
    from bayesfast_amd import _lib
    dd = den.device(ctx)
    C, d = x0.shape
    ch = DeviceChains(dd, x0, seed=seed, first_stream=first_stream)
    kw = dict(n_warmup=n_adapt, check=False, target_accept=target_accept)
    ch.run(n_adapt, 'NUTS', **kw)
    s = ctx.empty((C, iters, d))
    st = ctx.empty((C, iters, _lib.STAT_STRIDE))
    ch.run(iters, 'NUTS', samples=s, stats=st, **kw)   # one untimed post-adaptation launch
    ch.raise_on_error()
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # measurement hook of the lane-per-chain kernels: trips, trips with the bound's tiles, with a late exchange, without the early one
    gcount = torch.zeros(4, dtype=torch.int64, device=ctx.device)
    _lib.debug_buffer('group_counters', gcount)
    torch.cuda.synchronize()
    e0.record(ctx.stream)
    for _ in range(steps):
        ch.run(iters, 'NUTS', samples=s, stats=st, **kw)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    _lib.debug_buffer('group_counters', None)
    ch.raise_on_error()
    ms = e0.elapsed_time(e1)
    n_lf = ch.total_leapfrog - lf0
    kname = _lib.last_kernel
    stn = st.cpu().numpy()
    ts = stn[:, :, _lib.NSTATS.index('tree_size')]
    spec = den.spec()
    fl = flops_per_leapfrog_spec(spec)
    ach = n_lf * fl / (ms * 1e-3) / 1e12
    exec_share = 1.
    if spec.get('chi2') is not None:
        # the pipeline density compresses m outputs to min(m, n_monomials) rows at upload (exact: bfhip_pipeline_upload), so the
        # contractions EXECUTE 4 min(m, nf) nf flops where the reference's algorithm has 4 m nf
        m_out = int(spec['poly']['output_size'])
        nf = (fl - (2 * d * d if spec['poly'].get('use_bound') else 0) - (2 * d * d if spec.get('use_decay') else 0)) // (4 * m_out)
        exec_share = (fl - 4 * m_out * nf + 4 * min(m_out, nf) * nf) / fl
    out = {'workload': what, 'value': n_lf / (ms * 1e-3), 'unit': 'leapfrog steps/sec', 'chains': int(C), 'dim': int(d),
           'nuts_iterations_timed': steps * iters, 'nuts_adaptation_iterations': n_adapt, 'ms_per_launch': ms / steps,
           'target_accept': target_accept, 'mean_tree_size': float(ts.mean()), 'max_tree_depth': int(stn[:, :, _lib.NSTATS.index('tree_depth')].max()),
           'divergence_rate': float(stn[:, :, _lib.NSTATS.index('diverging')].mean()),
           # a launch lasts as long as its busiest chain: leapfrogs of the busiest chain / of the average chain in the last launch
           'launch_tail': float(ts.sum(1).max() / max(ts.sum(1).mean(), 1.)),
           # ... and how the work is spread over the chains: the share of all leapfrogs taken by the busiest 2 % of the chains
           'work_share_top_2pct_chains': float(np.sort(ts.sum(1))[-max(1, int(0.02 * C)):].sum() / max(ts.sum(), 1.)),
           'mean_accept': float(stn[:, :, _lib.NSTATS.index('mean_tree_accept')].mean()),
           'chain_layout': _layout_of(kname(), ch.last_layout),
           'roofline': {'bound': 'mfma', 'achieved': ach * exec_share, 'peak': 78.6, 'unit': 'TFLOP/s', 'frac': ach * exec_share / 78.6,
                        'traffic': None, 'achieved_algorithmic': ach, 'frac_algorithmic': ach / 78.6,
                        'executed_share_of_algorithmic_flops': exec_share,
                        'kernel': kname(), 'kernel_ms_per_launch': ms / steps, 'flops_per_leapfrog': fl},
           'roofline_hbm_algorithmic': {'bound': 'hbm', 'achieved': n_lf * B_STEP_BYTES(d) / (ms * 1e-3) / 1e9, 'peak': 8000.,
                                        'unit': 'GB/s', 'frac': n_lf * B_STEP_BYTES(d) / (ms * 1e-3) / 1e9 / 8000.}}
    gc = [int(v) for v in gcount.cpu().numpy()]
    if gc[0]:   # (the group / split kernels only)
        out['group_trips'] = {'trips': gc[0], 'with_bound_tiles': gc[1], 'with_late_exchange': gc[2], 'without_early_exchange': gc[3]}
    try:   # HBM-side bytes per launch: a STORED profile value (tools/profile_configs.sh), used when kernel and shape match
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'config_traffic.json')))
        for e in tj.values():
            for blk in e.values():
                if (blk.get('kernel_named_by_library') == out['roofline']['kernel'] and blk.get('chains') == int(C) and blk.get('dim') == int(d)
                        and abs(blk.get('mean_tree_size', -1.) / out['mean_tree_size'] - 1.) < 0.25 and 'hbm_bytes_per_leapfrog' in blk):
                    out['roofline']['traffic'] = blk['hbm_bytes_per_leapfrog'] * n_lf / steps
                    out['roofline']['traffic_source'] = 'stored profile (profiles/config_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), not this run'
    except Exception:
        pass
    if cpu_seconds > 0:
        try:
            step = float((ch.field('log_bar').exp() * d**0.25).mean())   # what _get_step_size hands to the next round
            out['cpu_baseline'] = _cpu_rate(spec, ch.field('q').cpu().numpy(), step, ch.field('var').mean(0).cpu().numpy(), seed,
                                            target_accept, cpu_seconds)
        except Exception as ex:
            out['cpu_baseline'] = {'error': repr(ex)}
    return out, s, st


def _layout_of(kernel, requested):
    """The layout that RAN (a requested 'split' runs the group kernel where bf_split_kernel has no instantiation)."""
    for key, lay in (('bf_split_kernel', 'split'), ('bf_group_kernel', 'group'), ('bf_nuts_pipe_kernel', 'wave'), ('bf_sampler_kernel', 'wave')):
        if kernel.startswith(key):
            return lay
    return requested


def config_block(name, ctx, seed, cpu_seconds=4., chains=None, iters=None, steps=2, n_adapt=None):
    """The BASELINE configs' own targets, one GPU's shard each, through the package API: fit the surrogate
    on 2 P points of the true model, adapt, time post-adaptation launches.
      banana_decay : config 3 -- 64-d rotated banana, quadratic surrogate (P = 2145) WITH the decay term the reference's
                     recipe uses for such targets (core/density.py:740-746), 4096 chains, and ONE refit cycle (2 P of the
                     first round's samples by their logq, true logp, refit, sample again); both rounds reported
      funnel       : config 4's shard -- 64-d funnel, target_accept 0.95, 4096 chains, decay on
      cubic128     : config 5's shard -- d = 128, linear + quadratic + cubic-2 + cubic-3 on 16 inputs (P = 9201), 1024 chains"""
    import torch
    import bayesfast_amd as bfa
    from bayesfast_amd.workloads import banana_logp, funnel_logp, planck_like_logp
    from bayesfast_amd.core.refit import select_fit_points
    rng = np.random.default_rng(seed)
    t_fit = {}
    # launch lengths that keep a block within seconds: the banana's refitted surrogate and config 5 run every tree to the
    # depth limit (1023 leapfrogs per iteration)
    iters = iters or {'gauss32': 250, 'banana_decay': 100, 'funnel': 100, 'cubic128': 20, 'des_pipeline': 100}[name]
    n_adapt = n_adapt or {'gauss32': 500, 'banana_decay': 200, 'funnel': 300, 'cubic128': 150, 'des_pipeline': 300}[name]

    def fit(den, x, lp, key):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        den.fit(x, lp)
        torch.cuda.synchronize()
        t_fit[key] = (time.perf_counter() - t0) * 1e3

    if name == 'des_pipeline':
        # SURVEY 8f-1, the reference's real use (examples/des-y1-w-cosmosis.ipynb): a 457-output surrogate (linear in all 27
        # parameters, quadratic in 9), a whitened chi-square and a Gaussian prior, behind the box transform with hard bounds;
        # NUTS runs on it inside the fused kernel (bfhip_pld.h: two FP64-MFMA contractions per gradient)
        from bayesfast_amd.workloads import des_like_pipeline
        w = des_like_pipeline()
        d, m, C = w['d'], w['m'], chains or 4096
        su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic', input_mask=w['nonlinear'])], input_size=d,
                           output_size=m, input_scales=w['para_range'])
        den = bfa.Chi2PipelineDensity(su, w['data'], prec_diag=np.ones(m), logp0=w['norm'], prior_mu=w['prior_mu'],
                                      prior_prec=w['prior_prec'], prior_c0=w['prior_c0'], input_scales=w['para_range'], hard_bounds=True)
        lo, hi = w['para_range'][:, 0], w['para_range'][:, 1]
        u_true = (w['x_true'] - lo) / (hi - lo)
        n_fit = 4 * su.n_param
        x_fit = lo + (hi - lo) * np.clip(u_true + 0.08 * rng.normal(size=(n_fit, d)), 0.02, 0.98)
        y_fit = w['model'](x_fit)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        den.fit(x_fit, w['logp'](x_fit), y=y_fit)
        torch.cuda.synchronize()
        t_fit['fit_ms'] = (time.perf_counter() - t0) * 1e3
        x0 = den.from_original(lo + (hi - lo) * np.clip(u_true + 0.02 * rng.normal(size=(C, d)), 0.02, 0.98))
        what = ('SURVEY 8f-1 / examples/des-y1-w-cosmosis.ipynb shape: %d chains x %d parameters (box transform, hard bounds), surrogate '
                'of %d outputs = linear + quadratic on %d inputs (%d coefficients per output, fitted on %d points), whitened chi-square '
                '+ Gaussian prior on 13 parameters, bound on' % (C, d, m, len(w['nonlinear']), su.n_param, n_fit))
        r, s_, _ = _sampler_block(ctx, den, x0, seed, 0.8, n_adapt, iters, steps, cpu_seconds, what)
        so = den.to_original_device(s_).reshape(-1, d)
        r['posterior_mean_offset_in_prior_sigma'] = float(np.max(np.abs((so.mean(0).cpu().numpy() - w['x_true']) / (0.05 * (hi - lo)))))
        sp = den.spec()
        pl = sp['poly']
        r['pipeline'] = {'outputs': m, 'monomials': 1 + d + len(w['nonlinear']) * (len(w['nonlinear']) + 1) // 2,
                         'coefficient_matrix_bytes': 8 * m * (1 + d + len(w['nonlinear']) * (len(w['nonlinear']) + 1) // 2),
                         'use_bound': bool(pl.get('use_bound'))}
        return dict(r, **t_fit)
    if name == 'gauss32':
        from bayesfast_amd.workloads import correlated_gaussian_spec, sobol_normal
        d, C = 32, chains or 1024
        _, cov = correlated_gaussian_spec(d)       # (its precision matrix is SURVEY 8d's P = L L^T, seed 123)
        prec = np.linalg.inv(cov)
        logp = lambda x: -0.5 * np.sum((x @ prec) * x, axis=1)
        su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
        den = bfa.SurrogateDensity(su)
        x_fit = 1.5 * sobol_normal(2 * su.n_param, d, seed=seed)   # (broader than the posterior: DESIGN.md section 5)
        fit(den, x_fit, logp(x_fit), 'fit_ms')
        x0 = sobol_normal(C, d, seed=seed + 1)
        what = ('config 2: %d chains x 32-d correlated Gaussian (P = L L^T, SURVEY 8d), quadratic PolyModel P = %d fitted on 2 P '
                'Sobol-normal points, bound on; NUTS defaults.  Departure from SURVEY 8d: the fit points are drawn 1.5 x wider than N(0, I) '
                '(with a training set as tight as the posterior the chains leak through the bound in 32 dimensions: DESIGN.md section 5)' % (C, su.n_param))
        r, s_, _ = _sampler_block(ctx, den, x0, seed, 0.8, n_adapt, iters, steps, cpu_seconds, what)
        var_ratio = float(np.mean(s_.reshape(-1, d).var(0).cpu().numpy() / np.diag(cov)))
        return dict(r, posterior_variance_ratio=var_ratio, **t_fit)
    if name == 'banana_decay':
        d, C = 64, chains or 4096
        logp = banana_logp(d)
        su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
        den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
        x_fit = rng.normal(size=(2 * su.n_param, d))
        fit(den, x_fit, logp(x_fit), 'fit_0_ms')
        x0 = x_fit[rng.integers(0, x_fit.shape[0], C)] * 0.5
        what = ('config 3: %d chains x 64-d rotated banana (Q = 0.01), quadratic surrogate P = %d fitted on 2 P N(0, I) points, '
                'bound and decay on; round %%d.  Departures from SURVEY 8d: the fit points are pseudo-random N(0, I) draws, not Sobol-normal, '
                'and the density carries the decay term the reference\'s GBS recipes use (core/density.py:740-746) -- without it the '
                'chains run away along the first fit\'s indefinite quadratic form (DESIGN.md section 5)' % (C, su.n_param))
        r0, s, st = _sampler_block(ctx, den, x0, seed, 0.8, n_adapt, iters, steps, cpu_seconds, what % 0)
        if os.environ.get('BENCH_ROUND0_ONLY'):   # (tools/profile_configs.sh: counters of the first round's kernel on its own)
            return r0
        # one refit cycle (core/recipe.py:1074-1155 without the cut-off): 2 P of round 0's samples by their logq
        xs, lq = s.reshape(-1, d).cpu().numpy(), st[:, :, 0].reshape(-1).cpu().numpy()
        ok = np.isfinite(lq) & np.all(np.isfinite(xs), axis=1)
        t0 = time.perf_counter()
        x_new, lp_new, _ = select_fit_points(xs[ok], lq[ok], logp, 2 * su.n_param, logp_cutoff=False)
        t_sel = (time.perf_counter() - t0) * 1e3
        fit(den, x_new, lp_new, 'fit_1_ms')
        x0b = x_new[rng.integers(0, x_new.shape[0], C)]
        r1, _, _ = _sampler_block(ctx, den, x0b, seed + 1, 0.8, n_adapt, iters, steps, 0., what % 1)
        return dict(r0, round_1={k: r1[k] for k in ('value', 'ms_per_launch', 'mean_tree_size', 'max_tree_depth', 'divergence_rate',
                                                    'mean_accept', 'chain_layout', 'roofline', 'group_trips') if k in r1},
                    refit={'select_and_true_logp_ms': t_sel, **t_fit, 'n_fit_points': int(x_new.shape[0])})
    if name == 'funnel':
        d, C = 64, chains or 4096
        logp = funnel_logp(d)
        su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
        den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
        x_fit = rng.normal(size=(2 * su.n_param, d))
        fit(den, x_fit, logp(x_fit), 'fit_ms')
        x0 = x_fit[rng.integers(0, x_fit.shape[0], C)] * 0.5
        what = ('config 4 (one GPU of 8): %d chains x 64-d funnel (a = 1, b = 0.5), quadratic surrogate P = %d fitted on 2 P N(0, I) '
                'points, bound and decay on, target_accept 0.95.  Departures from SURVEY 8d: pseudo-random N(0, I) fit points (not '
                'Sobol-normal) and the decay term (core/density.py:740-746), as in the reference\'s funnel-gbs notebook' % (C, su.n_param))
        r, _, _ = _sampler_block(ctx, den, x0, seed, 0.95, n_adapt, iters, steps, cpu_seconds, what)
        return dict(r, **t_fit)
    if name == 'cubic128':
        d, C = 128, chains or 1024
        logp, chol = planck_like_logp(d)
        m16 = np.arange(16)
        su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic'), bfa.PolyConfig('cubic-2', input_mask=m16),
                            bfa.PolyConfig('cubic-3', input_mask=m16)], input_size=d, output_size=1)
        den = bfa.SurrogateDensity(su)
        x_fit = rng.normal(size=(2 * su.n_param, d)) @ chol.T
        fit(den, x_fit, logp(x_fit), 'fit_ms')
        x0 = x_fit[rng.integers(0, x_fit.shape[0], C)] * 0.5
        what = ('config 5 (one GPU of 8): %d chains x 128-d Planck-18-like synthetic logp (cond 1e4 Gaussian + cubic terms on 16 '
                'inputs), cubic-cross PolyModel P = %d fitted on 2 P points, bound on.  Departure from SURVEY 8d: the fit points are '
                'pseudo-random draws from the target\'s Gaussian part (not Sobol-normal)' % (C, su.n_param))
        r, _, _ = _sampler_block(ctx, den, x0, seed, 0.8, n_adapt, iters, steps, cpu_seconds, what)
        return dict(r, **t_fit)
    raise ValueError(name)


def evidence_block(ctx, seed, chains=1024, n_iter=340, n_warmup=120, sit_iter=6):
    """BASELINE config 5's last clause, "evidence via GBS", at config 5's size on the device path: the config-5 surrogate (d = 128,
    linear + quadratic + cubic-2 + cubic-3 on 16 inputs, P = 9201) fitted on the GAUSSIAN part of the Planck-like target (the
    cubic perturbation switched off, so that the evidence has a closed form: log Z = d/2 log 2 pi + 1/2 log det Sigma), sampled by
    `sample()` with 1024 chains, and the samples handed to GBS (evidence/gaussianized.py:179-216: SIT, transforms/sit.py:223-459,
    fitted on the first half, bridge sampling on the second half and as many draws from the SIT).  Wall clock per stage."""
    import warnings
    import torch
    import bayesfast_amd as bfa
    from bayesfast_amd.workloads import planck_like_logp
    rng = np.random.default_rng(seed)
    d = 128
    logp, chol = planck_like_logp(d, amp=0.)
    m16 = np.arange(16)
    su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic'), bfa.PolyConfig('cubic-2', input_mask=m16),
                        bfa.PolyConfig('cubic-3', input_mask=m16)], input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
    den = bfa.SurrogateDensity(su)
    x_fit = rng.normal(size=(2 * su.n_param, d)) @ chol.T * 1.3
    out = {'workload': 'config 5, evidence via GBS: %d chains x 128-d, the cubic-cross surrogate (P = %d) fitted on the Gaussian part of the '
                       'Planck-like target (cond 1e4; closed-form log Z), sample() with %d iterations (%d warm-up) per chain, then GBS with %d '
                       'SIT iterations on half of the kept samples and bridge sampling on the other half' % (
                           chains, su.n_param, n_iter, n_warmup, sit_iter)}

    def timed(key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        out[key] = (time.perf_counter() - t0) * 1e3
        return r

    y_fit = logp(x_fit)   # (the true density on the host: not part of the path)
    timed('fit_ms', lambda: den.fit(x_fit, y_fit))
    tt = timed('sample_ms', lambda: bfa.sample(den, {'n_chain': chains, 'n_iter': n_iter, 'n_warmup': n_warmup, 'random_generator': seed},
                                               verbose=False))
    n_kept = chains * (n_iter - n_warmup)
    gbs = bfa.GBS(sit=dict(n_iter=sit_iter, random_generator=5), n_q=n_kept // 2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz, err = timed('gbs_ms', lambda: gbs(tt, den.logp))
    exact = 0.5 * d * np.log(2. * np.pi) + float(np.sum(np.log(np.diag(chol))))
    out.update({'log_z': float(logz), 'log_z_err': float(err), 'log_z_exact': exact, 'abs_error_in_sigma': float(abs(logz - exact) / max(err, 1e-300)),
                'samples_kept': int(n_kept), 'n_call': int(tt.n_call), 'chains': int(chains), 'dim': d,
                'sample_leapfrog_steps_per_sec': float(tt.n_call) / (out['sample_ms'] * 1e-3),
                'sit_ms_per_iteration': out['gbs_ms'] / sit_iter,
                'note': 'gbs_ms is SIT fit + draws + four logq / logp passes + the bridge iteration; sit_ms_per_iteration is gbs_ms / SIT iterations (an upper bound of one)'})
    return out


CONFIG_BLOCKS = ('gauss32', 'banana_decay', 'funnel', 'cubic128', 'des_pipeline')
CONFIG_KEYS = ('config2', 'config3', 'config4', 'config5', 'pipeline_des')


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


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-child':
        return _cpu_child_main(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--chains', type=int, default=4096, help='chains per GPU')
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--iters', type=int, default=250, help='NUTS iterations per step = per launch (the launch length DeviceChains.run uses)')
    ap.add_argument('--seed', type=int, default=2024)
    ap.add_argument('--backend', default='nccl', help="process-group backend: 'nccl' (= RCCL; default) or 'gloo' (plumbing tests)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fit', action='store_true', help='skip the (untimed, separately reported) surrogate fit')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary figures (hetero workload, refit cycle)')
    ap.add_argument('--no-configs', action='store_true', help="skip the blocks on the BASELINE configs' own targets (config2/3/4/5)")
    ap.add_argument('--workload', default=None, choices=CONFIG_BLOCKS + ('evidence128',),
                    help='run ONE config block only and print it (profiling: rocprofv3 -- python3 bench.py --workload funnel)')
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
    from bayesfast_amd.device import DeviceContext, DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec, B_STEP_BYTES, flops_per_leapfrog
    from bayesfast_amd import _lib

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

    d, C = a.dim, a.chains
    spec, cov = correlated_gaussian_spec(d)
    ctx = DeviceContext(dev_index)
    if a.workload:  # one config block on its own (the profiles of profiles/r03_config{3,4,5}.json come from these commands)
        with torch.cuda.device(ctx.device):
            if a.workload == 'evidence128':
                blk = evidence_block(ctx, a.seed, chains=1024 if a.chains == 4096 else a.chains)
            else:
                blk = config_block(a.workload, ctx, a.seed, cpu_seconds=0. if a.no_cpu_baseline else 4.,
                                   chains=None if a.chains == 4096 else a.chains)
        print(json.dumps({'config_block': a.workload, **blk}))
        return
    with torch.cuda.device(ctx.device):
        dens = DeviceDensity(spec, ctx)
        # chain starts: N(0, I) rows (core/sample.py:111-112), one global array sliced per rank
        x0 = np.random.default_rng(a.seed).normal(size=(world * C, d))[rank * C:(rank + 1) * C]
        chains = DeviceChains(dens, x0, seed=a.seed, first_stream=rank * C)
        # set-up, like the fit of the surrogate: the NUTS adaptation (step size, diagonal metric) of the chains, a fixed
        # N_ADAPT iterations whatever --warmup is; the W warm-up steps and the K timed steps are all post-adaptation
        # launches of the same steady-state transition loop
        n_warm_iter = N_ADAPT
        kw = dict(n_warmup=n_warm_iter, check=False)
        samples = ctx.empty((C, a.iters, d))
        stats = ctx.empty((C, a.iters, _lib.STAT_STRIDE))
        chains.run(n_warm_iter, 'NUTS', **kw)

        for _ in range(a.warmup):
            chains.run(a.iters, 'NUTS', samples=samples, stats=stats, **kw)
        chains.raise_on_error()
        lf0 = chains.total_leapfrog

        def sync():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        # measurement hook of the group kernel: trips, and trips that executed the bound's H (x - mu) tiles
        gcount = torch.zeros(4, dtype=torch.int64, device=ctx.device)
        import ctypes
        _lib.debug_buffer('group_counters', gcount)
        sync()
        t0 = time.perf_counter()
        for k in range(a.steps):
            ev[k][0].record(ctx.stream)  # HIP events on the stream the kernel is launched on
            chains.run(a.iters, 'NUTS', samples=samples, stats=stats, **kw)
            ev[k][1].record(ctx.stream)
        sync()
        elapsed = time.perf_counter() - t0
        chains.raise_on_error()
        _lib.debug_buffer('group_counters', None)
        g_trips, g_trips_h = [int(v) for v in gcount.cpu().numpy()[:2]]
        n_lf = chains.total_leapfrog - lf0
        kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) if a.steps else 0.
        st_last = stats.cpu().numpy()
        kname = _lib.last_kernel
        kernel_name = kname()

        red_dev = ctx.device if (dist is None or a.backend == 'nccl') else torch.device('cpu')
        tot = torch.tensor([float(n_lf)], dtype=torch.float64, device=red_dev)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total_lf, elapsed_max = float(tot.item()), float(tmax.item())
        # the ONE exchange step of the path: the refit's selection of 2 P of ALL ranks' samples by their logq
        # (core/recipe.py:1024-1025,1074) -- every rank sorts its own shard, four collectives move quantile keys, counts,
        # candidates and the selected rows (core/refit.py: select_rows_sharded).  Timed on its own, never in `value`.
        exchange = None
        if dist is not None:
            from bayesfast_amd.core.refit import select_rows_sharded
            from bayesfast_amd.utils.resample import SystematicResampler
            n_sel = 2 * (1 + d + d * (d + 1) // 2)
            n_rows = C * a.iters
            if world * n_rows >= n_sel:
                rk = SystematicResampler(require_unique=False).ranks(world * n_rows, n_sel)
                xl, ql = samples.reshape(-1, d), stats[:, :, 0].reshape(-1).contiguous()
                est = {}
                select_rows_sharded(ql, xl, rk, n_loc_max=n_rows)   # untimed first call (sort workspace, communicator)
                tms = []
                for _ in range(3):
                    sync()
                    t1 = time.perf_counter()
                    rows, vals = select_rows_sharded(ql, xl, rk, stats=est, n_loc_max=n_rows)
                    sync()
                    tms.append((time.perf_counter() - t1) * 1e3)
                tx = torch.tensor([min(tms)], dtype=torch.float64, device=red_dev)
                dist.all_reduce(tx, op=dist.ReduceOp.MAX)
                chk = torch.tensor([float(vals.sum())], dtype=torch.float64, device=red_dev)
                chk_all = [torch.zeros_like(chk) for _ in range(world)]
                dist.all_gather(chk_all, chk)
                exchange = {'ms': float(tx.item()), 'wire_bytes_per_rank': int(est['wire_bytes']), 'collectives': int(est['collectives']),
                            'rows_selected': int(n_sel), 'rows_per_rank': int(n_rows), 'splitters_per_rank': int(est['n_splitter']),
                            'candidates_per_rank_and_row': int(est['candidates_per_rank']),
                            'identical_on_all_ranks': bool(all(float(c.item()) == float(chk.item()) for c in chk_all)),
                            'includes': 'local device sort of %d keys + 4 collectives (%s), max over ranks, best of 3' % (n_rows, a.backend)}

        # What sample() adds per launch when the chains are sharded (core/sample.py:104-105, DeviceChains._note_trees): the histogram
        # of the launch's tree sizes summed over the ranks, so that every rank chooses the same layout for the next launch -- a
        # stream synchronisation and one 32 KB all-reduce.  Not part of `value` (the timed loop above runs one layout); reported so
        # that the weak-scaling loss of a sharded sample() has a prior.
        vote = None
        if dist is not None:
            from bayesfast_amd import parallel
            chains.hist_reduce = parallel.all_reduce_sum
            tv = []
            for _ in range(5):
                sync()
                t1 = time.perf_counter()
                chains._note_trees(stats, 0, a.iters, 'NUTS')
                tv.append((time.perf_counter() - t1) * 1e3)
            chains.hist_reduce = None
            tvx = torch.tensor([min(tv[1:])], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tvx, op=dist.ReduceOp.MAX)
            vote = {'ms_per_launch': float(tvx.item()), 'share_of_a_launch': float(tvx.item()) / (elapsed_max / max(a.steps, 1) * 1e3),
                    'includes': 'histogram of the last 32 iterations\' tree sizes (device), stream synchronisation, all-reduce of 4096 int64 '
                                '(%s), max over ranks, best of 4' % a.backend}

    if rank == 0:
        value = total_lf / elapsed_max
        lf_per_launch = n_lf / max(a.steps, 1)
        use_bound = bool(spec['poly']['use_bound'])
        flops = flops_per_leapfrog(d, use_bound) * lf_per_launch
        bytes_alg = B_STEP_BYTES(d) * lf_per_launch
        ach_tf = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms else 0.
        peak_tf = 78.6  # FP64 MFMA, 256 CUs x 4 SIMDs x 2.4 GHz x 2048 flop / 64 cyc; 77.7 measured (profiles/r01_probe_mfma_f64.log)
        # HBM-side bytes per launch: a STORED profile value (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,
        # profiles/hbm_traffic.json), used only when it was taken on this kernel at this dimension
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath) and not os.environ.get('BFHIP_LIBRARY'):
            try:
                tj = json.load(open(tpath))
                if tj.get('dim') == d and tj.get('kernel') == kernel_name:
                    traffic = tj['hbm_bytes_per_leapfrog'] * lf_per_launch
            except Exception:
                traffic = None
        exec_share = 0.5 * (1. + g_trips_h / g_trips) if (g_trips and use_bound) else 1.
        ts_last = st_last[:, :, _lib.NSTATS.index('tree_size')]
        out = {
            'metric': 'leapfrog steps/sec (all chains), %d chains x %d-d quadratic surrogate' % (C, d),
            'value': value, 'unit': 'leapfrog steps/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed_max / max(a.steps, 1) * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%d chains/GPU x %d-d correlated Gaussian (SURVEY 8d config-2 family), '
                                   "PolyModel('quadratic') surrogate = linear+quadratic, bound on; NUTS defaults "
                                   '(diag-adapt metric, target_accept 0.8, max_treedepth 10).  The surrogate is exact for this '
                                   'target and no sample leaves the bound: %.1f %% of the trees have %d leaves, so the chains of a '
                                   'workgroup run in step -- the best case of the path; the BASELINE configs\' own targets '
                                   '(banana round 0 / round 1, funnel, cubic-cross) are the config2..config5 blocks and the '
                                   'config3_round*_value keys of this line' % (
                                       C, d, 100. * float(np.mean(ts_last == np.median(ts_last))), int(np.median(ts_last))),
                       'chains_per_gpu': C, 'dim': d, 'nuts_iterations_per_step': a.iters,
                       'nuts_warmup_iterations': n_warm_iter,
                       'mean_tree_size': float(ts_last.mean()),
                       'parallelism': 'chains sharded over %d rank(s), no data-path collective' % world},
            'roofline': {'bound': 'mfma', 'achieved': ach_tf * exec_share, 'peak': peak_tf, 'unit': 'TFLOP/s',
                         'frac': ach_tf * exec_share / peak_tf, 'traffic': traffic,
                         'achieved_algorithmic': ach_tf, 'frac_algorithmic': ach_tf / peak_tf,
                         'kernel': kernel_name, 'kernel_ms_per_launch': kernel_ms,
                         'traffic_source': None if traffic is None else 'stored profile (profiles/hbm_traffic.json), not this run',
                         'flops_per_leapfrog': flops_per_leapfrog(d, use_bound),
                         'group_trips': g_trips, 'group_trips_with_bound_tiles': g_trips_h,
                         'executed_share_of_algorithmic_flops': exec_share,
                         'flops_note': 'achieved / frac count the flops the kernel EXECUTED; *_algorithmic count S x and H (x - mu) per '
                                       'step (4 d^2) whether executed or not.  The lane-per-chain kernels leave the H tiles out of a '
                                       'trip when lam_max(H) |x - mu|^2 < alpha^2 proves all 16 chains of the group inside the bound '
                                       '(identical results); on this workload that is nearly every trip (group_trips_with_bound_tiles of '
                                       'group_trips ran them), so about half of the algorithmic flops are decided, not executed'},
            'roofline_hbm_algorithmic': {'bound': 'hbm', 'achieved': bytes_alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.,
                                         'peak': 8000., 'unit': 'GB/s',
                                         'frac': (bytes_alg / (kernel_ms * 1e-3) / 1e9 / 8000.) if kernel_ms else 0.,
                                         'bytes_per_leapfrog': B_STEP_BYTES(d)},
        }
        if exchange is not None:
            out['refit_exchange'] = exchange
        if vote is not None:
            out['layout_vote'] = vote
        try:   # the AS-SHIPPED reference's rate: a stored measurement of the build container (tools/time_reference.py), never timed here
            rt = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_timing.json')))
            out['reference_as_shipped'] = {'leapfrog_steps_per_sec_per_core': rt['leapfrog_steps_per_sec_per_core'],
                                           'polymodel_fit_s': rt.get('polymodel_fit_s'), 'host': rt['host']['cpu'],
                                           'source': 'tests/golden/reference_timing.json (build container; the Python reference does not travel)'}
        except Exception:
            pass
        if not a.no_fit and not a.no_cpu_baseline and world == 1:
            try:
                with torch.cuda.device(ctx.device):
                    out['fit'] = fit_timing(d, cov)
            except Exception as ex:  # the fit is a side measurement; the headline line must still print
                out['fit'] = {'error': repr(ex)}
        if not a.no_extras and world == 1:
            try:
                with torch.cuda.device(ctx.device):
                    out['hetero'] = hetero_rate(ctx, d, C, a.seed, a.iters)
                    out['scaled_inputs'] = scaled_inputs_rate(ctx, d, C, a.seed, a.iters)
                    out['refit_cycle'] = refit_cycle(d, cov, C, a.seed)
                    out.update(other_samplers(ctx, d, cov, C, a.seed))
            except Exception as ex:  # side measurements; the headline line must still print
                out['extras_error'] = repr(ex)
        if not a.no_configs and world == 1:
            for name, key in zip(CONFIG_BLOCKS, CONFIG_KEYS):
                try:
                    with torch.cuda.device(ctx.device):
                        out[key] = config_block(name, ctx, a.seed, cpu_seconds=0. if a.no_cpu_baseline else 4.)
                except Exception as ex:  # side measurements; the headline line must still print
                    out[key] = {'error': repr(ex)}
            try:   # config 5's "evidence via GBS" at config 5's size
                with torch.cuda.device(ctx.device):
                    out['config5_evidence'] = evidence_block(ctx, a.seed)
            except Exception as ex:
                out['config5_evidence'] = {'error': repr(ex)}
            c3 = out.get('config3', {})   # SURVEY 8d's headline config, beside the benign-target `value`
            out['config3_round0_value'] = c3.get('value')
            out['config3_round1_value'] = (c3.get('round_1') or {}).get('value')
        if not a.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(spec, d, n_warm_iter, a.seed)
        elif not a.no_cpu_baseline:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
