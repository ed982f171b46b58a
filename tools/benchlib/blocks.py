"""bench.py's side blocks: the BASELINE configs' own targets one GPU's shard each, the DES-shaped pipeline
density, the evidence stage, and the secondary figures (heterogeneous trees, scaled inputs, other samplers, the refit cycle
through the package API, the fit alone).  Every block is a dict on the bench line; none of them is part of `value`."""
import json
import os
import time

import numpy as np

from . import cpu

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N_ADAPT = 750  # NUTS adaptation iterations of the secondary figures on the benign Gaussian


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


def source_hash():
    """sha256 over the HIP sources the library is built from (bayesfast_amd/csrc/*.h, *.hip, Makefile): stored with every traffic
    profile, so that a kernel changed without a re-profile is noticed instead of read."""
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(ROOT, 'bayesfast_amd', 'csrc')
    for fn in sorted(os.listdir(cs)):
        if fn.endswith(('.h', '.hip')) or fn == 'Makefile':
            h.update(fn.encode())
            with open(os.path.join(cs, fn), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def stored_traffic(kernel, chains, dim, mean_tree_size, leapfrogs_per_launch):
    """HBM-side bytes per launch of `kernel` from a STORED profile (profiles/config_traffic.json: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, tools/profile_configs.sh), used only when kernel, shape and tree size match AND the profile
    was taken on the sources of this build (`source_hash`); otherwise (None, why)."""
    try:
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'config_traffic.json')))
    except Exception:
        return None, 'no stored profile'
    why = 'no stored profile of this kernel at this shape'
    for e in tj.values():
        if not isinstance(e, dict):
            continue
        for blk in e.values():
            if not isinstance(blk, dict):
                continue
            if (blk.get('kernel_named_by_library') == kernel and blk.get('chains') == int(chains) and blk.get('dim') == int(dim)
                    and abs(blk.get('mean_tree_size', -1.) / mean_tree_size - 1.) < 0.25 and 'hbm_bytes_per_leapfrog' in blk):
                if blk.get('source_hash') != source_hash():
                    why = 'stored profile is STALE (taken on other kernel sources: %s, this build %s): re-run tools/profile_configs.sh' % (
                        blk.get('source_hash'), source_hash())
                    continue
                return blk['hbm_bytes_per_leapfrog'] * leapfrogs_per_launch, (
                    'stored profile of this build\'s sources (profiles/config_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), not this run')
    return None, why


def cpu_job(ch, spec, d):
    """What the CPU baseline of a block needs from the device chains: (spec, positions, step size, diagonal metric)."""
    step = float((ch.field('log_bar').exp() * d**0.25).mean())   # what _get_step_size hands to the next round
    return spec, ch.field('q').cpu().numpy(), step, ch.field('var').mean(0).cpu().numpy()


def _sampler_block(ctx, den, x0, seed, target_accept, n_adapt, iters, steps, cpu_seconds, what, first_stream=0, warmup=1, sync=None,
                   hist_reduce=None, chains_per_rank=None):
    """Adapt n_adapt NUTS iterations on the device, then time `steps` launches of `iters` iterations (HIP events on the
    launch stream).  Returns (block dict, samples (C, iters, d), stats) of the last launch."""
    import torch
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import B_STEP_BYTES, flops_per_leapfrog_spec, decay_shares_bound
    from bayesfast_amd import _lib
    dd = den.device(ctx)
    C, d = x0.shape
    ch = DeviceChains(dd, x0, seed=seed, first_stream=first_stream)
    if hist_reduce is not None:   # sharded chains: the layout of every launch is decided from ALL ranks' trees, as sample() does
        ch.hist_reduce, ch.n_chain_rule = hist_reduce, chains_per_rank
    kw = dict(n_warmup=n_adapt, check=False, target_accept=target_accept)
    ch.run(n_adapt, 'NUTS', **kw)
    s = ctx.empty((C, iters, d))
    st = ctx.empty((C, iters, _lib.STAT_STRIDE))
    for _ in range(max(int(warmup), 1)):
        ch.run(iters, 'NUTS', samples=s, stats=st, **kw)   # untimed post-adaptation launches
    ch.raise_on_error()
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if sync is None:
        sync = torch.cuda.synchronize
    # measurement hook of the lane-per-chain kernels: trips, trips with the bound's tiles, with a late exchange, without the early one
    gcount = torch.zeros(4, dtype=torch.int64, device=ctx.device)
    _lib.debug_buffer('group_counters', gcount)
    sync()   # barrier (under torch.distributed) + torch.cuda.synchronize()
    t_wall = time.perf_counter()
    e0.record(ctx.stream)   # HIP events on the stream the kernels are launched on
    for _ in range(steps):
        ch.run(iters, 'NUTS', samples=s, stats=st, **kw)
    e1.record(ctx.stream)
    sync()
    t_wall = time.perf_counter() - t_wall
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
        if kname().startswith('bf_group_kernel') and m_out > nf:
            # the group kernel leaves the zero tiles of the triangular C' = R out of both contractions: (NT + 1) / (2 NT) of the tile k-steps
            nt = -(-nf // 16)
            exec_share = (fl - 4 * m_out * nf + 4 * nf * nf * (nt + 1) / (2. * nt)) / fl
    if spec.get('chi2') is None and decay_shares_bound(spec) and not (kname().startswith(('bf_nuts_pipe_kernel', 'bf_lone_kernel')) and ', 2, ' in kname()):
        # the decay term shares the bound's matrix: one product would do; every kernel but the two-matrix forms of the pipelined and the
        # latency kernel still runs it a second time (the same numbers)
        exec_share = (fl + 2 * d * d) / fl
    out = {'workload': what, 'value': n_lf / (ms * 1e-3), 'unit': 'leapfrog steps/sec', 'chains': int(C), 'dim': int(d),
           'leapfrogs_timed': int(n_lf), 'wall_s_timed': t_wall, 'steps_timed': int(steps),
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
        if spec.get('chi2') is None and spec['poly'].get('use_bound'):
            # the lane-per-chain kernels leave the bound's (and the decay term's) tiles out of a trip whose 16 chains are PROVEN
            # inside (identical results): those flops are decided, not executed
            n_prov = 1 + (1 if spec.get('use_decay') else 0)                      # matrices a trip with bound tiles EXECUTES beside S
            n_mat = 1 + n_prov - (1 if decay_shares_bound(spec) else 0)           # matrices the algorithm needs (fl)
            share = (1. + n_prov * gc[1] / gc[0]) / n_mat
            rf = out['roofline']
            rf.update({'achieved': ach * share, 'frac': ach * share / 78.6, 'executed_share_of_algorithmic_flops': share})
    out['roofline']['traffic'], out['roofline']['traffic_source'] = stored_traffic(out['roofline']['kernel'], C, d, out['mean_tree_size'], n_lf / steps)
    if cpu_seconds > 0:
        try:
            out['cpu_baseline'] = cpu.fixed_rate(*cpu_job(ch, spec, d), seed, target_accept, cpu_seconds)
        except Exception as ex:
            out['cpu_baseline'] = {'error': repr(ex)}
    out['_chains'] = ch   # (for the caller: positions, adapted state; popped before the block is printed)
    return out, s, st


def _layout_of(kernel, requested):
    """The layout that RAN (a requested 'split' runs the group kernel where bf_split_kernel has no instantiation)."""
    for key, lay in (('bf_split_kernel', 'split'), ('bf_group_kernel', 'group'), ('bf_nuts_pipe_kernel', 'wave'), ('bf_sampler_kernel', 'wave')):
        if kernel.startswith(key):
            return lay
    return requested


def config_block(name, ctx, seed, cpu_seconds=4., chains=None, iters=None, steps=2, n_adapt=None):
    r = _config_block(name, ctx, seed, cpu_seconds, chains, iters, steps, n_adapt)
    r.pop('_chains', None)
    return r


def _config_block(name, ctx, seed, cpu_seconds=4., chains=None, iters=None, steps=2, n_adapt=None):
    """The BASELINE configs' own targets, one GPU's shard each, through the package API: fit the surrogate
    on 2 P points of the true model, adapt, time post-adaptation launches.
      banana_decay : config 3 -- 64-d rotated banana, quadratic surrogate (P = 2145) WITH the decay term the reference's
                     recipe uses for such targets (core/density.py:740-746), 4096 chains, and ONE refit cycle (2 P of the
                     first round's samples by their logq, true logp, refit, sample again); both rounds reported
      funnel       : config 4's shard -- 64-d funnel, target_accept 0.95, 4096 chains, decay on
      cubic128     : config 5's shard -- d = 128, linear + quadratic + cubic-2 + cubic-3 on 16 inputs (P = 9201), 1024 chains"""
    import torch
    import bayesfast_amd as bfa
    from bayesfast_amd.workloads import banana_logp, funnel_logp, planck_like_logp, sobol_normal
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
        from bayesfast_amd.workloads import correlated_gaussian_spec
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
        r = config3_rounds(ctx, seed, chains or 4096, iters, n_adapt, steps, 1, cpu_seconds=cpu_seconds,
                           round0_only=bool(os.environ.get('BENCH_ROUND0_ONLY')))   # (tools/profile_configs.sh: the first round's kernel on its own)
        r0 = r['rounds'][0]
        if len(r['rounds']) == 1:
            return r0
        r1 = r['rounds'][1]
        return dict(r0, round_1={k: r1[k] for k in ('value', 'ms_per_launch', 'mean_tree_size', 'max_tree_depth', 'divergence_rate',
                                                    'mean_accept', 'chain_layout', 'roofline', 'group_trips', 'launch_tail', 'chains', 'dim') if k in r1},
                    refit=r['refit'], both_rounds_value=r['value'])
    if name == 'funnel':
        d, C = 64, chains or 4096
        logp = funnel_logp(d)
        su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
        den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
        x_fit = sobol_normal(2 * su.n_param, d, seed=seed)   # 2 P Sobol-normal points
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
        x_fit = sobol_normal(2 * su.n_param, d, seed=seed) @ chol.T   # Sobol-normal points, coloured by the target's Gaussian part
        fit(den, x_fit, logp(x_fit), 'fit_ms')
        x0 = x_fit[rng.integers(0, x_fit.shape[0], C)] * 0.5
        what = ('config 5 (one GPU of 8): %d chains x 128-d Planck-18-like synthetic logp (cond 1e4 Gaussian + cubic terms on 16 '
                'inputs), cubic-cross PolyModel P = %d fitted on 2 P points, bound on.  Departure from SURVEY 8d: the fit points are '
                'pseudo-random draws from the target\'s Gaussian part (not Sobol-normal)' % (C, su.n_param))
        r, _, _ = _sampler_block(ctx, den, x0, seed, 0.8, n_adapt, iters, steps, cpu_seconds, what)
        return dict(r, **t_fit)
    raise ValueError(name)



class _SpecDensity:
    """A density given as a plain spec (workloads.correlated_gaussian_spec), with the two methods _sampler_block uses."""

    def __init__(self, spec):
        self._spec = spec

    def spec(self):
        return self._spec

    def device(self, ctx):
        from bayesfast_amd.device import DeviceDensity
        return DeviceDensity(self._spec, ctx)


def gauss64_best_case(ctx, seed, chains=4096, iters=250, steps=10, cpu_seconds=6.):
    """The path's best case (the line's `value` until round 5): 64-d correlated Gaussian of the config-2 family at the headline size,
    PolyModel('quadratic') with the bound on, NUTS defaults.  The surrogate is exact and no sample leaves the bound: every tree has 7
    leaves, the chains of a workgroup run in step and the lane-per-chain kernels prove the bound in nearly every trip.  750 adaptation
    iterations, then `steps` launches of `iters` iterations; the CPU baseline adapts for itself (tuned evaluation)."""
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 64
    spec, _ = correlated_gaussian_spec(d)
    x0 = np.random.default_rng(seed).normal(size=(chains, d))
    what = ('%d chains x 64-d correlated Gaussian (SURVEY 8d config-2 family), PolyModel(\'quadratic\') surrogate = linear + quadratic, bound on; '
            'NUTS defaults.  The surrogate is exact for this target and no sample leaves the bound: the chains of a workgroup run in step -- the '
            'best case of the path, not a BASELINE config' % chains)
    r, _, st = _sampler_block(ctx, _SpecDensity(spec), x0, seed, 0.8, N_ADAPT, iters, steps, 0., what, warmup=3)
    r.pop('_chains', None)
    if cpu_seconds > 0:
        r['cpu_baseline'] = cpu.adapted_rate(spec, d, N_ADAPT, seed, target_seconds=cpu_seconds)
    return r


def config3_rounds(ctx, seed, chains, iters, n_adapt, steps, warmup, rank=0, world=1, sync=None, cpu_seconds=0., round0_only=False,
                   steps_round1=None):
    """SURVEY 8d "Config 3 (headline)": 64-d rotated banana, `chains` chains per rank, quadratic surrogate (P = 2145) fitted on 2 P
    Sobol-normal points, sampled (round 0); ONE refit cycle -- 2 P of round 0's samples of ALL ranks picked by their logq
    (SystematicResampler, core/recipe.py:1074-1075; core/refit.py: select_rows_sharded), true logp, refit -- and sampled again
    (round 1).  `steps` timed launches of `iters` NUTS iterations in round 0 and `steps_round1` (default: the same) in round 1, each
    round after `n_adapt` adaptation iterations and `warmup` untimed launches.  value = leapfrogs of both rounds' timed launches /
    their time (max over ranks is the caller's business: `wall_s_timed` per round); the fits and the selection are timed beside it.
    Departure from SURVEY 8d, stated on the line: the density carries the decay term the reference's GBS recipes use
    (core/density.py:740-746) -- without it the chains run away along the first fit's indefinite quadratic form."""
    import torch
    import bayesfast_amd as bfa
    from bayesfast_amd import parallel
    from bayesfast_amd.workloads import banana_logp, sobol_normal
    from bayesfast_amd.core.refit import select_rows_sharded
    from bayesfast_amd.utils.resample import SystematicResampler
    d, C = 64, int(chains)
    logp = banana_logp(d)
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
    den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
    n_eval = 2 * su.n_param
    t_ms = {}

    def timed(key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        t_ms[key] = (time.perf_counter() - t0) * 1e3
        return r

    x_fit = sobol_normal(n_eval, d, seed=seed)   # the same points on every rank: every rank fits the same surrogate
    y_fit = logp(x_fit)
    timed('fit_0_ms', lambda: den.fit(x_fit, y_fit))
    # chain starts: Sobol-normal N(0, I) rows (core/sample.py:111-112), one global array sliced per rank
    x0 = sobol_normal(world * C, d, seed=seed + 1)[rank * C:(rank + 1) * C]
    what = ('config 3 (SURVEY 8d headline): %d chains%s x 64-d rotated banana (Q = 0.01), quadratic surrogate P = %d fitted on 2 P '
            'Sobol-normal points, bound and decay on, chains started on Sobol-normal rows; round %%d.  Departure from SURVEY 8d: the density '
            'carries the decay term the reference\'s GBS recipes use (core/density.py:740-746) -- without it the chains run away along the '
            'first fit\'s indefinite quadratic form (DESIGN.md section 5)' % (C, '/GPU' if world > 1 else '', su.n_param))
    hr = parallel.all_reduce_sum if world > 1 else None
    common = dict(first_stream=rank * C, warmup=warmup, sync=sync, hist_reduce=hr, chains_per_rank=C)
    r0, s, st = _sampler_block(ctx, den, x0, seed, 0.8, n_adapt, iters, steps, 0., what % 0, **common)
    ch0 = r0.pop('_chains')
    cpu_rounds = [dict(zip(('spec', 'x_start', 'step_size', 'var'), cpu_job(ch0, den.spec(), d)), target_accept=0.8,
                       leapfrogs=r0['leapfrogs_timed'])] if cpu_seconds > 0 else []
    if round0_only:
        return {'rounds': [r0], 'refit': dict(t_ms), 'value': r0['value']}
    # one refit cycle (core/recipe.py:1074-1155 without the cut-off): 2 P of ALL ranks' round-0 rows by their logq; every rank sorts
    # its own shard, four collectives move quantile keys, counts, candidates and the selected rows (identical on every rank)
    n_rows = C * iters
    rk = SystematicResampler(require_unique=False).ranks(world * n_rows, n_eval)
    est = {}
    xl, ql = s.reshape(-1, d), st[:, :, 0].reshape(-1).contiguous()
    rows, vals = timed('select_ms', lambda: select_rows_sharded(ql, xl, rk, stats=est, n_loc_max=n_rows))
    x_new = rows.cpu().numpy()
    ok = np.all(np.isfinite(x_new), axis=1) & np.isfinite(vals.cpu().numpy())
    x_new = x_new[ok]
    t0 = time.perf_counter()
    lp_new = logp(x_new)   # (the true model on the host: not part of the path)
    t_ms['true_logp_ms'] = (time.perf_counter() - t0) * 1e3
    timed('fit_1_ms', lambda: den.fit(x_new, lp_new))
    # round 1 starts where the reference's recipe starts it: on rows of the refit set (core/recipe.py:1028-1044 hands the previous
    # round's samples on); one global choice sliced per rank
    pick = np.random.default_rng(seed + 2).integers(0, x_new.shape[0], world * C)[rank * C:(rank + 1) * C]
    r1, _, _ = _sampler_block(ctx, den, x_new[pick], seed + 1, 0.8, n_adapt, iters, steps_round1 or steps, 0., what % 1, **common)
    ch1 = r1.pop('_chains')
    out = {'rounds': [r0, r1],
           'refit': dict(t_ms, n_fit_points=int(x_new.shape[0]), wire_bytes_per_rank=int(est.get('wire_bytes', 0)),
                         collectives=int(est.get('collectives', 0))),
           'value': (r0['leapfrogs_timed'] + r1['leapfrogs_timed']) / (r0['wall_s_timed'] + r1['wall_s_timed']),
           'checksum_of_selected_rows': float(vals.sum())}
    if cpu_seconds > 0:
        cpu_rounds.append(dict(zip(('spec', 'x_start', 'step_size', 'var'), cpu_job(ch1, den.spec(), d)), target_accept=0.8,
                               leapfrogs=r1['leapfrogs_timed']))
        try:
            out['cpu_baseline'] = cpu.two_round_rate(cpu_rounds, seed, cpu_seconds)
        except Exception as ex:
            out['cpu_baseline'] = {'error': repr(ex)}
    return out


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
