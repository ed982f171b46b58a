"""GPU parity of the fused NUTS/HMC kernel (through the C ABI) against the CPU oracle.

T1 (trajectory): same xoshiro256++ stream => the device chain reproduces the oracle chain: tree depth,
tree size and divergence flags exactly, positions to 1e-8 over the compared horizon (float64; the only
differences are summation order and libm rounding, which chaotic warm-up trajectories amplify).
T2 (statistical): posterior moments of an exactly-quadratic target vs the analytic values."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEED = 20240917


@pytest.fixture(scope='module')
def ctx():
    from bayesfast_amd.device import get_context
    return get_context(0)


@pytest.fixture(scope='module')
def samp():
    return np.load(os.path.join(G, 'sampler.npz'))


def _spec(z, prefix):
    from specio import rebuild_spec
    return rebuild_spec(z, prefix)


def _oracle_chains(spec, x0, n_iter, n_warmup, sampler='NUTS', first_stream=0, **kw):
    from oracle import oracle as orc
    out = []
    for i in range(x0.shape[0]):
        ch = orc.Chain(x0[i], **{k: v for k, v in kw.items() if k in ('step_size', 'target_accept')})
        rng = orc.make_rng('xoshiro', seed=SEED, stream=first_stream + i)
        if sampler == 'NUTS':
            out.append(orc.nuts_run(spec, ch, rng, n_iter, n_warmup, max_treedepth=kw.get('max_treedepth', 10),
                                    max_change=kw.get('max_change', 1000.)) + (ch,))
        else:
            out.append(orc.hmc_run(spec, ch, rng, n_iter, n_warmup, n_int_step=kw.get('n_int_step', 32),
                                   max_change=kw.get('max_change', 1000.)) + (ch,))
    return out


_LAYOUT = {'v': 'group'}


@pytest.fixture(autouse=True, params=['group', 'wave', 'split'])
def _both_layouts(request):
    """Every test of this module runs on all chain layouts of the sampler (lane per chain: bfhip_group.hip; wave per
    chain: bf_nuts_pipe_kernel / bf_sampler_kernel; lane per chain with integrator and bookkeeper waves: bfhip_split.h, which
    the library runs wherever it applies -- NUTS on the plain surrogate at 33 <= d <= 64 -- and as 'group' elsewhere)."""
    _LAYOUT['v'] = request.param
    yield


def _device_chains(ctx, spec, x0, n_iter, n_warmup, sampler='NUTS', first_stream=0, split=None, **kw):
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd import _lib
    dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=SEED, first_stream=first_stream,
                      step_size=kw.get('step_size', 1.))
    run_kw = {k: v for k, v in kw.items() if k in ('max_treedepth', 'max_change', 'n_int_step', 'target_accept')}
    run_kw['layout'] = _LAYOUT['v']
    if split is None:
        s, st = dc.run(n_iter, sampler, n_warmup=n_warmup, **run_kw)
    else:  # two launches: the state arrays carry the chains across (resume)
        s1, st1 = dc.run(split, sampler, n_warmup=n_warmup, **run_kw)
        s2, st2 = dc.run(n_iter - split, sampler, n_warmup=n_warmup, **run_kw)
        import torch
        s, st = torch.cat([s1, s2], 1), torch.cat([st1, st2], 1)
    names = _lib.NSTATS if sampler == 'NUTS' else _lib.HSTATS
    st = st.cpu().numpy()
    return s.cpu().numpy(), {k: st[:, :, i] for i, k in enumerate(names)}, dc


def _compare_nuts(dev, orc_runs, n_exact, rtol_q=1e-5, n_head=8, tol_head=1e-9):
    """Discrete fields exactly over the whole horizon; positions to 1e-9 on the first n_head iterations and
    to rtol_q overall (warm-up trajectories at step sizes near the stability limit amplify the 1e-16
    summation-order differences by orders of magnitude per iteration)."""
    s, st, dc = dev
    for i, (so, sto, ch) in enumerate(orc_runs):
        for f in ('tree_depth', 'tree_size', 'diverging', 'warmup'):
            assert np.array_equal(st[f][i][:n_exact], sto[f][:n_exact]), (i, f, st[f][i][:n_exact], sto[f][:n_exact])
        np.testing.assert_allclose(s[i][:n_head], so[:n_head], rtol=tol_head, atol=tol_head, err_msg='chain %d' % i)
        np.testing.assert_allclose(s[i][:n_exact], so[:n_exact], rtol=rtol_q, atol=rtol_q, err_msg='chain %d' % i)
        for f in ('logp', 'energy', 'mean_tree_accept', 'step_size', 'step_size_bar', 'energy_change', 'max_energy_change'):
            np.testing.assert_allclose(st[f][i][:n_head], sto[f][:n_head], rtol=1e-8, atol=1e-8, err_msg=f)
            np.testing.assert_allclose(st[f][i][:n_exact], sto[f][:n_exact], rtol=1e-4, atol=1e-4, err_msg=f)


@pytest.mark.parametrize('name,n_chain,n_iter,n_warmup', [('plain16', 5, 40, 25), ('d64', 19, 30, 20), ('full5', 3, 25, 15)])
def test_nuts_trajectories_match_oracle(ctx, samp, name, n_chain, n_iter, n_warmup):
    spec = _spec(samp, name + '.')
    rng = np.random.default_rng(3)
    x0 = rng.normal(size=(n_chain, spec['d'])) * 0.5
    dev = _device_chains(ctx, spec, x0, n_iter, n_warmup)
    orc_runs = _oracle_chains(spec, x0, n_iter, n_warmup)
    _compare_nuts(dev, orc_runs, n_iter)
    # adapted metric and n_leapfrog accounting
    s, st, dc = dev
    for i, (so, sto, ch) in enumerate(orc_runs):
        np.testing.assert_allclose(dc.field('var')[i].cpu().numpy(), ch.vec('var'), rtol=1e-5)
    assert dc.total_leapfrog == int(sum(r[1]['tree_size'].sum() for r in orc_runs))


def test_nuts_resume_and_shard_invariance(ctx, samp):
    """Two launches == one launch; a chain's result depends on its GLOBAL stream index only."""
    spec = _spec(samp, 'plain16.')
    rng = np.random.default_rng(4)
    x0 = rng.normal(size=(20, 16)) * 0.5
    s_all, st_all, _ = _device_chains(ctx, spec, x0, 30, 20)
    s_split, st_split, _ = _device_chains(ctx, spec, x0, 30, 20, split=13)
    assert np.array_equal(s_all, s_split)
    assert np.array_equal(st_all['tree_size'], st_split['tree_size'])
    s_tail, st_tail, _ = _device_chains(ctx, spec, x0[7:], 30, 20, first_stream=7)
    assert np.array_equal(s_all[7:], s_tail)


@pytest.mark.parametrize('d', [64, 32, 10])
def test_tail_matvec_is_bit_identical_to_mfma_path(ctx, d):
    """The plain kernel switches its matvec from MFMA tiles to per-row FMA chains while at most 4 chains of a
    16-chain group are evaluating.  v_mfma_f64_16x16x4_f64 accumulates every entry as one sequential fma chain
    over k, which the FMA path repeats, so samples and statistics must agree bit for bit with the MFMA-only
    run (bfhip_debug_set('tail_max', 0) disables the switch)."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(1).normal(size=(150, d))  # 9 full groups and a ragged one
    out = {}
    try:
        _lib.debug_set('no_pipe', 1)  # (at d <= 64 NUTS runs on the pipelined kernel by default, which has no tail path)
        for tm in (0, 4):
            _lib.debug_set('tail_max', tm)
            dc = DeviceChains(dens, x0, seed=5)
            s, st = dc.run(40, 'NUTS', n_warmup=20, layout='wave')
            out[tm] = (s.cpu().numpy(), st.cpu().numpy())
    finally:
        _lib.debug_set('tail_max', 4)
        _lib.debug_set('no_pipe', 0)
    assert np.array_equal(out[0][0], out[4][0])
    assert np.array_equal(out[0][1], out[4][1], equal_nan=True)


@pytest.mark.parametrize('case', ['balanced', 'leaky_bound', 'divergent', 'd40', 'd10', 'bounded', 'decay', 'hmc'])
def test_group_kernel_bound_proof_never_changes_results(ctx, case):
    """The group kernel leaves out the H (x - mu) tiles of a trip when lam_max(H) |x - mu|^2 < alpha^2 proves every chain of
    the group inside the bound (modules/poly.py:467-469 decided without the matvec).  With the proof switched off every
    trip computes them: samples, statistics, adapted state, random streams and the leapfrog count must be bit-identical --
    with groups that always skip, groups mostly outside the bound (never skip), mixed groups, transforms, decay, HMC."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    d = int(case[1:]) if case[0] == 'd' and case[1:].isdigit() else 64
    spec, _ = correlated_gaussian_spec(d, fit_scale=1.0 if case == 'leaky_bound' else 1.5)
    if case == 'bounded':
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * 16, dtype=np.uint8))
    if case == 'decay':
        pm = spec['poly']
        spec = dict(spec, use_decay=True, decay_mu=np.asarray(pm['mu']), decay_hess=np.asarray(pm['hess']),
                    decay_alpha2=float(pm['alpha'])**2 * 0.5, decay_gamma=0.1)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(2).normal(size=(150, d)) * (3. if case == 'divergent' else (0.3 if case == 'bounded' else 1.))
    kw = {'divergent': dict(max_change=5.)}.get(case, {})
    sampler = 'HMC' if case == 'hmc' else 'NUTS'
    out = {}
    hook = lambda v: _lib.debug_set('no_bound_proof', v)
    try:
        for off in (0, 1):
            hook(off)
            dc = DeviceChains(dens, x0, seed=11, step_size=2. if case == 'divergent' else 1.)
            s1, st1 = dc.run(45, sampler, n_warmup=30, **kw, layout='group')
            s2, st2 = dc.run(15, sampler, n_warmup=30, **kw, layout='group')
            out[off] = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)] + [dc.total_leapfrog]
    finally:
        hook(0)
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert np.array_equal(a, b, equal_nan=True)
    assert out[0][-1] == out[1][-1]


@pytest.mark.parametrize('case', ['outside', 'mixed', 'inside', 'bounded'])
def test_leaves_outside_the_bound_by_linearity_match_the_oracles_second_evaluation(ctx, case):
    """Outside the bound (modules/poly.py:480-503) the reference evaluates the surrogate a second time, at the projected point
    x_0; the sampler kernels get S x_0 from the S x they have by linearity and the sums of the reference's formulas as
    polynomials in alpha / beta (bfhip_oob.h) -- no second pass.  Against the oracle, which evaluates at x_0 as the reference
    does: with every leaf outside the bound (the regime of BASELINE configs 3 and 4), with chains crossing it, with none
    outside, and behind the constraint transform; the decay penalty on in all of them; on every chain layout."""
    from bayesfast_amd.workloads import correlated_gaussian_spec
    spec, _ = correlated_gaussian_spec(64, fit_scale={'outside': 0.3, 'mixed': 1.0, 'inside': 1.5, 'bounded': 0.8}[case])
    po = spec['poly']
    spec = dict(spec, use_decay=True, decay_mu=po['mu'] + 0.05, decay_hess=po['hess'], decay_alpha2=(1.5 * po['alpha'])**2, decay_gamma=0.1)
    if case == 'bounded':  # behind the constraint transform as well: all four kinds of bounds (density.py:92-140)
        lo = np.full(64, -9.) + np.arange(64) * 0.01
        spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * 16, dtype=np.uint8))
    x0 = np.random.default_rng(2).normal(size=(60, 64)) * (0.3 if case == 'bounded' else 1.)
    dev = _device_chains(ctx, spec, x0, 30, 20)
    orc_runs = _oracle_chains(spec, x0[:6], 12, 20)
    # (the first 12 iterations: in the 'mixed' regime the 1e-16 summation-order differences grow to 3e-4 by iteration 14 and reach
    # a tree decision by iteration ~17 on every layout -- with the reference's second pass as well as without it)
    _compare_nuts(dev, orc_runs, 12, n_head=6, tol_head=1e-8, rtol_q=1e-4)
    if case != 'bounded':   # (there the samples live in the transformed space)
        x = dev[0].reshape(-1, 64)
        beta = np.sqrt(np.einsum('ij,jk,ik->i', x - po['mu'], po['hess'], x - po['mu']))
        frac = float(np.mean(beta > po['alpha']))
        assert {'outside': frac > 0.99, 'mixed': 0.02 < frac < 0.98, 'inside': frac < 0.01}[case], frac


@pytest.mark.parametrize('kernel', ['pipe', 'pipe20', 'pipe10', 'pipebounded', 'pipedecay', 'sliced', 'sliced128', 'cubic24', 'cubic128'])
def test_chains_per_workgroup_never_change_results(ctx, kernel):
    """The wave-per-chain kernels with 16, 4 and 1 chains per workgroup (bfhip_sampler.hip: wave_layout_cpg; the waves without a
    chain only run matvec jobs): samples, statistics, adapted state and random streams are EQUAL, for a chain count that
    leaves every workgroup size a ragged last group.  With at most four chains in a workgroup the gradient tiles run on
    v_mfma_f64_4x4x4_4b instead of v_mfma_f64_16x16x4_f64: this is also the test that the two give the same bits.  The cubic
    cases: with chain-less waves in the workgroup the chains' waves take the cubic configs beside the matvec jobs (which the
    chain-less waves run two at a time), with 16 chains per workgroup they take them in phase C: the same numbers."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    if kernel.startswith('cubic'):
        d = int(kernel[5:])
        rng = np.random.default_rng(21)
        spec = _cubic_spec(d=d, seed=5, m2=np.sort(rng.choice(d, 20, replace=False)), m3=np.sort(rng.choice(d, 16, replace=False)),
                           amp=0.15 if d == 24 else 0.05)
    else:
        d = 128 if kernel == 'sliced128' else (int(kernel[4:]) if kernel[4:].isdigit() else 48) if kernel.startswith('pipe') else 48
        spec = correlated_gaussian_spec(d)[0]
        if kernel == 'pipedecay':   # the decay penalty (bf_nuts_pipe_kernel<W, false, true>), most leaves outside the bound
            po = spec['poly'] = dict(spec['poly'], alpha=0.4 * spec['poly']['alpha'])
            spec = dict(spec, use_decay=True, decay_mu=po['mu'] + 0.05, decay_hess=po['hess'], decay_alpha2=(1.5 * po['alpha'])**2, decay_gamma=0.1)
        if kernel == 'pipebounded':   # behind the constraint transform (bf_nuts_pipe_kernel<W, true>): all four kinds of bounds
            lo = np.full(d, -9.) + np.arange(d) * 0.01
            spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * (d // 4), dtype=np.uint8))
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(5).normal(size=(37, d)) * (0.5 if kernel.startswith('cubic') else (0.3 if kernel == 'pipebounded' else 1.))
    out = {}
    L = _lib.lib()
    try:
        _lib.debug_set('no_group', 1)
        _lib.debug_set('no_pipe', 0 if kernel.startswith('pipe') else 1)
        for cpg in (16, 8, 4, 1):   # (8: the pipelined kernel takes two 4 x 4 x 4 instructions per k-step)
            _lib.debug_set('wave_cpg', cpg)
            dc = DeviceChains(dens, x0, seed=4)
            s1, st1 = dc.run(24, 'NUTS', n_warmup=16, layout='wave')
            s2, st2 = dc.run(8, 'NUTS', n_warmup=16, layout='wave')
            out[cpg] = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)] + [dc.total_leapfrog]
    finally:
        _lib.debug_set('wave_cpg', 0)
        _lib.debug_set('no_pipe', 0)
        _lib.debug_set('no_group', 0)
    for cpg in (8, 4, 1):
        for a, b in zip(out[16][:-1], out[cpg][:-1]):
            assert np.array_equal(a, b, equal_nan=True), cpg
        assert out[16][-1] == out[cpg][-1]


@pytest.mark.parametrize('case', ['balanced', 'leaky_bound', 'depth_limit', 'divergent', 'd40', 'd32', 'd10', 'bounded', 'decay', 'decay_out', 'decay32', 'decay10'])
def test_pipelined_nuts_kernel_is_bit_identical_to_sliced_kernel(ctx, case):
    """bf_nuts_pipe_kernel (deferred bookkeeping, speculative next step, tree vectors in LDS) performs the same
    arithmetic per chain in the same order as bf_sampler_kernel: samples, statistics, adapted state and the random
    streams must agree bit for bit -- through warm-up (long and short trees, direction changes), with most evaluations
    outside the bound (S x_0 by linearity: the same helper in both kernels, bfhip_oob.h), at the depth limit (no speculation past the last doubling), with divergent
    first steps, and for a ragged dimension.  The leapfrog count excludes the dropped speculative evaluations."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    dec = case[:5] == 'decay'
    d = int(case[1:]) if case[0] == 'd' and case[1:].isdigit() else (int(case[5:]) if dec and case[5:].isdigit() else 64)
    spec, _ = correlated_gaussian_spec(d, fit_scale=0.3 if case == 'decay_out' else (1.0 if case == 'leaky_bound' else 1.5))
    if dec:  # the decay penalty (density.py:740-746), active for about half the points ('decay_out': every leaf outside the bound)
        po = spec['poly']
        spec = dict(spec, use_decay=True, decay_mu=po['mu'] + 0.05, decay_hess=po['hess'],
                    decay_alpha2=((1.5 if case == 'decay_out' else 0.8) * po['alpha'])**2, decay_gamma=0.1)
    if case == 'bounded':  # the same surrogate behind the constraint transform: all four kinds of bounds (density.py:92-140)
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * 16, dtype=np.uint8))
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(2).normal(size=(150, d)) * (3. if case == 'divergent' else (0.3 if case == 'bounded' else 1.))
    kw = {'depth_limit': dict(max_treedepth=2), 'divergent': dict(max_change=5.)}.get(case, {})
    out = {}
    try:
        _lib.debug_set('no_group', 1)  # (the default dispatch is the group kernel, bfhip_group.hip)
        for sliced in (0, 1):
            _lib.debug_set('no_pipe', sliced)
            dc = DeviceChains(dens, x0, seed=11, step_size=2. if case == 'divergent' else 1.)
            s1, st1 = dc.run(45, 'NUTS', n_warmup=30, **kw, layout='wave')
            s2, st2 = dc.run(15, 'NUTS', n_warmup=30, **kw, layout='wave')   # resume: the second launch starts from the stored state
            out[sliced] = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)] + [dc.total_leapfrog]
    finally:
        _lib.debug_set('no_pipe', 0)
        _lib.debug_set('no_group', 0)
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert np.array_equal(a, b, equal_nan=True)
    assert out[0][-1] == out[1][-1] == int(out[0][1][:, :, _lib.NSTATS.index('tree_size')].sum() + out[0][3][:, :, _lib.NSTATS.index('tree_size')].sum())
    ts = out[0][1][:, :, _lib.NSTATS.index('tree_size')]
    if case == 'depth_limit':
        assert ts.max() == 3 and (out[0][1][:, :, _lib.NSTATS.index('tree_depth')] == 2).any()
    if case == 'divergent':
        assert out[0][1][:, :, _lib.NSTATS.index('diverging')].sum() > 0
    if case == 'balanced':
        assert ts.max() >= 15 and ts.min() <= 3   # warm-up went through long and short trees


@pytest.mark.parametrize('case', ['balanced', 'leaky_bound', 'depth_limit', 'divergent', 'd40', 'd32', 'd10', 'bounded', 'bounded20', 'decay', 'decay_out', 'decay32', 'decay10'])
def test_lone_kernel_is_bit_identical_to_pipelined_kernel(ctx, case):
    """bf_lone_kernel (bfhip_lone.h: one chain per workgroup -- an integrator wave, a bookkeeper wave one leaf behind, W matvec
    waves on 4 x 4 x 4 tiles, three barriers per trip) performs the pipelined kernel's arithmetic per chain in the same order:
    samples, statistics, adapted state, random streams and the leapfrog count must agree bit for bit with bf_nuts_pipe_kernel
    at sixteen chains per workgroup -- through warm-up (long and short trees, direction changes), outside the bound, at the
    depth limit, with divergent first steps, ragged dimensions, behind the constraint transform and with the decay term; a
    second launch resumes from the stored state."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    import ctypes
    dec = case[:5] == 'decay'
    d = int(case[1:]) if case[0] == 'd' and case[1:].isdigit() else (int(case[5:]) if dec and case[5:].isdigit() else (20 if case == 'bounded20' else 64))
    spec, _ = correlated_gaussian_spec(d, fit_scale=0.3 if case == 'decay_out' else (1.0 if case == 'leaky_bound' else 1.5))
    if dec:
        po = spec['poly']
        spec = dict(spec, use_decay=True, decay_mu=po['mu'] + 0.05, decay_hess=po['hess'],
                    decay_alpha2=((1.5 if case == 'decay_out' else 0.8) * po['alpha'])**2, decay_gamma=0.1)
    if case[:7] == 'bounded':
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * (d // 4), dtype=np.uint8))
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(2).normal(size=(70, d)) * (3. if case == 'divergent' else (0.3 if case[:7] == 'bounded' else 1.))
    kw = {'depth_limit': dict(max_treedepth=2), 'divergent': dict(max_change=5.)}.get(case, {})
    out = {}
    L = _lib.lib()
    try:
        _lib.debug_set('no_group', 1)
        for lone in (0, 2):
            _lib.debug_set('lone', lone)
            _lib.debug_set('wave_cpg', 16 if lone == 0 else 0)
            dc = DeviceChains(dens, x0, seed=11, step_size=2. if case == 'divergent' else 1.)
            s1, st1 = dc.run(45, 'NUTS', n_warmup=30, **kw, layout='wave')
            s2, st2 = dc.run(15, 'NUTS', n_warmup=30, **kw, layout='wave')
            out[lone] = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)] + [dc.total_leapfrog]
            assert ('bf_lone_kernel' in _lib.last_kernel()) == (lone == 2)
    finally:
        _lib.debug_set('lone', 1)
        _lib.debug_set('wave_cpg', 0)
        _lib.debug_set('no_group', 0)
    names = ['samples', 'stats', 'samples2', 'stats2', 'sc', 'vec', 'rng']
    for nm, a, b in zip(names, out[0][:-1], out[2][:-1]):
        assert np.array_equal(a, b, equal_nan=True), (nm, np.argwhere(~((a == b) | ((a != a) & (b != b))))[:5])
    assert out[0][-1] == out[2][-1]
    ts = out[0][1][:, :, _lib.NSTATS.index('tree_size')]
    if case == 'balanced':
        assert ts.max() >= 15 and ts.min() <= 3
    if case in ('balanced', 'decay', 'd32', 'bounded'):
        # the kernel's other forms (three job waves at d > 32; the tight-register instantiation of a crowded tail): the same numbers
        try:
            _lib.debug_set('no_group', 1)
            _lib.debug_set('lone', 2)
            for form in (1, 2):
                _lib.debug_set('lone_form', form)
                dc = DeviceChains(dens, x0, seed=11, step_size=1.)
                s1, st1 = dc.run(45, 'NUTS', n_warmup=30, **kw, layout='wave')
                s2, st2 = dc.run(15, 'NUTS', n_warmup=30, **kw, layout='wave')
                got = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)]
                for nm, a, b in zip(names, out[2][:-1], got):
                    assert np.array_equal(a, b, equal_nan=True), (form, nm)
        finally:
            _lib.debug_set('lone_form', -1)
            _lib.debug_set('lone', 1)
            _lib.debug_set('no_group', 0)


def test_launch_cuts_do_not_change_results(ctx):
    """DeviceChains.run queues launches of launch_iters iterations; the cut must not show in any output."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    spec, _ = correlated_gaussian_spec(64)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(3).normal(size=(40, 64))
    out = []
    for li in (None, 7, 250, [3, 11], 'auto'):   # (a sequence: its last length repeats; 'auto': 100 while adapting, then 250)
        dc = DeviceChains(dens, x0, seed=3)
        s, st = dc.run(50, 'NUTS', n_warmup=30, launch_iters=li, layout=_LAYOUT['v'])
        out.append((s.cpu().numpy(), st.cpu().numpy(), dc.sc.cpu().numpy(), dc.total_leapfrog))
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1], equal_nan=True)
        assert np.array_equal(o[2], out[0][2], equal_nan=True) and o[3] == out[0][3]
    with pytest.raises(ValueError):
        DeviceChains(dens, x0, seed=3).run(5, 'NUTS', n_warmup=3, launch_iters='sometimes')


def test_nuts_divergences_and_max_treedepth(ctx, samp):
    """Huge step size => divergent leaves and immediate U-turns; tiny max_treedepth => depth cap."""
    spec = _spec(samp, 'div5.')
    x0 = np.repeat(samp['div5.x0'], 4, 0) + np.arange(4)[:, None] * 0.1
    kw = dict(step_size=40., max_change=50.)
    dev = _device_chains(ctx, spec, x0, 30, 10, **kw)
    orc_runs = _oracle_chains(spec, x0, 30, 10, **kw)
    assert sum(r[1]['diverging'].sum() for r in orc_runs) >= 1
    _compare_nuts(dev, orc_runs, 30)
    spec = _spec(samp, 'plain16.')
    x0 = np.random.default_rng(5).normal(size=(4, 16))
    kw = dict(step_size=0.05, max_treedepth=3)
    dev = _device_chains(ctx, spec, x0, 12, 0, **kw)
    orc_runs = _oracle_chains(spec, x0, 12, 0, **kw)
    assert dev[1]['tree_depth'].max() == 3 and dev[1]['tree_size'].max() == 7
    _compare_nuts(dev, orc_runs, 12)


def test_nuts_far_start_extrapolation_branch(ctx, samp):
    """Start 6 sigma out: the chains begin outside the surrogate's alpha-ellipsoid (poly.py:480-503) and
    the first trees see energy changes of hundreds (weights over a huge dynamic range)."""
    spec = _spec(samp, 'd64.')
    rng = np.random.default_rng(6)
    x0 = rng.normal(size=(6, 64)) * 6.
    from oracle import oracle as orc
    mu, H = spec['poly']['mu'], spec['poly']['hess']
    assert (np.sqrt(np.einsum('ij,jk,ik->i', x0 - mu, H, x0 - mu)) > spec['poly']['alpha']).all()
    dev = _device_chains(ctx, spec, x0, 12, 8)
    orc_runs = _oracle_chains(spec, x0, 12, 8)
    # positions of order 20 and energy changes of hundreds: the head tolerance is 1e-8 here (summation-order
    # differences of the wave reductions grow fastest on these trajectories); discrete fields stay exact
    _compare_nuts(dev, orc_runs, 12, tol_head=1e-8)


@pytest.mark.parametrize('name', ['plain16', 'full5'])
def test_hmc_trajectories_match_oracle(ctx, samp, name):
    spec = _spec(samp, name + '.')
    rng = np.random.default_rng(7)
    x0 = rng.normal(size=(6, spec['d'])) * 0.5
    s, st, dc = _device_chains(ctx, spec, x0, 30, 20, sampler='HMC', n_int_step=8)
    orc_runs = _oracle_chains(spec, x0, 30, 20, sampler='HMC', n_int_step=8)
    for i, (so, sto, ch) in enumerate(orc_runs):
        for f in ('accepted', 'diverging', 'n_int_step'):
            assert np.array_equal(st[f][i], sto[f]), (i, f)
        np.testing.assert_allclose(s[i][:8], so[:8], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(s[i], so, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(st['accept_stat'][i], sto['accept_stat'], rtol=1e-3, atol=1e-3)
    assert dc.total_leapfrog == 6 * 30 * 8


def test_bad_initial_energy_raises(ctx, samp):
    """Non-finite initial energy is an error, not a divergence (base_hmc.py:72-76)."""
    spec = _spec(samp, 'plain16.')
    x0 = np.zeros((3, 16))
    x0[1, 0] = np.inf
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=1)
    with pytest.raises(RuntimeError):
        dc.run(3, n_warmup=1)


def test_nuts_posterior_moments_quadratic_target(ctx):
    """T2: an exactly quadratic log density; 2048 chains x (150 warm-up + 100) draws vs N(0, P^-1)."""
    d = 32
    rng = np.random.default_rng(123)
    L = np.eye(d) + 0.3 * np.tril(rng.normal(size=(d, d)), -1) / np.sqrt(d)
    P = L @ L.T
    cov = np.linalg.inv(P)
    quad = np.zeros((d, d))
    iu = np.triu_indices(d)
    A = -0.5 * P
    quad[iu] = A[iu] * np.where(iu[0] == iu[1], 1., 2.)
    spec = dict(d=d, ranges=None, hard_bounds=None, su_lo=None, su_diff=None, use_decay=False,
                poly=dict(input_size=d, output_size=1, use_bound=False, configs=[
                    dict(order='linear', input_mask=np.arange(d), output_mask=[0], coef=np.zeros((1, d + 1))),
                    dict(order='quadratic', input_mask=np.arange(d), output_mask=[0], coef=quad[None])]))
    n_chain = 2048
    x0 = rng.normal(size=(n_chain, d))
    s, st, dc = _device_chains(ctx, spec, x0, 250, 150)
    draws = s[:, 150:].reshape(-1, d)
    assert st['diverging'][:, 150:].sum() == 0
    acc = st['mean_tree_accept'][:, 150:].mean()
    assert 0.7 < acc < 0.92, acc
    sd = np.sqrt(np.diag(cov))
    n_eff = n_chain * 100 / 4.  # conservative
    assert np.all(np.abs(draws.mean(0)) < 5 * sd / np.sqrt(n_eff))
    np.testing.assert_allclose(draws.var(0), np.diag(cov), rtol=0.05)
    emp = np.cov(draws, rowvar=False)
    assert np.max(np.abs(emp - cov) / np.sqrt(np.outer(np.diag(cov), np.diag(cov)))) < 0.03


@pytest.mark.parametrize('d', [64, 40, 32, 20, 16, 7])
def test_split_layout_is_bit_identical_to_group_layout_on_the_device(ctx, d):
    """chain_layout 3 (bfhip_split.h: integrator and bookkeeper waves, the bookkeepers one leaf behind; four + four waves at d > 32,
    two + two at d > 16, one + one below) against chain_layout 1 on the GPU: equal samples, statistics, chain state and random
    streams -- ragged groups through warm-up, starts far outside the bound (second passes, the third barrier), launch cuts,
    padded dimensions.  (The CPU emulation checks the same and more: tests/test_group_emu.py.)"""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    import ctypes as C
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    rng = np.random.default_rng(3)
    kname = _lib.last_kernel
    for x0, n, nw, kw in ((rng.normal(size=(37, d)), 60, 40, {}), (rng.normal(size=(20, d)) * 6., 30, 20, {}),
                          (rng.normal(size=(300, d)), 90, 50, dict(launch_iters=17))):
        out = []
        for layout in ('group', 'split'):
            ch = DeviceChains(dens, x0, seed=5)
            s, st = ch.run(n, 'NUTS', n_warmup=nw, layout=layout, **kw)
            out.append([t.cpu().numpy() for t in (s, st, ch.sc, ch.vec, ch.rng)] + [ch.total_leapfrog])
            assert kname().startswith('bf_split_kernel' if layout == 'split' else 'bf_group_kernel')
        for u, v in zip(*out):
            assert np.array_equal(u, v, equal_nan=True) if isinstance(u, np.ndarray) else u == v


def test_config2_fitted_bound_on_surrogate_posterior_moments(ctx):
    """T2 for BASELINE config 2 as it is specified: d = 32, 1024 chains, ``PolyModel('quadratic')`` FITTED on
    2 P Sobol-normal points of the target (P = 561) with its extrapolation bound ON, NUTS defaults (1500 iterations, 500
    warm-up, diagonal adaptation), through the public entry point; against the analytic posterior N(0, P^-1): means within
    4 sigma / sqrt(ESS), variance ratios within 4 Monte-Carlo standard errors, no draw outside the bound, no divergence.
    (The fit points are spread 1.5 x wider than the posterior, as in a recipe's first rounds: DESIGN.md section 5.)"""
    from bayesfast_amd import PolyModel, SurrogateDensity, sample
    from bayesfast_amd.workloads import sobol_normal
    d, C = 32, 1024
    rng = np.random.default_rng(123)
    L = np.eye(d) + 0.3 * np.tril(rng.normal(size=(d, d)), -1) / np.sqrt(d)
    P = L @ L.T
    cov = np.linalg.inv(P)
    su = PolyModel('quadratic', input_size=d, output_size=1)
    assert su.n_param == 561 and su.bound_options.use_bound
    den = SurrogateDensity(su)
    x_fit = 1.5 * sobol_normal(2 * su.n_param, d, seed=1) @ np.linalg.cholesky(cov).T
    den.fit(x_fit, -0.5 * np.einsum('ij,jk,ik->i', x_fit, P, x_fit))
    tt = sample(den, dict(n_chain=C, n_iter=1500, n_warmup=500, random_generator=7), verbose=False)
    draws = tt.get()                                   # (C * 1000, d), post-warm-up
    assert tt.stat('diverging')[:, 500:].sum() == 0
    xm = draws - su._mu
    assert np.sqrt(np.einsum('ij,jk,ik->i', xm, su._hess, xm)).max() < su._alpha     # every draw inside the bound ellipsoid
    # effective sample size per dimension from the chains' own autocorrelation (batch means over chains of 1000)
    per_chain = tt.get(flatten=False)                  # (C, 1000, d)
    sd = np.sqrt(np.diag(cov))
    ess = C * 1000 * np.clip(per_chain.var(1, ddof=1).mean(0) / (1000 * per_chain.mean(1).var(0, ddof=1)), 0.02, 1.)
    assert np.all(np.abs(draws.mean(0)) < 4 * sd / np.sqrt(ess)), (draws.mean(0) / (sd / np.sqrt(ess)))
    ratio = draws.var(0) / np.diag(cov)
    assert np.all(np.abs(ratio - 1.) < 4 * np.sqrt(2. / ess)), ratio
    emp = np.cov(draws[::5], rowvar=False)
    assert np.max(np.abs(emp - cov) / np.outer(sd, sd)) < 0.03


def test_headline_size_properties(ctx):
    """BASELINE's headline size (4096 chains x 64-d, bound on) through properties that do not need the oracle: posterior
    moments against the analytic N(0, Sigma), no sample outside the bound ellipsoid of a well-posed surrogate, the leapfrog
    counter equal to the sum of the tree sizes, shard invariance (two halves with their global stream indices == the
    full run) and invariance to the launch length."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    d, C, n_it, n_w = 64, 4096, 420, 220
    spec, cov = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(5).normal(size=(C, d))
    dc = DeviceChains(dens, x0, seed=17)
    s, st = dc.run(n_it, 'NUTS', n_warmup=n_w, layout=_LAYOUT['v'])
    ts = st[:, :, _lib.NSTATS.index('tree_size')]
    assert dc.total_leapfrog == int(ts.sum().item())
    assert float(st[:, n_w:, _lib.NSTATS.index('diverging')].sum().item()) == 0.
    x = s[:, n_w:].reshape(-1, d).cpu().numpy()
    sd = np.sqrt(np.diag(cov))
    assert np.abs(x.mean(0) / sd).max() < 0.01          # 820 k draws, autocorrelated: se of the mean ~ 0.002 sd
    np.testing.assert_allclose(x.var(0), np.diag(cov), rtol=0.02)
    emp = np.cov(x[::3], rowvar=False)
    assert np.max(np.abs(emp - cov) / np.outer(sd, sd)) < 0.02
    pm = spec['poly']
    xm = x[::7] - np.asarray(pm['mu'])
    assert np.einsum('ni,ij,nj->n', xm, np.asarray(pm['hess']), xm).max() < float(pm['alpha'])**2   # never outside the bound
    # shards and launch cuts
    dc_a = DeviceChains(dens, x0[:C // 2], seed=17, first_stream=0)
    dc_b = DeviceChains(dens, x0[C // 2:], seed=17, first_stream=C // 2)
    sa, _ = dc_a.run(60, 'NUTS', n_warmup=n_w, launch_iters=25, layout=_LAYOUT['v'])
    sb, _ = dc_b.run(60, 'NUTS', n_warmup=n_w, launch_iters=None, layout=_LAYOUT['v'])
    assert np.array_equal(sa.cpu().numpy(), s[:C // 2, :60].cpu().numpy())
    assert np.array_equal(sb.cpu().numpy(), s[C // 2:, :60].cpu().numpy())


def _cubic_spec(d=6, seed=3, m2=(1, 2, 4), m3=(0, 1, 3, 5), amp=1.):
    """Negative-definite quadratic + small masked cubic-2 / cubic-3 terms, bound on."""
    rng = np.random.default_rng(seed)
    L = np.eye(d) + 0.3 * np.tril(rng.normal(size=(d, d)), -1) / np.sqrt(d)
    A = -0.5 * (L @ L.T)
    quad = np.zeros((d, d))
    iu = np.triu_indices(d)
    quad[iu] = A[iu] * np.where(iu[0] == iu[1], 1., 2.)
    m2, m3 = np.array(m2), np.array(m3)
    n2, n3 = m2.size, m3.size
    c2 = 0.02 * amp * rng.normal(size=(1, n2, n2))
    c3 = np.zeros((1, n3, n3, n3))
    j, k, l = np.meshgrid(*[np.arange(n3)] * 3, indexing='ij')
    c3[0][(j < k) & (k < l)] = 0.03 * amp * rng.normal(size=int(((j < k) & (k < l)).sum()))
    x = rng.normal(size=(200, d))
    mu = x.mean(0)
    hess = np.linalg.inv(np.cov(x, rowvar=False))
    alpha = float(np.sqrt(np.einsum('ij,jk,ik->i', x - mu, hess, x - mu)).max())
    poly = dict(input_size=d, output_size=1, use_bound=True, mu=mu, hess=hess, alpha=alpha, f_mu=np.array([0.3]),
                configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.array([0]), coef=0.1 * rng.normal(size=(1, d + 1))),
                         dict(order='quadratic', input_mask=np.arange(d), output_mask=np.array([0]), coef=quad[None]),
                         dict(order='cubic-2', input_mask=m2, output_mask=np.array([0]), coef=c2),
                         dict(order='cubic-3', input_mask=m3, output_mask=np.array([0]), coef=c3)])
    return dict(d=d, ranges=None, hard_bounds=None, su_lo=None, su_diff=None, poly=poly, use_decay=False)


def test_cubic_configs_eval_and_nuts(ctx):
    """cubic-2 / cubic-3 PolyConfigs with masks (modules/_poly.pyx:49-137): batched logp+grad inside and
    outside the bound, then NUTS trajectories, vs the oracle."""
    from bayesfast_amd.device import DeviceDensity
    from oracle import oracle as orc
    spec = _cubic_spec()
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.normal(size=(40, 6)) * 0.7, rng.normal(size=(9, 6)) * 3.])
    lp0, g0 = orc.logp_and_grad(spec, x)
    lp, g = DeviceDensity(spec, ctx).logp_and_grad(x)
    np.testing.assert_allclose(lp.cpu().numpy(), lp0, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(g.cpu().numpy(), g0, rtol=1e-10, atol=1e-10)
    x0 = rng.normal(size=(5, 6)) * 0.5
    dev = _device_chains(ctx, spec, x0, 25, 15)
    orc_runs = _oracle_chains(spec, x0, 25, 15)
    _compare_nuts(dev, orc_runs, 25)


def test_cubic_configs_with_ragged_multi_chunk_masks(ctx):
    """Cubic masks that are neither one 16-wide tile nor a multiple of it (20 cubic-2 inputs, 18 cubic-3 inputs of 24: two
    chunks of the LDS-resident cubic-3 table, guarded tails in both contractions of bf_sampler_kernel) vs the oracle."""
    from bayesfast_amd.device import DeviceDensity
    from oracle import oracle as orc
    d = 24
    rng = np.random.default_rng(21)
    m2 = np.sort(rng.choice(d, 20, replace=False))
    m3 = np.sort(rng.choice(d, 18, replace=False))
    spec = _cubic_spec(d=d, seed=5, m2=m2, m3=m3, amp=0.15)
    x = np.concatenate([rng.normal(size=(40, d)) * 0.7, rng.normal(size=(9, d)) * 3.])
    lp0, g0 = orc.logp_and_grad(spec, x)
    lp, g = DeviceDensity(spec, ctx).logp_and_grad(x)
    np.testing.assert_allclose(lp.cpu().numpy(), lp0, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(g.cpu().numpy(), g0, rtol=1e-10, atol=1e-10)
    x0 = rng.normal(size=(5, d)) * 0.5
    dev = _device_chains(ctx, spec, x0, 25, 15)
    orc_runs = _oracle_chains(spec, x0, 25, 15)
    _compare_nuts(dev, orc_runs, 25)


# ---- full-rank metric: QuadMetricFull / QuadMetricFullAdapt (samplers/hmc_utils/metrics.py:94-132,240-330) -----

def _full_metric_compare(ctx, spec, x0, n_iter, n_warmup, sampler, chain_kw, n_head=6, tol=2e-3, n_exact=None):
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd import _lib
    from oracle import oracle as orc
    dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=SEED, **{k: v for k, v in chain_kw.items() if k != 'adapt_metric'})
    s, st = dc.run(n_iter, sampler, n_warmup=n_warmup, n_int_step=8,
                   **{k: v for k, v in chain_kw.items() if k in ('adapt_metric',)})
    s, st = s.cpu().numpy(), st.cpu().numpy()
    cov = dc.covariance().cpu().numpy()
    names = _lib.NSTATS if sampler == 'NUTS' else _lib.HSTATS
    for i in range(x0.shape[0]):
        ch = orc.Chain(x0[i], **chain_kw)
        rng = orc.make_rng('xoshiro', seed=SEED, stream=i)
        if sampler == 'NUTS':
            so, sto = orc.nuts_run(spec, ch, rng, n_iter, n_warmup)
            for f in ('tree_depth', 'tree_size', 'diverging'):
                assert np.array_equal(st[i, :n_exact, names.index(f)], sto[f][:n_exact]), (i, f)
        else:
            so, sto = orc.hmc_run(spec, ch, rng, n_iter, n_warmup, n_int_step=8)
            assert np.array_equal(st[i, :, names.index('accepted')], sto['accepted']), i
        np.testing.assert_allclose(s[i, :n_head], so[:n_head], rtol=1e-9, atol=1e-9)
        if np.isfinite(tol):
            np.testing.assert_allclose(s[i], so, rtol=tol, atol=tol)
            np.testing.assert_allclose(cov[i], ch.mat('cov'), rtol=tol, atol=tol)


def test_full_metric_adaptive_and_fixed_match_oracle(ctx):
    """Device chains with the full-rank metric against the oracle (itself pinned to the reference's trajectories by
    tests/golden/sampler_fullmetric.npz): adaptive from the identity (window switch and doubling inside the
    warm-up), adaptive from a given covariance, and fixed; NUTS and HMC."""
    from specio import rebuild_spec
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'sampler_fullmetric.npz'))
    spec = rebuild_spec(z, 'fm8.')
    cov0 = z['fm8.cov0']
    x0 = np.random.default_rng(4).normal(size=(5, 8)) * 0.5
    _full_metric_compare(ctx, spec, x0, 40, 28, 'NUTS', dict(metric='full', adapt_window=8))
    _full_metric_compare(ctx, spec, x0, 30, 20, 'NUTS', dict(metric=cov0, adapt_metric=False))
    _full_metric_compare(ctx, spec, x0, 30, 20, 'HMC', dict(metric=cov0, adapt_window=8), tol=1e-2)


def test_full_metric_64d_and_bad_covariance(ctx, samp):
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    spec = _spec(samp, 'd64.')
    x0 = np.random.default_rng(6).normal(size=(18, 64)) * 0.5
    # 64-d trees are hundreds of leapfrogs long while the covariance estimate is still rough: the discrete fields
    # are compared on the first iterations, the positions on the head (rounding differences amplify after that)
    _full_metric_compare(ctx, spec, x0[:3], 14, 10, 'NUTS', dict(metric='full'), n_head=5, tol=np.inf, n_exact=8)
    with pytest.raises(ValueError):
        DeviceChains(DeviceDensity(spec, ctx), x0, metric=-np.eye(64))


@pytest.mark.parametrize('d', [1, 3, 17, 33, 65, 100])
def test_odd_shapes_match_oracle(ctx, d):
    """Dimensions that are not multiples of the 16-wide tiles, and chain counts that leave ragged workgroups (1, 17):
    NUTS and HMC chains against the oracle (tree sizes / accept flags exactly, head positions to 1e-8)."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    from oracle import oracle as orc
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    for C in (1, 17):
        x0 = np.random.default_rng(d * 1000 + C).normal(size=(C, d)) * 0.7
        for smp in ('NUTS', 'HMC'):
            dc = DeviceChains(dens, x0, seed=9)
            s, st = dc.run(12, smp, n_warmup=8, n_int_step=6, layout=_LAYOUT['v'])
            s, st = s.cpu().numpy(), st.cpu().numpy()
            for i in sorted(set((0, C - 1))):
                ch = orc.Chain(x0[i])
                rng = orc.make_rng('xoshiro', seed=9, stream=i)
                if smp == 'NUTS':
                    so, sto = orc.nuts_run(spec, ch, rng, 12, 8)
                    assert np.array_equal(st[i, :, _lib.NSTATS.index('tree_size')], sto['tree_size']), (d, C, i)
                else:
                    so, sto = orc.hmc_run(spec, ch, rng, 12, 8, n_int_step=6)
                    assert np.array_equal(st[i, :, _lib.HSTATS.index('accepted')], sto['accepted']), (d, C, i)
                np.testing.assert_allclose(s[i, :5], so[:5], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('d', [5, 16, 30, 48])
def test_bounded_parameters_small_dims_match_oracle(ctx, d):
    """NUTS behind the constraint transform at d <= 64 runs on the pipelined kernel's transform instantiation for every
    padded dimension: chains with all four kinds of bounds against the oracle (ragged workgroup of 17 chains)."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    from oracle import oracle as orc
    spec, _ = correlated_gaussian_spec(d)
    lo = np.full(d, -9.) + np.arange(d) * 0.01
    hb = np.array(([[1, 1], [1, 0], [0, 1], [0, 0]] * d)[:d], dtype=np.uint8)
    spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=hb)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(d).normal(size=(17, d)) * 0.3
    dc = DeviceChains(dens, x0, seed=4)
    s, st = dc.run(14, 'NUTS', n_warmup=9, layout=_LAYOUT['v'])
    s, st = s.cpu().numpy(), st.cpu().numpy()
    for i in (0, 8, 16):
        so, sto = orc.nuts_run(spec, orc.Chain(x0[i]), orc.make_rng('xoshiro', seed=4, stream=i), 14, 9)
        assert np.array_equal(st[i, :, _lib.NSTATS.index('tree_size')], sto['tree_size']), (d, i)
        np.testing.assert_allclose(s[i, :6], so[:6], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(st[i, :6, _lib.NSTATS.index('logp')], sto['logp'][:6], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('decay,bounds', [(True, False), (False, True), (True, True)])
def test_compile_time_feature_sets_64d_match_oracle(ctx, samp, decay, bounds):
    """The 64-d instantiations with the decay penalty and / or the constraint transform fixed at compile time
    (sampler template parameter FS = 3, 5, 7) against the oracle."""
    spec = dict(_spec(samp, 'd64.'))
    d = 64
    if decay:
        spec.update(use_decay=True, decay_mu=spec['poly']['mu'], decay_hess=spec['poly']['hess'],
                    decay_alpha2=float(spec['poly']['alpha'])**2 * 0.6, decay_gamma=0.1)
    if bounds:
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        spec.update(ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * 16, dtype=np.uint8))
    x0 = np.random.default_rng(8).normal(size=(5, d)) * 0.3
    dev = _device_chains(ctx, spec, x0, 14, 9)
    orc_runs = _oracle_chains(spec, x0, 14, 9)
    _compare_nuts(dev, orc_runs, 14, n_head=6, tol_head=1e-8)


@pytest.mark.parametrize('order', ['linear', 'quadratic'])
def test_config1_donut_shapes_2d_decay_no_bound(ctx, order):
    """BASELINE config 1 (examples/2d-donut.ipynb): d = 2, 4 chains, the surrogate of the radius module is 'linear'
    (OptimizeStep) and then 'quadratic' with use_bound=False, and the density carries the decay penalty
    (use_decay=True, core/density.py:740-746).  Here the same kernel instantiation -- d = 2, no bound, decay on -- with
    a surrogate log-density of that shape, NUTS and HMC against the oracle."""
    from oracle import oracle as orc
    d = 2
    rng = np.random.default_rng(12)
    xf = rng.normal(size=(40, d)) * 1.5 + np.array([0.3, -0.2])
    cfgs = [dict(order='linear', input_mask=np.arange(d), output_mask=np.array([0]), coef=np.array([[-1.2, 0.4, -0.3]]))]
    if order == 'quadratic':
        A = np.zeros((1, d, d))
        A[0, 0, 0], A[0, 0, 1], A[0, 1, 1] = -0.45, 0.12, -0.3   # only j <= k is read (modules/_poly.pyx:13-28)
        cfgs.append(dict(order='quadratic', input_mask=np.arange(d), output_mask=np.array([0]), coef=A))
    mu = xf.mean(0)
    hess = np.linalg.inv(np.cov(xf, rowvar=False))
    beta = np.einsum('ij,jk,ik->i', xf - mu, hess, xf - mu)**0.5
    alpha = float(np.max(beta) * 1.5)                              # alpha_p = 150 (density.py:761-762,803-810)
    spec = dict(d=d, ranges=None, hard_bounds=None, su_lo=None, su_diff=None,
                poly=dict(input_size=d, output_size=1, configs=cfgs, use_bound=False),
                use_decay=True, decay_mu=mu, decay_hess=hess, decay_alpha2=alpha**2, decay_gamma=0.1)
    x0 = rng.normal(size=(4, d))
    dev = _device_chains(ctx, spec, x0, 40, 25)
    orc_runs = _oracle_chains(spec, x0, 40, 25)
    _compare_nuts(dev, orc_runs, 40)
    s, st, dc = _device_chains(ctx, spec, x0, 20, 12, sampler='HMC', n_int_step=6)
    for i, (so, sto, ch) in enumerate(_oracle_chains(spec, x0, 20, 12, sampler='HMC', n_int_step=6)):
        assert np.array_equal(st['accepted'][i], sto['accepted'])
        np.testing.assert_allclose(s[i][:8], so[:8], rtol=1e-9, atol=1e-9)


def test_tree_size_mode_share_flag_matches_numpy(ctx):
    """bfhip_tree_size_mode_share (the layout choice's helper): the most common tree_size of the given rows when its share
    >= threshold, else 0, for uniform, mixed and out-of-range sizes; the work buffer is left clean between calls."""
    import torch
    from bayesfast_amd import _lib
    from bayesfast_amd.device import _ptr
    rng = np.random.default_rng(8)
    n_chain, n_out = 300, 40
    work = torch.zeros(_lib.TREE_MODE_WORK, dtype=torch.int32, device=ctx.device)
    ts_col = _lib.NSTATS.index('tree_size')
    seen = set()
    for p_mode, row0, n_rows, share, slow in ((1.0, 0, 40, 0.98, 0), (0.99, 8, 32, 0.98, 0), (0.9, 8, 32, 0.98, 0), (0.9, 0, 7, 0.85, 0),
                                              (0.5, 39, 1, 0.5, 0), (1.0, 8, 32, 0.98, 63), (1.0, 8, 32, 0.98, 31), (1.0, 8, 32, 0.98, 15),
                                              (0.5, 8, 32, 0.98, 4095)):
        st = np.zeros((n_chain, n_out, _lib.STAT_STRIDE))
        sizes = np.where(rng.uniform(size=(n_chain, n_out)) < p_mode, 7, rng.choice([1, 3, 15, 31, 5000], size=(n_chain, n_out)))
        if slow:
            # (one chain whose trees are larger than everybody else's: 63 leaves make it a laggard -- nine times the mean --; 31 -- 4.4
            # times the mean, but under four times it by its size class's lower edge -- and 15 do not; round 6: the threshold is
            # four times the mean, and the bit is reported whether or not the trees are in step -- the last case)
            sizes[123] = slow
        st[:, :, ts_col] = sizes
        blk = np.minimum(sizes[:, row0:row0 + n_rows], 4095).reshape(-1)
        cnt = np.bincount(blk)
        mode = max(1, int(cnt.argmax()))
        sums = np.minimum(sizes[:, row0:row0 + n_rows], 4095).sum(1)   # per chain; the busiest one's size class against four times the mean
        top = int(np.clip(np.searchsorted(np.array(_lib.LAG_EDGES), sums, side='right') - 1, 0, 63).max())
        lag = _lib.LAG_EDGES[top] * n_chain >= 4 * int(blk.sum())
        want = (mode if cnt.max() >= share * blk.size else 0) + (4096 if lag else 0)   # (size 0: not in step)
        t = ctx.tensor(st)
        _lib.check(ctx._lib.bfhip_tree_size_mode_share(ctx.handle, n_chain, n_out, _ptr(t), row0, n_rows, float(share), _ptr(work)))
        torch.cuda.synchronize()
        w = work.cpu().numpy()
        assert int(w[0]) == want, (p_mode, row0, n_rows, slow, int(w[0]), want)
        assert not w[1:].any()  # histograms and arrival counter cleared for the next call
        seen.add(((want & 4095) > 0, want >= 4096))
    assert seen >= {(True, False), (True, True), (False, True)}   # (in step without and with a laggard; a laggard among trees that differ)


@pytest.mark.parametrize('case', ['inside', 'leaky'])
def test_sliced_kernel_bound_proof_at_d128_never_changes_results(ctx, case):
    """The same proof in bf_sampler_kernel at d = 128 (two tile jobs per wave: the H (x - mu) job is skipped while every
    evaluating chain of the 8-chain group is provably inside the bound): bit-identical with the proof switched off."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    d = 128
    spec, _ = correlated_gaussian_spec(d, fit_scale=1.0 if case == 'leaky' else 1.5)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(2).normal(size=(44, d))
    hook = lambda v: _lib.debug_set('no_bound_proof', v)
    out = {}
    try:
        for off in (0, 1):
            hook(off)
            dc = DeviceChains(dens, x0, seed=11)
            s1, st1 = dc.run(30, 'NUTS', n_warmup=20)
            s2, st2 = dc.run(10, 'NUTS', n_warmup=20)
            out[off] = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)] + [dc.total_leapfrog]
    finally:
        hook(0)
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert np.array_equal(a, b, equal_nan=True)
    assert out[0][-1] == out[1][-1]


def test_launches_may_alternate_between_the_layouts(ctx):
    """chain_layout = auto switches layouts between launches: the chain state (positions, adaptation, streams) written by
    one layout's kernel is what the other continues from.  Four launches group / wave / group / wave through the warm-up
    and beyond follow the oracle like a single-layout run: tree sizes and divergences exactly, positions to rounding."""
    from oracle import oracle as orc
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    import torch
    d = 24
    spec, _ = correlated_gaussian_spec(d)
    x0 = np.random.default_rng(9).normal(size=(37, d))
    dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=21)
    parts = []
    for n, lay in ((9, 'group'), (8, 'wave'), (7, 'group'), (6, 'wave')):
        parts.append(dc.run(n, 'NUTS', n_warmup=20, layout=lay))
        assert dc.last_layout == lay
    s = torch.cat([p[0] for p in parts], 1).cpu().numpy()
    st = torch.cat([p[1] for p in parts], 1).cpu().numpy()
    for i in (0, 15, 16, 36):
        so, sto = orc.nuts_run(spec, orc.Chain(x0[i]), orc.make_rng('xoshiro', seed=21, stream=i), 30, 20)
        assert np.array_equal(st[i, :, _lib.NSTATS.index('tree_size')], sto['tree_size']), i
        assert np.array_equal(st[i, :, _lib.NSTATS.index('diverging')], sto['diverging']), i
        np.testing.assert_allclose(s[i, :10], so[:10], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(s[i], so, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('n_chain', [37, 600])
def test_config5_four_wave_form_is_bit_identical_to_the_eight_wave_form(ctx, n_chain):
    """bf_sampler_kernel<8, ..., 17> (config 5's shard: four waves of 512 registers, each with a chain and two row tiles of S in
    registers for the whole launch) against <8, ..., 16> (eight waves, S streamed from L2 in every trip): samples, statistics,
    adapted state, random streams and leapfrog counts EQUAL -- with one chain per workgroup (37 chains) and with four (600)."""
    if _LAYOUT['v'] != 'group':
        pytest.skip('does not depend on the module-wide layout parameter')
    import bayesfast_amd as bfa
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import planck_like_logp
    from bayesfast_amd import _lib
    d = 128
    rng = np.random.default_rng(2024)
    logp, chol = planck_like_logp(d)
    m16 = np.arange(16)
    su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic'), bfa.PolyConfig('cubic-2', input_mask=m16),
                        bfa.PolyConfig('cubic-3', input_mask=m16)], input_size=d, output_size=1)
    den = bfa.SurrogateDensity(su)
    x_fit = rng.normal(size=(su.n_param + 800, d)) @ chol.T
    den.fit(x_fit, logp(x_fit))
    x0 = x_fit[:n_chain] * 0.5
    dd = den.device(ctx)
    out = {}
    try:
        for form in (8, 4, 40):   # (40: the four-wave form with the cubic contraction by its general loops instead of the written-out block)
            _lib.debug_set('cubic_form', form % 10 if form > 8 else form)
            _lib.debug_set('cubic_loops', int(form > 8))
            dc = DeviceChains(dd, x0, seed=5)
            kw = dict(n_warmup=24, max_treedepth=5)
            s1, st1 = dc.run(24, 'NUTS', **kw)
            s2, st2 = dc.run(8, 'NUTS', **kw)
            out[form] = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)] + [dc.total_leapfrog, _lib.last_kernel()]
    finally:
        _lib.debug_set('cubic_form', 0)
        _lib.debug_set('cubic_loops', 0)
    assert ', 16, 0>' in out[8][-1] and ', 17, 0>' in out[4][-1], (out[8][-1], out[4][-1])
    for other in (4, 40):
        for a, b in zip(out[8][:-2], out[other][:-2]):
            assert np.array_equal(a, b, equal_nan=True), other
        assert out[8][-2] == out[other][-2] > 0


def test_auto_layout_takes_the_wave_kernel_for_deep_trees_with_the_decay_term(ctx):
    """chains._deep_trees_prefer_waves: the common surrogate with the decay term at d = 64, trees in step -- 'auto' runs the group
    kernel while the trees have 7 leaves and the pipelined wave-per-chain kernel once they have 31 (the helper of the layout vote
    reports the common tree size); the forced group layout gives the same trees and moments."""
    if _LAYOUT['v'] != 'group':
        pytest.skip('does not depend on the module-wide layout parameter')
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    import torch
    d = 64
    n = 9 * torch.cuda.get_device_properties(0).multi_processor_count   # (more than eight chains per CU: not a "small problem")
    spec, _ = correlated_gaussian_spec(d)
    po = spec['poly']
    spec = dict(spec, use_decay=True, decay_mu=po['mu'], decay_hess=po['hess'], decay_alpha2=(1.5 * po['alpha'])**2, decay_gamma=0.1)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(4).normal(size=(n, d))
    ts = _lib.NSTATS.index('tree_size')
    for ta, leaves, want in ((0.8, 7, 'in step'), (0.99, 31, 'wave')):
        out = {}
        for lay in ('auto', 'group'):
            dc = DeviceChains(dens, x0, seed=5)
            kw = dict(n_warmup=500, layout=lay, launch_iters=100, target_accept=ta)
            dc.run(500, 'NUTS', **kw)
            s, st = dc.run(200, 'NUTS', **kw)
            out[lay] = (s.cpu().numpy(), st.cpu().numpy(), dc.sc.cpu().numpy(), dc.rng.cpu().numpy(), dc.last_layout, _lib.last_kernel())
        assert np.median(out['auto'][1][:, -50:, ts]) == leaves
        assert (out['auto'][4] == 'wave') == (want == 'wave'), (ta, out['auto'][4:])   # ('split' falls back to the group kernel here)
        assert ('bf_nuts_pipe_kernel<4' if want == 'wave' else 'bf_group_kernel<4, true, 3>') in out['auto'][5], out['auto'][5]
        # (the lane-per-chain and the wave-per-chain kernels agree to rounding, not bit for bit: the same trees, the same moments)
        assert np.median(out['group'][1][:, -50:, ts]) == leaves
        np.testing.assert_allclose(out['auto'][0][:, -100:].var((0, 1)), out['group'][0][:, -100:].var((0, 1)), rtol=0.05)


def test_auto_layout_runs_two_groups_per_cu_in_the_group_kernel_at_d32(ctx):
    """chains._two_groups_fit_a_cu: 32 chains per CU at 17 <= d <= 32 with the trees in step -- 'auto' runs the group kernel (two
    workgroups per CU, a wave per SIMD) where it ran the split kernel (one), with the same results bit for bit."""
    if _LAYOUT['v'] != 'group':
        pytest.skip('does not depend on the module-wide layout parameter')
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    import torch
    d = 32
    n = 32 * torch.cuda.get_device_properties(0).multi_processor_count
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(4).normal(size=(n, d))
    out = {}
    for lay in ('auto', 'split'):
        dc = DeviceChains(dens, x0, seed=5)
        kw = dict(n_warmup=400, layout=lay, launch_iters=100)
        dc.run(400, 'NUTS', **kw)
        s, st = dc.run(200, 'NUTS', **kw)
        out[lay] = (s.cpu().numpy(), st.cpu().numpy(), dc.sc.cpu().numpy(), dc.rng.cpu().numpy(), dc.last_layout, _lib.last_kernel())
    assert out['auto'][4] == 'group' and 'bf_group_kernel<2' in out['auto'][5], out['auto'][4:]
    assert out['split'][4] == 'split' and 'bf_split_kernel<2' in out['split'][5]
    for a, b in zip(out['auto'][:4], out['split'][:4]):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize('case', ['plain', 'decay_out'])
def test_the_tail_of_a_launch_in_few_chain_workgroups_never_changes_results(ctx, case):
    """Sixteen-chain launches of the wave-per-chain kernel run in two parts (bfhip_sampler.hip: launch_nuts_pipe): one of the last
    chains of a workgroup, with at least a quarter of the launch's iterations left while three quarters of the launch's chains are
    through, stops at the end of its iteration, and the stopped chains go on in a second launch, one to four per workgroup on
    4 x 4 x 4 tiles.  Samples, statistics, adapted state, random streams and the leapfrog count are EQUAL to the one-part launch's.
    Stragglers made on purpose: every 19th chain gets a step size of 0.03 (63-leaf trees at a depth limit of 6 where the others
    build 7), step-size adaptation off; the plain and the decay instantiation (every leaf outside the bound); two launches each (the
    second resumes)."""
    import ctypes
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    d = 64
    spec, _ = correlated_gaussian_spec(d, fit_scale=0.3 if case == 'decay_out' else 1.5)
    if case == 'decay_out':
        po = spec['poly']
        spec = dict(spec, use_decay=True, decay_mu=po['mu'] + 0.05, decay_hess=po['hess'], decay_alpha2=(1.5 * po['alpha'])**2, decay_gamma=0.1)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(7).normal(size=(150, d))
    L = _lib.lib()
    L.bfhip_debug_tail_count.argtypes = [ctypes.c_void_p]
    L.bfhip_debug_tail_count.restype = ctypes.c_int
    out, listed = {}, {}
    try:
        _lib.debug_set('wave_cpg', 16)
        for two in (1, 0):
            _lib.debug_set('tail_relaunch', two)
            dc = DeviceChains(dens, x0, seed=13, step_size=0.6)
            for f in ('log_step', 'log_bar'):
                dc.sc[::19, _lib.SC_FIELDS.index(f)] = np.log(0.03)
            kw = dict(n_warmup=0, layout='wave', adapt_step_size=False, adapt_metric=False, max_treedepth=6)
            s1, st1 = dc.run(40, 'NUTS', **kw)
            listed[two] = L.bfhip_debug_tail_count(ctx.handle) if two else 0
            s2, st2 = dc.run(12, 'NUTS', **kw)
            out[two] = [t.cpu().numpy() for t in (s1, st1, s2, st2, dc.sc, dc.vec, dc.rng)] + [dc.total_leapfrog]
    finally:
        _lib.debug_set('wave_cpg', 0)
        _lib.debug_set('tail_relaunch', 1)
    for a, b in zip(out[1][:-1], out[0][:-1]):
        assert np.array_equal(a, b, equal_nan=True)
    assert out[1][-1] == out[0][-1]
    ts = out[1][1][:, :, _lib.NSTATS.index('tree_size')].sum(1)
    assert ts[::19].min() > 3 * np.delete(ts, np.arange(0, 150, 19)).max()   # (the stragglers are stragglers)
    assert 1 <= listed[1] <= 8, listed   # (... and the second part ran them: 8 of the 150 chains)


def test_surrogate_with_input_scales_runs_on_the_fused_fast_kernels_and_matches_the_oracle(ctx):
    """A linear + quadratic surrogate WITH Surrogate.input_scales (module.py:190-226; every surrogate of the reference's recipes
    has them): the scaling is folded into the coefficients and the bound at upload (device.density_desc_from_spec), so NUTS runs
    on the pipelined / split / group kernels instead of the sliced kernel's generic instantiation -- and equals the oracle, which
    scales the input as the reference does: logp and gradient inside and outside the bound to 1e-10, trajectories, on every layout."""
    from oracle import oracle as orc
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    d = 64
    rng = np.random.default_rng(31)
    lo, diff = rng.normal(size=d), rng.uniform(0.5, 3., size=d)
    spec, _ = correlated_gaussian_spec(d)     # a surrogate in ITS input space; the density's input is x = lo + diff x_s
    spec = dict(spec, su_lo=lo, su_diff=diff)
    dens = DeviceDensity(spec, ctx)
    x = lo + diff * rng.normal(size=(40, d)) * np.where(np.arange(40)[:, None] < 20, 1., 6.)   # (half of them outside the bound)
    lp, g = [np.asarray(t.cpu()) if hasattr(t, 'cpu') else np.asarray(t) for t in dens.logp_and_grad(x)]
    lp0, g0 = orc.logp_and_grad(spec, x)
    np.testing.assert_allclose(lp, lp0, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(g, g0, rtol=1e-10, atol=1e-9)
    assert (lp0[20:] < lp0[:20].min()).all()   # (the second half is far out)
    x0 = lo + diff * rng.normal(size=(6, d))
    dev = _device_chains(ctx, spec, x0, 14, 9)
    orc_runs = _oracle_chains(spec, x0, 14, 9)
    _compare_nuts(dev, orc_runs, 14, n_head=6, tol_head=1e-8)
    kname = _lib.last_kernel
    assert not kname().startswith('bf_sampler_kernel'), kname()


def test_cubic_surrogate_with_input_scales_is_folded_too(ctx):
    """Cubic configs behind Surrogate.input_scales: the third-order expansion around x = 0 (device.density_desc_from_spec) against the
    oracle's scaled evaluation -- logp and gradient to 1e-9, NUTS trajectories -- at d = 24 (sliced kernel) with ragged masks."""
    from oracle import oracle as orc
    from bayesfast_amd.device import DeviceDensity, folds_input_scales
    d = 24
    rng = np.random.default_rng(33)
    spec = _cubic_spec(d=d, seed=5, m2=np.sort(rng.choice(d, 9, replace=False)), m3=np.sort(rng.choice(d, 7, replace=False)), amp=0.15)
    lo, diff = rng.normal(size=d) * 0.5, rng.uniform(0.7, 2., size=d)
    spec = dict(spec, su_lo=lo, su_diff=diff)
    assert folds_input_scales(spec)
    dens = DeviceDensity(spec, ctx)
    x = lo + diff * rng.normal(size=(30, d)) * 0.7
    lp, g = [np.asarray(t.cpu()) if hasattr(t, 'cpu') else np.asarray(t) for t in dens.logp_and_grad(x)]
    lp0, g0 = orc.logp_and_grad(spec, x)
    np.testing.assert_allclose(lp, lp0, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(g, g0, rtol=1e-9, atol=1e-9)
    x0 = lo + diff * rng.normal(size=(5, d)) * 0.5
    dev = _device_chains(ctx, spec, x0, 12, 8)
    orc_runs = _oracle_chains(spec, x0, 12, 8)
    _compare_nuts(dev, orc_runs, 12, n_head=6, tol_head=1e-7, rtol_q=1e-4)
