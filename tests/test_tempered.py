"""Tempered samplers: the oracle's restatement of TCpuLeapfrogIntegrator and TNUTS against fixtures
recorded from the reference with logged random draws (tests/golden/tempered.npz, make_golden.py:gen_tempered), and --
marked gpu -- the device kernel against the oracle on shared xoshiro streams."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(G, 'tempered.npz'))


def _specs(fx):
    from specio import rebuild_spec
    from oracle import oracle as orc
    spec = rebuild_spec(fx, 't6.')
    base = orc.gaussian_base_spec(fx['t6.base_mean'], fx['t6.base_cov'])
    return spec, base, float(fx['t6.logxi'])


def test_oracle_tempered_integrator_matches_reference(fx):
    """compute_state and two steps of TCpuLeapfrogIntegrator (integration.py:132-222)."""
    from oracle import oracle as orc
    spec, base, logxi = _specs(fx)
    u0, v0 = fx['t6.lf.u0v0']
    st = orc.tempered_states(spec, base, logxi, fx['t6.lf.var'], fx['t6.lf.q0'], fx['t6.lf.p0'], u0, v0, fx['t6.lf.eps'])
    for k, tag in enumerate(('s0', 's1', 's2')):
        for f in ('q', 'p', 'u', 'v', 'weight', 'energy', 'logp'):
            np.testing.assert_allclose(st[f][k], fx['t6.lf.%s.%s' % (tag, f)], rtol=1e-11, atol=1e-11, err_msg='%s %s' % (tag, f))


@pytest.mark.parametrize('c', [0, 1])
def test_oracle_tnuts_replays_reference_trajectories(fx, c):
    """TNUTS with the reference's logged draws: tree depth / size / divergence exactly, samples, u and weights closely."""
    from oracle import oracle as orc
    spec, base, logxi = _specs(fx)
    k = 't6.tnuts%d.' % c
    n_iter, n_warmup = int(fx['t6.n_iter']), int(fx['t6.n_warmup'])
    ch = orc.Chain(fx['t6.x0'][c])
    rng = orc.make_rng('replay', normals=fx[k + 'normals'], uniforms=fx[k + 'uniforms'])
    s, st, u_last = orc.tnuts_run(spec, base, logxi, ch, rng, float(fx[k + 'u_first']), n_iter, n_warmup)
    for f in ('tree_depth', 'tree_size', 'diverging', 'warmup'):
        assert np.array_equal(st[f], fx[k + f]), f
    np.testing.assert_allclose(s, fx[k + 'samples'], rtol=1e-8, atol=1e-8)
    for f in ('u', 'weight', 'logp', 'energy', 'mean_tree_accept', 'step_size', 'step_size_bar', 'energy_change', 'max_energy_change'):
        np.testing.assert_allclose(st[f], fx[k + f], rtol=1e-7, atol=1e-8, err_msg=f)
    assert rng[0].i_normal == fx[k + 'normals'].size and rng[0].i_uniform == fx[k + 'uniforms'].size


@pytest.mark.gpu
def test_device_tnuts_matches_oracle_on_shared_streams(fx):
    """bfhip_tnuts_run against the oracle's TNUTS on the same xoshiro256++ streams: tree depth / size / divergence exactly,
    positions, u and weights to 1e-8 on the head of the run (17 chains: a ragged workgroup)."""
    from oracle import oracle as orc
    from bayesfast_amd.device import get_context, DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd import _lib
    spec, base, logxi = _specs(fx)
    ctx = get_context(0)
    rng = np.random.default_rng(21)
    n_chain, n_iter, n_warmup = 17, 30, 20
    x0 = rng.normal(size=(n_chain, spec['d'])) * 0.5
    u0 = rng.normal(size=n_chain)
    dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=99)
    s, st, stt = dc.run_tempered(n_iter, fx['t6.base_mean'], fx['t6.base_cov'], logxi=logxi, u_0=u0, n_warmup=n_warmup)
    s, st, stt = s.cpu().numpy(), st.cpu().numpy(), stt.cpu().numpy()
    nl = 0
    for i in (0, 5, 16):
        so, sto, _ = orc.tnuts_run(spec, base, logxi, orc.Chain(x0[i]), orc.make_rng('xoshiro', seed=99, stream=i), u0[i], n_iter, n_warmup)
        for f in ('tree_depth', 'tree_size', 'diverging'):
            assert np.array_equal(st[i, :, _lib.NSTATS.index(f)], sto[f]), (i, f)
        np.testing.assert_allclose(s[i, :8], so[:8], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(stt[i, :8, 0], sto['u'][:8], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(stt[i, :8, 1], sto['weight'][:8], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(s[i], so, rtol=1e-4, atol=1e-4)
        for f in ('logp', 'energy', 'step_size'):
            np.testing.assert_allclose(st[i, :8, _lib.NSTATS.index(f)], sto[f][:8], rtol=1e-8, atol=1e-8, err_msg=f)
    assert dc.total_leapfrog == int(st[:, :, _lib.NSTATS.index('tree_size')].sum())
    # resume: a second call continues the chains (u included)
    s2, st2, stt2 = dc.run_tempered(5, fx['t6.base_mean'], fx['t6.base_cov'], logxi=logxi, n_warmup=n_warmup)
    so, sto, _ = orc.tnuts_run(spec, base, logxi, orc.Chain(x0[5]), orc.make_rng('xoshiro', seed=99, stream=5), u0[5], n_iter + 5, n_warmup)
    assert np.array_equal(st2.cpu().numpy()[5, :, _lib.NSTATS.index('tree_size')], sto['tree_size'][n_iter:])


@pytest.mark.gpu
@pytest.mark.parametrize('feature', ['bounds', 'decay', 'bounds_decay', 'decay32'])
def test_device_tnuts_behind_the_transform_and_with_decay_matches_oracle(fx, feature):
    """Round 5: the tempered sampler on targets with hard bounds / input scales (the constraint transform: density.py:92-140,
    747-750) and with the decay penalty (:740-746) -- what every GBS example of the reference has -- against the oracle's TNUTS
    (whose potentials are the general Density.logp_and_grad) on shared streams.  The base density stays in the sampler's space."""
    from oracle import oracle as orc
    from bayesfast_amd.device import get_context, DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    d = 32 if feature == 'decay32' else 12
    spec = dict(correlated_gaussian_spec(d, fit_scale=1.5)[0])
    po = spec['poly']
    if 'decay' in feature:
        spec.update(use_decay=True, decay_mu=np.asarray(po['mu']) + 0.05, decay_hess=po['hess'], decay_alpha2=(0.8 * float(po['alpha']))**2, decay_gamma=0.1)
    if 'bounds' in feature:
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        spec.update(ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array(([[1, 1], [1, 0], [0, 1], [0, 0]] * d)[:d], dtype=np.uint8))
    rng = np.random.default_rng(5)
    base_mean, base_cov = np.zeros(d), np.eye(d) * (0.3 if 'bounds' in feature else 1.5)
    base = orc.gaussian_base_spec(base_mean, base_cov)
    ctx = get_context(0)
    n_chain, n_iter, n_warmup = 11, 16, 10
    x0 = rng.normal(size=(n_chain, d)) * 0.3
    u0 = rng.normal(size=n_chain)
    dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=77)
    s, st, stt = dc.run_tempered(n_iter, base_mean, base_cov, logxi=0.3, u_0=u0, n_warmup=n_warmup)
    s, st, stt = s.cpu().numpy(), st.cpu().numpy(), stt.cpu().numpy()
    for i in (0, 4, 10):
        so, sto, _ = orc.tnuts_run(spec, base, 0.3, orc.Chain(x0[i]), orc.make_rng('xoshiro', seed=77, stream=i), u0[i], n_iter, n_warmup)
        for f in ('tree_depth', 'tree_size', 'diverging'):
            assert np.array_equal(st[i, :, _lib.NSTATS.index(f)], sto[f]), (feature, i, f, st[i, :, _lib.NSTATS.index(f)], sto[f])
        np.testing.assert_allclose(s[i, :6], so[:6], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(stt[i, :6, 0], sto['u'][:6], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(stt[i, :6, 1], sto['weight'][:6], rtol=1e-7, atol=1e-8)
        for f in ('logp', 'energy', 'step_size'):
            np.testing.assert_allclose(st[i, :6, _lib.NSTATS.index(f)], sto[f][:6], rtol=1e-8, atol=1e-8, err_msg=f)


@pytest.mark.gpu
def test_device_tnuts_workgroup_size_never_changes_results(fx):
    """bf_tnuts_kernel with 4 and 8 chains per workgroup (always eight waves: the ones without a chain run the shared matvec jobs
    only; the library takes four chains per workgroup when that spreads them over more CUs): samples, statistics, tempering
    coordinate and weights are EQUAL -- 37 chains, ragged for both."""
    from bayesfast_amd.device import get_context, DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd import _lib
    spec, base, logxi = _specs(fx)
    ctx = get_context(0)
    rng = np.random.default_rng(22)
    x0, u0 = rng.normal(size=(37, spec['d'])) * 0.5, rng.normal(size=37)
    out = {}
    try:
        for wpb in (4, 8):
            _lib.debug_set('tnuts_wpb', wpb)
            dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=98)
            out[wpb] = [t.cpu().numpy() for t in dc.run_tempered(24, fx['t6.base_mean'], fx['t6.base_cov'], logxi=logxi, u_0=u0, n_warmup=16)]
    finally:
        _lib.debug_set('tnuts_wpb', 0)
    for a, b in zip(out[4], out[8]):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.gpu
def test_sample_entry_point_runs_tnuts_and_tempering_brings_base_mass(fx):
    """bayesfast_amd.sample(..., sampler='TNUTS'): TraceTuple with the (u, weight) statistics; continuing a run; 'THMC' is
    refused (the reference's THTrace cannot be constructed, samplers/sample_trace.py:600)."""
    import bayesfast_amd as bfa
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 8
    _, cov = correlated_gaussian_spec(d)
    prec = np.linalg.inv(cov)
    rng = np.random.default_rng(2)
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
    dens = bfa.SurrogateDensity(su)
    xf = rng.multivariate_normal(np.zeros(d), cov * 2.25, size=4 * su.n_param)
    dens.fit(xf, -0.5 * np.einsum('ij,jk,ik->i', xf, prec, xf))
    np.random.seed(0)
    kw = dict(density_base=bfa.GaussianBase(np.zeros(d), cov * 1.5), logxi=0., n_chain=24, n_iter=200, n_warmup=100, random_generator=4)
    tt = bfa.sample(dens, kw, sampler='TNUTS', n_run=150, verbose=False)
    assert tt.sampler == 'TNUTS' and tt.samples.shape == (24, 150, d) and tt.stat('u').shape == (24, 150)
    w = tt.stat('weight')
    assert np.isfinite(w).all() and (w > 0).all()
    tt = bfa.sample(dens, tt, verbose=False)
    assert tt.samples.shape == (24, 200, d) and tt.stat('weight').shape == (24, 200)
    # the tempered chain mixes target and base: its spread lies between the two (base covariance = 1.5 x target's)
    ratio = np.mean(tt.samples[:, 100:].reshape(-1, d).var(0) / np.diag(cov))
    assert 0.9 < ratio < 1.7
    with pytest.raises(NotImplementedError):
        bfa.sample(dens, dict(n_chain=4), sampler='THMC')


@pytest.mark.gpu
def test_sample_tnuts_on_a_bounded_density_with_decay():
    """bayesfast_amd.sample(..., sampler='TNUTS') on a density with input_scales, hard bounds and the decay term -- the shape of
    the reference's GBS notebooks -- runs on the device (round 5; it was refused before) and stays inside the bounds."""
    import bayesfast_amd as bfa
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 8
    _, cov = correlated_gaussian_spec(d)
    prec = np.linalg.inv(cov)
    rng = np.random.default_rng(2)
    rg = np.stack([np.full(d, -6.), np.full(d, 6.)], 1)
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
    dens = bfa.SurrogateDensity(su, input_scales=rg, hard_bounds=True, decay_options=dict(use_decay=True))
    xf = np.clip(rng.multivariate_normal(np.zeros(d), cov * 2.25, size=4 * su.n_param), -5.9, 5.9)
    dens.fit(xf, -0.5 * np.einsum('ij,jk,ik->i', xf, prec, xf))
    np.random.seed(0)
    kw = dict(density_base=bfa.GaussianBase(np.zeros(d), np.eye(d) * 0.05), logxi=0., n_chain=24, n_iter=160, n_warmup=80, random_generator=4)
    tt = bfa.sample(dens, kw, sampler='TNUTS', verbose=False)
    xs = tt.get(flatten=True, original_space=True)
    assert tt.sampler == 'TNUTS' and xs.shape == (24 * 80, d)
    assert np.isfinite(xs).all() and (xs > -6.).all() and (xs < 6.).all()
    w = tt.stat('weight')
    assert np.isfinite(w).all() and (w > 0).all()
    ratio = np.mean(xs.var(0) / np.diag(cov))
    assert 0.5 < ratio < 2.0, ratio


# ---- round 6: the generic tempered kernel (bfhip_tnuts_gen.hip) -- TNUTS on everything NUTS runs on ----

def _tnuts_against_oracle(spec, x0, u0, base_mean, base_cov, logxi, n_iter, n_warmup, seed, chains, head=6, metric=None, tol=1e-8):
    """Run the device's TNUTS and the oracle's on shared xoshiro streams; discrete fields exactly, the head of the run closely."""
    from oracle import oracle as orc
    from bayesfast_amd.device import get_context, DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd import _lib
    base = orc.gaussian_base_spec(base_mean, base_cov)
    dc = DeviceChains(DeviceDensity(spec, get_context(0)), x0, seed=seed, metric=metric)
    s, st, stt = dc.run_tempered(n_iter, base_mean, base_cov, logxi=logxi, u_0=u0, n_warmup=n_warmup)
    s, st, stt = s.cpu().numpy(), st.cpu().numpy(), stt.cpu().numpy()
    for i in chains:
        ch = orc.Chain(x0[i]) if metric is None else orc.Chain(x0[i], metric=np.eye(x0.shape[1]) if isinstance(metric, str) else metric)
        so, sto, _ = orc.tnuts_run(spec, base, logxi, ch, orc.make_rng('xoshiro', seed=seed, stream=i), u0[i], n_iter, n_warmup)
        for f in ('tree_depth', 'tree_size', 'diverging'):
            assert np.array_equal(st[i, :, _lib.NSTATS.index(f)], sto[f]), (i, f, st[i, :, _lib.NSTATS.index(f)], sto[f])
        np.testing.assert_allclose(s[i, :head], so[:head], rtol=tol, atol=tol)
        np.testing.assert_allclose(stt[i, :head, 0], sto['u'][:head], rtol=tol, atol=tol)
        np.testing.assert_allclose(stt[i, :head, 1], sto['weight'][:head], rtol=10 * tol, atol=tol)
        for f in ('logp', 'energy', 'step_size'):
            np.testing.assert_allclose(st[i, :head, _lib.NSTATS.index(f)], sto[f][:head], rtol=tol, atol=tol, err_msg=f)
    assert dc.total_leapfrog == int(st[:, :, _lib.NSTATS.index('tree_size')].sum())
    return dc, s, st, stt


@pytest.mark.gpu
@pytest.mark.parametrize('feature', ['plain', 'decay', 'bounds'])
def test_generic_tempered_kernel_agrees_with_the_tuned_one(fx, feature):
    """The generic kernel forced onto densities the tuned kernel covers (bfhip_debug_set('tnuts_generic', 1)): the same trees,
    the same numbers to rounding -- and the kernel that ran is the one asked for."""
    from bayesfast_amd.device import get_context, DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    d = 24
    spec = dict(correlated_gaussian_spec(d, fit_scale=1.5)[0])
    po = spec['poly']
    if feature == 'decay':
        spec.update(use_decay=True, decay_mu=np.asarray(po['mu']) + 0.05, decay_hess=po['hess'], decay_alpha2=(0.8 * float(po['alpha']))**2, decay_gamma=0.1)
    if feature == 'bounds':
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        spec.update(ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array(([[1, 1], [1, 0], [0, 1], [0, 0]] * d)[:d], dtype=np.uint8))
    rng = np.random.default_rng(8)
    x0, u0 = rng.normal(size=(19, d)) * 0.3, rng.normal(size=19)
    out = {}
    try:
        for gen in (0, 1):
            _lib.debug_set('tnuts_generic', gen)
            dc = DeviceChains(DeviceDensity(spec, get_context(0)), x0, seed=31)
            out[gen] = [t.cpu().numpy() for t in dc.run_tempered(20, np.zeros(d), np.eye(d) * (0.3 if feature == 'bounds' else 1.5), logxi=0.2, u_0=u0, n_warmup=12)]
    finally:
        _lib.debug_set('tnuts_generic', 0)
    ts = _lib.NSTATS.index('tree_size')
    assert np.array_equal(out[0][1][:, :, ts], out[1][1][:, :, ts])
    np.testing.assert_allclose(out[1][0][:, :8], out[0][0][:, :8], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out[1][2][:, :8], out[0][2][:, :8], rtol=1e-8, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('cubic,decay,transform,su,link', [(1, 0, 0, 0, 0), (1, 1, 1, 1, 0), (0, 0, 0, 1, 1), (1, 0, 0, 0, 1)])
def test_device_tnuts_on_cubic_scaled_and_linked_surrogates_matches_oracle(cubic, decay, transform, su, link):
    """Cubic configs (inside and outside the bound: the second rendezvous at the projected point), device-side surrogate input
    scaling, the Gaussian link, and their combinations with transform and decay."""
    from test_oracle_golden import _random_single_output_spec
    d = 12
    rng = np.random.default_rng(300 + 16 * cubic + 8 * decay + 4 * transform + 2 * su + link)
    spec = _random_single_output_spec(d, rng, cubic, decay, transform, su, link)
    # make the quadratic part a proper log-density so that the chains stay put: a negative definite form
    A = rng.normal(size=(d, d)) * 0.2
    P = A @ A.T + np.eye(d)
    q = np.zeros((d, d))
    iu = np.triu_indices(d)
    q[iu] = (-0.5 * P)[iu] * np.where(iu[0] == iu[1], 1., 2.)
    spec['poly']['configs'][1]['coef'] = q[None]
    from oracle import oracle as orc
    xs = rng.normal(size=(40 * d, d))
    spec['poly'].update(orc.set_bound(dict(spec['poly'], use_bound=False), xs, -0.5 * np.einsum('ij,jk,ik->i', xs, P, xs), dict(alpha_p=60.)))
    n_chain = 13
    x0 = rng.normal(size=(n_chain, d)) * 0.4
    u0 = rng.normal(size=n_chain)
    _tnuts_against_oracle(spec, x0, u0, np.zeros(d), np.eye(d) * (0.4 if transform else 1.2), 0.1, 14, 8, 41, (0, 5, 12))


@pytest.mark.gpu
@pytest.mark.parametrize('cubic', [0, 1])
def test_device_tnuts_at_128_dimensions_matches_oracle(cubic):
    """d = 128 (two elements per lane, the A fragments streamed from L2), plain and with config 5's cubic configs on 16 inputs."""
    from oracle import oracle as orc
    d = 128
    rng = np.random.default_rng(128 + cubic)
    A = rng.normal(size=(d, d)) * 0.05
    P = A @ A.T + np.diag(np.logspace(0, 1, d))
    q = np.zeros((d, d))
    iu = np.triu_indices(d)
    q[iu] = (-0.5 * P)[iu] * np.where(iu[0] == iu[1], 1., 2.)
    cfgs = [dict(order='linear', input_mask=np.arange(d), output_mask=np.arange(1), coef=np.concatenate(([[0.3]], 0.05 * rng.normal(size=(1, d))), 1)),
            dict(order='quadratic', input_mask=np.arange(d), output_mask=np.arange(1), coef=q[None])]
    if cubic:
        m16 = np.arange(16)
        a3 = np.zeros((1, 16, 16, 16))
        for j in range(16):
            for k in range(j + 1, 16):
                for l in range(k + 1, 16):
                    a3[0, j, k, l] = 0.01 * rng.normal()
        cfgs += [dict(order='cubic-2', input_mask=m16, output_mask=np.arange(1), coef=0.01 * rng.normal(size=(1, 16, 16))),
                 dict(order='cubic-3', input_mask=m16, output_mask=np.arange(1), coef=a3)]
    poly = dict(input_size=d, output_size=1, configs=cfgs, use_bound=False)
    xs = rng.normal(size=(6 * d, d)) / np.sqrt(np.diag(P))
    poly.update(orc.set_bound(poly, xs, -0.5 * np.einsum('ij,jk,ik->i', xs, P, xs), dict(alpha_p=70.)))
    spec = dict(d=d, poly=poly)
    n_chain = 9
    x0 = rng.normal(size=(n_chain, d)) * 0.5 / np.sqrt(np.diag(P))
    u0 = rng.normal(size=n_chain)
    _tnuts_against_oracle(spec, x0, u0, np.zeros(d), np.diag(1.3 / np.diag(P)), 0., 8, 5, 52, (0, 8), head=4)


@pytest.mark.gpu
@pytest.mark.parametrize('adapt', [False, True])
def test_device_tnuts_with_the_full_rank_metric_matches_oracle(adapt):
    """QuadMetricFull / QuadMetricFullAdapt under the tempered sampler (per-chain covariance, momentum through the Cholesky factor,
    Welford windows and refactorisation while adapting)."""
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 10
    spec, cov = correlated_gaussian_spec(d, fit_scale=1.5)
    rng = np.random.default_rng(17)
    n_chain = 7
    x0 = rng.normal(size=(n_chain, d)) * 0.4
    u0 = rng.normal(size=n_chain)
    n_iter, n_warm = (14, 10) if adapt else (10, 0)
    _tnuts_against_oracle(spec, x0, u0, np.zeros(d), cov * 1.4, 0.1, n_iter, n_warm, 61, (0, 6), metric=cov * 0.9, head=5)


@pytest.mark.gpu
def test_device_tnuts_on_the_pipeline_density_matches_oracle():
    """The pipeline density (multi-output surrogate + chi-square + prior, behind the transform, with the bound): the tempered
    kernel's rendezvous carries the two contractions; against the oracle's TNUTS on the same density."""
    from bayesfast_amd.workloads import random_pipeline_spec
    for m_out, d, nq, seed in ((40, 9, 4, 2), (120, 20, 6, 3)):
        spec = random_pipeline_spec(m_out, d, nq, seed=seed)
        rng = np.random.default_rng(seed)
        n_chain = 11
        x0 = rng.normal(size=(n_chain, d)) * 0.3
        u0 = rng.normal(size=n_chain)
        _tnuts_against_oracle(spec, x0, u0, np.zeros(d), np.eye(d) * 0.5, 0.2, 10, 6, 71 + seed, (0, 3, 10), head=5)


@pytest.mark.gpu
def test_device_tnuts_on_the_des_shaped_pipeline_matches_oracle():
    """TNUTS on the DES-shaped fixture density (27 inputs behind the box transform with hard bounds, surrogate input scales,
    whitened chi-square, Gaussian prior; pipeline_des.npz 'a' and, with the decay term, 'b')."""
    from specio import rebuild_pipeline_des
    z = np.load(os.path.join(G, 'pipeline_des.npz'))
    for tag in ('a', 'b'):
        spec = rebuild_pipeline_des(z, tag)
        d = int(spec['d'])
        rng = np.random.default_rng(9)
        x0 = np.asarray(z[tag + '.x0'])[:2]
        x0 = np.concatenate([x0, x0[:1] + 0.05 * rng.normal(size=(3, d))])
        u0 = rng.normal(size=x0.shape[0])
        _tnuts_against_oracle(spec, x0, u0, x0.mean(0), np.eye(d) * 0.05, 0., 8, 5, 91, (0, 4), head=4, tol=1e-7)
