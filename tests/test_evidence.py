"""Evidence path (GBS = SIT + bridge) against fixtures recorded from the reference
(tests/golden/evidence.npz, make_golden.py:gen_evidence).  CPU: the oracle's restatement of utils/_cubic.pyx and
kde.cdf, and the host-side spline construction; GPU (marked): the device kernels through the C ABI, a SIT fit that
reproduces the reference's rotations and splines, logq / forward / backward transforms, bridge(), and the 16-d funnel's
known logZ."""
import os
import warnings

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(G, 'evidence.npz'))


def test_oracle_spline_kernels_and_kde_cdf_match_reference(fx):
    from oracle import oracle as orc
    c, x, y = fx['spl.c'], fx['spl.x'], fx['spl.y']
    for mode, pts in (('evaluate', fx['spl.pts']), ('derivative', fx['spl.pts']), ('solve', fx['spl.ypts'])):
        np.testing.assert_allclose(orc.spline_apply(mode, c, x, y, pts), fx['spl.' + mode], rtol=1e-13, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(orc.kde_cdf(fx['kde.x'], fx['kde.w'], float(fx['kde.h']), fx['kde.pts']), fx['kde.cdf'],
                               rtol=1e-12, atol=1e-15)
    assert abs(orc.kde_bandwidth(fx['kde.x'], fx['kde.w']) - float(fx['kde.h'])) < 1e-14


def test_spline_construction_matches_reference(fx):
    """utils/cubic.py:61-140 restated in bayesfast_amd/utils/spline.py: same knots (percentiles, gap filling, monotonicity
    refinement), values and coefficient rows, with the Gaussianizing map evaluated by the oracle's kde.cdf."""
    from scipy.stats import norm
    from oracle import oracle as orc
    from bayesfast_amd.utils.spline import GaussianizingSpline
    xs, w, h = fx['kde.x'], fx['kde.w'], float(fx['kde.h'])
    sp = GaussianizingSpline(xs, lambda p: norm.ppf(orc.kde_cdf(xs, w, h, p)))
    assert sp.x.shape == fx['spl.x'].shape
    np.testing.assert_allclose(sp.x, fx['spl.x'], rtol=0, atol=0)
    np.testing.assert_allclose(sp.y, fx['spl.y'], rtol=1e-11, atol=1e-11)
    # the cubic and quadratic coefficients are differences of the values over knot spacings of 1e-4: the last digits of
    # kde.cdf (scipy's ndtr vs libm's erfc) are amplified there (they are not compared); the map itself agrees to 1e-9
    np.testing.assert_allclose(sp.c[:, 2:], fx['spl.c'][:, 2:], rtol=1e-7, atol=1e-8)
    for mode, pts in (('evaluate', fx['spl.pts']), ('derivative', fx['spl.pts']), ('solve', fx['spl.ypts'])):
        np.testing.assert_allclose(orc.spline_apply(mode, sp.c, sp.x, sp.y, pts), fx['spl.' + mode], rtol=1e-9, atol=1e-10, equal_nan=True)


def test_percentile_sorted_is_bitwise_numpys_percentile():
    """utils/spline.py computes its three percentile sets from one sorted copy of the data; the values must be numpy's."""
    from bayesfast_amd.utils.spline import percentile_sorted
    rng = np.random.default_rng(5)
    for n in (7, 100, 1001, 40000):
        x = np.round(rng.normal(size=n) * 3., 2 if n < 1001 else 6)  # (rounded: ties)
        xs = np.sort(x)
        for q in (np.linspace(0, 100, 101), np.linspace(0, 100, 12)[1:-1], np.array([0., 100., 50., 33.3333]), rng.uniform(0, 100, 200)):
            assert np.array_equal(percentile_sorted(xs, q), np.percentile(x, q))


def test_integrated_time_matches_the_reference(fx):
    """utils/acor.py against values recorded from the reference's estimator (vectorised here over walkers and dimensions)."""
    from bayesfast_amd.utils.acor import integrated_time
    np.testing.assert_allclose(integrated_time(fx['acor.x']), fx['acor.tau3'], rtol=1e-12)
    np.testing.assert_allclose(integrated_time(fx['acor.x'][0]), fx['acor.tau2'], rtol=1e-12)
    np.testing.assert_allclose(integrated_time(fx['acor.x'][1, :, 0]), fx['acor.tau1'], rtol=1e-12)


def test_integrated_time_matches_known_ar1():
    """AR(1) with coefficient 0.6: tau = (1 + rho) / (1 - rho) = 4."""
    from bayesfast_amd.utils.acor import integrated_time
    rng = np.random.default_rng(0)
    e = rng.normal(size=(8, 20000))
    for t in range(1, e.shape[1]):
        e[:, t] = 0.6 * e[:, t - 1] + 0.8 * e[:, t]
    tau = integrated_time(e[..., None])[0]
    assert abs(tau - 4.) < 0.25


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_kde_cdf_spline_apply_and_bridge_match_reference(fx):
    import torch
    from bayesfast_amd import _lib
    from bayesfast_amd.device import get_context, _ptr
    from bayesfast_amd.utils.spline import SplineTable
    from bayesfast_amd.evidence import bridge
    ctx = get_context(0)
    x = ctx.tensor(fx['kde.x'][None])
    w, h, p = ctx.tensor(fx['kde.w']), ctx.tensor(fx['kde.h'].reshape(1)), ctx.tensor(fx['kde.pts'][None])
    out = torch.empty_like(p)
    _lib.check(ctx._lib.bfhip_kde_cdf(ctx.handle, 1, x.shape[1], _ptr(x), _ptr(w), _ptr(h), p.shape[1], _ptr(p), _ptr(out)))
    np.testing.assert_allclose(out.cpu().numpy()[0], fx['kde.cdf'], rtol=1e-12, atol=1e-15)

    class S:  # the recorded arrays of one spline, twice (d = 2) to exercise the per-dimension offsets
        pass
    s = S()
    s.x, s.y, s.c = fx['spl.x'], fx['spl.y'], fx['spl.c']
    tab = SplineTable([s, s], ctx)
    for mode, pts in (('evaluate', fx['spl.pts']), ('derivative', fx['spl.pts']), ('solve', fx['spl.ypts'])):
        got = tab.apply(mode, ctx.tensor(np.stack([pts, pts[::-1]], 1))).cpu().numpy()
        np.testing.assert_allclose(got[:, 0], fx['spl.' + mode], rtol=1e-13, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(got[::-1, 1], fx['spl.' + mode], rtol=1e-13, atol=1e-13, equal_nan=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logr, err = bridge(fx['br.lpp'], fx['br.lpq'], fx['br.lqp'], fx['br.lqq'])
    assert abs(logr - float(fx['br.logr'])) < 1e-9
    assert abs(err - float(fx['br.err'])) < 1e-9 * max(1., float(fx['br.err']))


@pytest.mark.gpu
def test_sit_fit_logq_and_transforms_reproduce_the_reference(fx):
    """Same data, same seeds, same FastICA: rotations, splines, Gaussianized data, logq, forward and backward
    transforms of the reference's SIT (transforms/sit.py:223-459)."""
    from bayesfast_amd.transforms import SIT
    D = fx['sit.data'].shape[1]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sit = SIT(n_iter=3, random_generator=11)
        sit.fit(fx['sit.data'])
    # the first iteration is reproduced closely (FastICA has not converged after its 100 iterations -- the reference warns
    # too -- so the 1e-10 differences of the Gaussianized data move the later rotations: those are checked through the
    # model they define, below)
    np.testing.assert_allclose(sit._A[0], fx['sit.A'][0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(sit._m[0], fx['sit.m'][0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sit._logdetA[0], fx['sit.logdetA'][0], rtol=1e-8, atol=1e-9)
    for j in range(D):
        s = sit._tables[0].splines[j]
        assert s.x.shape == fx['sit.it0.d%d.x' % j].shape, j
        np.testing.assert_allclose(s.x, fx['sit.it0.d%d.x' % j], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(s.y, fx['sit.it0.d%d.y' % j], rtol=1e-7, atol=1e-8)
    # my own three iterations Gaussianize the data: unit covariance, and logq integrates to one over the fitted sample
    z = sit.data
    assert np.abs(np.cov(z, rowvar=False) - np.eye(D)).max() < 0.08 and np.abs(z.mean(0)).max() < 0.05
    xq, ljq, yq = sit.sample(3000)
    lq = sit.logq(xq)
    np.testing.assert_allclose(lq, np.sum(-0.5 * yq**2 - 0.9189385332046727, -1) + ljq, rtol=1e-6, atol=1e-5)  # q(x) = N(y) |dy/dx|
    # the reference's fitted parameters, loaded as they are: logq, forward and backward transforms (transforms/sit.py:372-459)
    ref = SIT._from_parts(fx['sit.A'], fx['sit.B'], fx['sit.m'], fx['sit.logdetA'],
                          [[tuple(fx['sit.it%d.d%d.%s' % (i, j, k)] for k in 'xyc') for j in range(D)] for i in range(3)])
    np.testing.assert_allclose(ref.logq(fx['sit.xt']), fx['sit.logq'], rtol=1e-10, atol=1e-9)
    y, lj = ref.forward_transform(fx['sit.xt'])
    np.testing.assert_allclose(y, fx['sit.forward_y'], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(lj, fx['sit.forward_logj'], rtol=1e-10, atol=1e-9)
    xb, ljb = ref.backward_transform(fx['sit.yb'])
    np.testing.assert_allclose(xb, fx['sit.backward_x'], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(ljb, fx['sit.backward_logj'], rtol=1e-8, atol=1e-7)


@pytest.mark.gpu
def test_gbs_recovers_the_16d_funnel_evidence():
    """examples/funnel-gbs.ipynb: 16-d funnel (a = 1, b = 0.5) under a flat prior on [-4, 4] x [-30, 30]^15; fiducial
    logZ = -63.4988.  Exact posterior draws stand in for the NUTS chains (8 x 1500)."""
    from bayesfast_amd.evidence import GBS
    D, a, b = 16, 1., 0.5
    const = np.log(8.) + (D - 1) * np.log(60.)

    def logp(x):
        n = x.shape[-1]
        return (-0.5 * x[..., 0]**2 / a**2 - 0.5 * np.sum(x[..., 1:]**2, axis=-1) * np.exp(-2 * b * x[..., 0])
                - 0.5 * np.log(2 * np.pi * a**2) - 0.5 * (n - 1) * np.log(2 * np.pi) - (n - 1) * b * x[..., 0] - const)

    rng = np.random.default_rng(16)
    x0 = rng.normal(size=(8, 1500)) * a
    xs = np.concatenate((x0[..., None], rng.normal(size=(8, 1500, D - 1)) * np.exp(b * x0)[..., None]), -1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz, err = GBS(sit=dict(random_generator=5), n_q=12000)(xs, logp)
    assert 0. < err < 0.2
    assert abs(logz - (-63.4988)) < 3. * err + 0.02


def test_device_fastica_algorithm_equals_scikit_learns(monkeypatch):
    """transforms/ica.py against ``sklearn.decomposition.FastICA`` (what the reference's SIT calls, transforms/sit.py:235-244)
    on the same data and ``random_state``: the same unmixing matrix and mean to 1e-8, the same number of iterations.  The
    device products are replaced by CPU tensors here (the algorithm is what is checked; tests/test_evidence.py's GPU tests
    run it on the device, where the Gram matrix comes from ``bfhip_gram``)."""
    import sys
    import warnings
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers'))
    from sklearn.decomposition import FastICA
    from oracle_standin import _CpuCtx
    from bayesfast_amd.transforms import ica
    monkeypatch.setattr(ica, '_gram', lambda ctx, xc: xc.T @ xc)
    rng = np.random.default_rng(0)
    for n, d, seed in ((4000, 5, 3), (20000, 16, 11)):
        s = np.stack([rng.laplace(size=n) if k % 3 == 0 else (rng.uniform(-1, 1, size=n) if k % 3 == 1 else rng.normal(size=n)**3)
                      for k in range(d)], 1)
        x = s @ rng.normal(size=(d, d)) + rng.normal(size=d)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref = FastICA(max_iter=100, random_state=seed).fit(x)
            comp, mean, n_iter = ica.fastica_device(x, random_state=seed, max_iter=100, ctx=_CpuCtx())
        assert n_iter == ref.n_iter_
        np.testing.assert_allclose(mean, ref.mean_, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(comp, ref.components_, rtol=1e-8, atol=1e-8 * np.abs(ref.components_).max())


@pytest.mark.gpu
def test_gbs_recovers_the_32d_banana_evidence():
    """examples/banana-gbs.ipynb: 32-d rotated bananas (Q = 0.01) under a flat prior on [-15, 15]^32; fiducial
    logZ = 16 log(pi sqrt(Q)) - 32 log 30 = -127.364 (the reference's own run printed
    -127.276 +- 0.053).  Exact posterior draws stand in for the NUTS chains (8 x 1500)."""
    from scipy.stats import special_ortho_group
    from bayesfast_amd.evidence import GBS
    D, Q = 32, 0.01
    const = D * np.log(30.)
    A = special_ortho_group.rvs(D, random_state=0)

    def logp(x):
        x = x @ A.T
        return -np.sum((x[..., ::2]**2 - x[..., 1::2])**2 / Q + (x[..., ::2] - 1)**2, axis=-1) - const

    rng = np.random.default_rng(1)
    xe = 1. + rng.normal(size=(8, 1500, D // 2)) / np.sqrt(2.)
    xo = xe**2 + rng.normal(size=(8, 1500, D // 2)) * np.sqrt(Q / 2.)
    z = np.empty((8, 1500, D))
    z[..., ::2], z[..., 1::2] = xe, xo
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz, err = GBS(sit=dict(random_generator=5), n_q=12000)(z @ A, logp)
    assert 0. < err < 0.25
    assert abs(logz - (-127.364)) < 3. * err + 0.02


@pytest.mark.gpu
def test_gbs_recovers_the_48d_cauchy_mixture_evidence():
    """examples/cauchy-gbs.ipynb: per dimension an equal mixture of two unit Cauchy densities at -5 and 5, under a flat prior
    on [-100, 100]^48 -- heavy tails, two modes per dimension.  logZ = 48 log((atan 105 + atan 95) / pi) - 48 log 200 =
    -254.627.  Exact (truncated) posterior draws stand in for the NUTS chains (8 x 1500)."""
    from bayesfast_amd.evidence import GBS
    D, a = 48, 5.
    const = D * np.log(200.)

    def logp(x):
        return (np.sum(np.log(1 / ((x + a)**2 + 1) + 1 / ((x - a)**2 + 1)), axis=-1) + x.shape[-1] * np.log(0.5 / np.pi) - const)

    rng = np.random.default_rng(1)
    shape = (8, 1500, D)
    cen = np.where(rng.uniform(size=shape) < 0.5, -a, a)
    lo, hi = np.arctan(-100. - cen), np.arctan(100. - cen)  # the truncated Cauchy by its inverse cdf
    x = cen + np.tan(lo + rng.uniform(size=shape) * (hi - lo))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz, err = GBS(sit=dict(random_generator=5), n_q=12000)(x, logp)
    fiducial = D * np.log((np.arctan(105.) + np.arctan(95.)) / np.pi) - const
    assert abs(fiducial - (-254.627)) < 1e-3
    assert 0. < err < 0.2
    assert abs(logz - fiducial) < 3. * err + 0.02


@pytest.mark.gpu
def test_gbs_recovers_the_64d_ring_evidence():
    """examples/ring-gbs.ipynb: the 64-d ring, logp = -sum_i (x_i^2 + x_{i+1}^2 - 2)^2 (cyclic) under a flat prior on
    [-5, 5]^64 -- a curved, strongly non-Gaussian ridge in every pair of neighbours.  Fiducial logZ = -114.492 (BASELINE.md
    section 2, examples/ring-gbs.ipynb:284; the reference's own run printed -114.473 +- 0.065 from 8 x 1500 NUTS draws).
    Posterior draws come from a vectorised HMC (tests/helpers/ring.py); SIT's rotations run on the device
    (transforms/ica.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers'))
    import ring
    from bayesfast_amd.evidence import GBS
    xs, acc = ring.hmc_draws(n_chain=8, n_keep=1500, seed=0)
    assert acc > 0.8
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz, err = GBS(sit=dict(random_generator=5), n_q=12000)(xs, ring.logp)
    assert 0. < err < 0.25
    assert abs(logz - (-114.492)) < 3. * err + 0.05, (logz, err)


@pytest.mark.gpu
def test_sit_iteration_at_full_size_runs_on_the_device():
    """One SIT iteration at 64 dimensions x 400 000 points (the size of a config-5 evidence run): with FastICA on the device
    it takes a fraction of a second (3.6 s with scikit-learn's on the host); the rotation it finds whitens the data."""
    import time
    import torch
    from bayesfast_amd.transforms import SIT
    rng = np.random.default_rng(2)
    x = rng.laplace(size=(400000, 64)) @ (np.eye(64) + 0.2 * rng.normal(size=(64, 64)))
    sit = SIT(n_iter=2, random_generator=3)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sit.fit(x, n_run=1)          # first call: kernels, allocator
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sit.fit(n_run=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert dt < 1.5, dt
    assert sit.i_iter == 2 and np.isfinite(sit.data).all()
    print('SIT iteration, 400000 x 64: %.3f s' % dt)


@pytest.mark.gpu
def test_sample_then_gbs_end_to_end_on_a_gaussian_surrogate():
    """The whole device path in one line of a recipe: fit a quadratic surrogate, sample it with NUTS (bayesfast_amd.sample), hand
    the TraceTuple to GBS with the surrogate's own logp.  The surrogate is an unnormalised Gaussian, so its evidence is
    known: log Z = c0 + d/2 log(2 pi) + 1/2 log det Sigma."""
    import bayesfast_amd as bfa
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 12
    _, cov = correlated_gaussian_spec(d)
    prec = np.linalg.inv(cov)
    c0 = -3.5
    rng = np.random.default_rng(7)
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
    den = bfa.SurrogateDensity(su)
    xf = rng.normal(size=(4 * su.n_param, d)) @ np.linalg.cholesky(cov).T * 1.6
    den.fit(xf, c0 - 0.5 * np.einsum('ij,jk,ik->i', xf, prec, xf))
    tt = bfa.sample(den, {'n_chain': 16, 'n_iter': 1500, 'n_warmup': 500, 'random_generator': 4}, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz, err = bfa.GBS(sit=dict(random_generator=5), n_q=12000)(tt, den.logp)
    exact = c0 + 0.5 * d * np.log(2 * np.pi) + 0.5 * np.linalg.slogdet(cov)[1]
    assert 0. < err < 0.1
    assert abs(logz - exact) < 3. * err + 0.02, (logz, err, exact)
    # that call took GBS's device-resident route (a TraceTuple, the default generator, a SurrogateDensity's logp: halves, draws and
    # log-densities stay on the GPU); the host route -- the same TraceTuple with logp wrapped so that it is just a callable -- is the
    # same estimate: same SIT (same seeds), same draws, log-densities to summation order
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz_h, err_h = bfa.GBS(sit=dict(random_generator=5), n_q=12000)(tt, lambda x: den.logp(x))
        logz_a, err_a = bfa.GBS(sit=dict(random_generator=5), n_q=12000)(tt.get(flatten=False), den.logp)
    assert abs(logz - logz_h) < 1e-9 and abs(err - err_h) < 1e-9 * err
    assert abs(logz_a - logz_h) < 1e-9
    # logp_p given (evidence/gaussianized.py:203-211): used when its shape fits, recomputed (with the reference's warning) when not
    lp = tt.get(return_type='logp', flatten=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logz_k, _ = bfa.GBS(sit=dict(random_generator=5), n_q=12000)(tt, den.logp, logp_p=lp)
    assert abs(logz_k - logz) < 1e-6
    with pytest.warns(RuntimeWarning, match='seems not correct'):
        logz_w, _ = bfa.GBS(sit=dict(random_generator=5), n_q=12000)(tt, den.logp, logp_p=lp[:, :-1])
    assert abs(logz_w - logz) < 1e-9


@pytest.mark.gpu
def test_device_ndtri_is_scipys():
    """bfhip_ndtri (Cephes' algorithm, as scipy.special.ndtri -- what the reference's norm.ppf calls evaluate, utils/sobol.py:57,
    transforms/sit.py:225) against SciPy's values over the middle, both tails, the far tails, the branch points and the edges."""
    import torch
    from scipy.special import ndtri
    from bayesfast_amd import _lib
    from bayesfast_amd.device import get_context, _ptr
    ctx = get_context(0)
    rng = np.random.default_rng(11)
    e2 = np.exp(-2.)
    p = np.concatenate([rng.uniform(size=20000), 10.**rng.uniform(-300, -1, 5000), 1. - 10.**rng.uniform(-16, -1, 5000),
                        np.nextafter(e2, [0., 1.]), [e2, 1. - e2, 0.5, np.exp(-32.), 5e-324, 1. - 2.**-53],
                        np.nextafter(1. - e2, [0., 1.])])
    d = ctx.tensor(p)
    out = torch.empty_like(d)
    _lib.check(ctx._lib.bfhip_ndtri(ctx.handle, p.size, _ptr(d), _ptr(out)))
    got, want = out.cpu().numpy(), ndtri(p)
    # (the device's log and sqrt may round differently in the last place: a few ulp, 4e-16 relative + 1e-16 absolute at the centre)
    np.testing.assert_allclose(got, want, rtol=2e-15, atol=2e-16)
    edge = ctx.tensor(np.array([0., 1., -0.1, 1.1, np.nan]))
    _lib.check(ctx._lib.bfhip_ndtri(ctx.handle, 5, _ptr(edge), _ptr(edge)))   # in place
    e = edge.cpu().numpy()
    assert e[0] == -np.inf and e[1] == np.inf and np.isnan(e[2:]).all()
    # the Sobol-normal draws SIT.sample takes are the reference's (utils/sobol.py:40-57: norm.ppf of the scrambled points)
    from bayesfast_amd.utils import sobol
    dev = sobol.standard_normal_device(7, 1000, ctx).cpu().numpy()
    host = sobol.multivariate_normal(np.zeros(7), np.eye(7), 1000)
    np.testing.assert_allclose(dev, host, rtol=2e-15, atol=2e-16)


@pytest.mark.gpu
def test_fastica_chunk_graph_equals_the_eager_chunks(monkeypatch):
    """The FastICA iteration's chunks replayed as ONE HIP graph (library products, the glue kernels and the polar kernel its nodes,
    kept per shape) find what the eager chunks find: same iteration count, same components bit for bit (the same kernels on the same
    data in the same order); a second fit of the same shape reuses the graph."""
    from bayesfast_amd.transforms import ica
    from bayesfast_amd.device import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(4)
    x = rng.laplace(size=(20100, 48)) @ (np.eye(48) + 0.3 * rng.normal(size=(48, 48)))    # (20100 rows: a padded last batch)
    ica._DEVICE_STATE.clear()
    before = dict(ica.GRAPH_STATS)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        comp_g, mean_g, n_g = ica.fastica_device(x, random_state=7, tol=1e-9, ctx=ctx)
        assert ica.GRAPH_STATS['captured'] == before['captured'] + 1 and ica.GRAPH_STATS['failed'] == before['failed']
        assert ica.GRAPH_STATS['replayed'] > before['replayed'] and n_g > ica._CHUNK
        comp_2, _, n_2 = ica.fastica_device(x, random_state=7, tol=1e-9, ctx=ctx)
    assert ica.GRAPH_STATS['captured'] == before['captured'] + 1 and n_2 == n_g
    np.testing.assert_array_equal(comp_2, comp_g)
    ica._DEVICE_STATE.clear()
    monkeypatch.setitem(ica.GRAPH_STATS, 'failed', 99)    # (no capture: eager chunks)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        comp_e, mean_e, n_e = ica.fastica_device(x, random_state=7, tol=1e-9, ctx=ctx)
    assert n_e == n_g
    np.testing.assert_array_equal(comp_g, comp_e)
    ica._DEVICE_STATE.clear()
    # ... and scikit-learn's own fit of the same data: the same components (to the fixed point's tolerance)
    from sklearn.decomposition import FastICA
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        comp_d, _, n_d = ica.fastica_device(x, random_state=7, ctx=ctx)
        sk = FastICA(random_state=7, whiten='unit-variance').fit(x)
    assert n_d == sk.n_iter_
    np.testing.assert_allclose(comp_d, sk.components_, rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['laplace', 'mixture', 'weighted', 'options'])
def test_device_spline_build_is_the_host_construction(kind):
    """bfhip_spline_build (one workgroup per coordinate runs cubic_spline's construction, utils/cubic.py:19-260, cdf sums included)
    against the host construction (utils/spline.py, pinned to the reference's fixtures by test_spline_construction_matches_reference)
    fed by bfhip_kde_cdf: the same knots bit for bit (percentile / linspace arithmetic as NumPy's), values and coefficient rows to the
    cdf sums' summation order."""
    import torch
    from bayesfast_amd.transforms import SIT
    from bayesfast_amd.utils.spline import GaussianizingSpline
    from bayesfast_amd.device import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(21)
    n, d = 30000, 12
    if kind == 'mixture':     # gaps in the data: wide-gap knots, non-monotone rounds, flat stretches
        y = np.where(rng.uniform(size=(n, d)) < 0.3, rng.normal(-6., 0.2, size=(n, d)), rng.normal(5., 1.5, size=(n, d)))
        y[:, 3] = np.round(y[:, 3], 1)        # ties: duplicate percentiles
    else:
        y = rng.laplace(size=(n, d)) * np.linspace(0.5, 3., d)
    sit = SIT(n_iter=1, random_generator=1)
    if kind == 'options':
        sit.cubic_options = dict(bins=60, edge_bins=2, edge_points=7, max_width=3, split=3, max_add=3)
    sit._weights = rng.uniform(0.2, 1., size=n) if kind == 'weighted' else np.ones(n) / n
    yd = ctx.tensor(y)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        table = sit._gaussianize(yd)
        dev = table.splines
        # the host construction on the same inputs
        yT = yd.T.contiguous()
        wn = sit._weights / np.sum(sit._weights)
        w = ctx.tensor(wn)
        neff = 1. / np.sum(wn**2)
        mean = (yT * w).sum(1) / w.sum()
        var = (((yT - mean[:, None])**2) * w).sum(1) / w.sum() / (1. - float(np.sum(wn**2)))
        h = (torch.sqrt(var) * (neff**(-1. / 5)) * sit.bw_factor).contiguous()
        ys = torch.sort(yT, dim=1).values.cpu().numpy()
        host = GaussianizingSpline.build_many(ys, SIT._batch_fun(ctx, yT, w, h, ys), presorted=True, **sit.cubic_options)
    sizes = set()
    for a, b in zip(dev, host):
        np.testing.assert_array_equal(a.x, b.x)
        np.testing.assert_allclose(a.y, b.y, rtol=0, atol=1e-11)
        # (the cubic rows divide value differences of 1e-14 by knot spacings down to 1e-5, up to three times: compare the cubics
        # and their slopes inside every interval, not the leading coefficients)
        np.testing.assert_allclose(a.c[:, 2:], b.c[:, 2:], rtol=1e-7, atol=1e-9)
        for f in (0.25, 0.5, 0.9):
            t = f * np.diff(a.x)
            val = lambda c: ((c[1:-1, 0] * t + c[1:-1, 1]) * t + c[1:-1, 2]) * t + c[1:-1, 3]
            der = lambda c: (3 * c[1:-1, 0] * t + 2 * c[1:-1, 1]) * t + c[1:-1, 2]
            np.testing.assert_allclose(val(a.c), val(b.c), rtol=0, atol=1e-11)
            np.testing.assert_allclose(der(a.c), der(b.c), rtol=1e-7, atol=1e-9)
        sizes.add(a.x.size)
    assert len(sizes) > 1 or kind == 'options'    # (knots were added somewhere: the rounds ran)
    print(kind, sorted(sizes))


@pytest.mark.gpu
@pytest.mark.parametrize('d', [5, 16, 48, 100, 128, 200, 256, 300])
def test_polar_ns_is_the_orthogonal_polar_factor(d):
    """bfhip_polar_ns (FastICA's symmetric decorrelation, scikit-learn's _sym_decorrelation as SIT calls it, transforms/sit.py:235-244)
    against the SVD's polar factor; its forms (X in LDS, a workgroup per row block, one grid barrier per step / the same with operands from L2 / a tile per
    wave with two barriers) take bit-identical steps; residual reported, early stop, n_iter = 0."""
    import torch
    from bayesfast_amd import _lib
    from bayesfast_amd._lib import debug_set
    from bayesfast_amd.device import get_context, _ptr
    ctx = get_context(0)
    rng = np.random.default_rng(d)
    A = rng.normal(size=(d, d)) * 0.03 + np.diag(rng.uniform(0.05, 2., size=d))
    a = ctx.tensor(A)
    work = torch.empty(2 * d * d + 80, dtype=torch.float64, device=a.device)

    def run(n_iter, tiles=0):
        debug_set('polar_tiles', tiles)
        x = torch.empty_like(a)
        _lib.check(ctx._lib.bfhip_polar_ns(ctx.handle, d, _ptr(a), _ptr(x), n_iter, _ptr(work), _ptr(work[-1:])))
        return x.cpu().numpy(), float(work[-1])

    u, s, vt = np.linalg.svd(A)
    x, res = run(60)
    assert res < 1e-13
    np.testing.assert_allclose(x, u @ vt, rtol=0, atol=5e-13 * s.max() / s.min())
    np.testing.assert_allclose(x @ x.T, np.eye(d), rtol=0, atol=1e-13)
    x3, res3 = run(3)
    assert res3 > 1e-13                                      # three steps are not enough for this matrix ...
    np.testing.assert_allclose(np.abs(x3 @ x3.T - np.eye(d)).max(), res3, rtol=1e-9)   # ... and the residual says how far they got
    for form in ((1, 2) if d <= 128 else (1,) if d <= 256 else ()):    # 1: a tile per wave, 2: row blocks with operands from L2
        x3t, res3t = run(3, tiles=form)
        np.testing.assert_array_equal(x3, x3t)
        assert res3 == res3t
        xt, rest = run(60, tiles=form)
        np.testing.assert_allclose(xt, x, rtol=0, atol=2e-13)     # (converged runs end a step apart: rounding noise of orthogonal iterates)
    x0, res0 = run(0)
    scale = np.sqrt(np.abs(A).sum(0).max() * np.abs(A).sum(1).max())
    np.testing.assert_allclose(x0, A / scale, rtol=1e-15)
    debug_set('polar_tiles', 0)


@pytest.mark.gpu
def test_device_spline_build_edge_columns_follow_the_host_construction(monkeypatch):
    """Columns the construction struggles with -- five distinct values (too few distinct percentile knots: the kernel gives the
    coordinate up and the host construction builds it, degenerate as in the reference), values rounded to one decimal (ties among the
    percentiles), two far clusters (wide-gap knots) -- next to well-behaved ones: the device route returns what the host route returns."""
    from bayesfast_amd.transforms import SIT
    from bayesfast_amd.device import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(0)
    n, d = 20000, 6
    y = rng.normal(size=(n, d))
    y[:, 1] = rng.integers(0, 5, size=n)
    y[:, 2] = np.round(y[:, 2], 1)
    y[:, 3] = np.where(rng.uniform(size=n) < 0.5, -50., 50.) + 0.01 * rng.normal(size=n)
    sit = SIT(n_iter=1, random_generator=1)
    sit._weights = np.ones(n) / n
    yd = ctx.tensor(y)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dev = sit._gaussianize(yd).splines
        monkeypatch.setattr(SIT, '_build_on_device', lambda self, *a: None)
        host = sit._gaussianize(yd).splines
    assert [s.x.size for s in dev] == [s.x.size for s in host]
    assert dev[1].x.size > 512          # (the discrete column: built by the host construction on both routes)
    for a, b in zip(dev, host):
        np.testing.assert_array_equal(a.x, b.x)
        np.testing.assert_allclose(a.y, b.y, rtol=0, atol=1e-10, equal_nan=True)


@pytest.mark.gpu
def test_sit_fit_on_a_device_tensor_is_the_fit_on_the_array():
    """SIT.fit takes the data as a device tensor too (GBS's device route hands it the samples where sample() left them): the same
    model as from the host array -- rotations, log-determinants, logq -- and the same argument checks."""
    import torch
    from bayesfast_amd.transforms import SIT
    from bayesfast_amd.device import get_context
    ctx = get_context(0)
    rng = np.random.default_rng(12)
    x = rng.laplace(size=(6000, 5)) @ (np.eye(5) + 0.3 * rng.normal(size=(5, 5)))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a = SIT(n_iter=2, random_generator=4)
        a.fit(x)
        b = SIT(n_iter=2, random_generator=4)
        b.fit(ctx.tensor(x).reshape(3, 2000, 5))          # (chain, iteration, d) as a TraceTuple holds it
    np.testing.assert_array_equal(a._A, b._A)
    np.testing.assert_array_equal(a._logdetA, b._logdetA)
    pts = rng.normal(size=(50, 5))
    np.testing.assert_allclose(a.logq(pts), b.logq(pts), rtol=0, atol=1e-12)
    np.testing.assert_allclose(b._logq_device(ctx.tensor(pts)).cpu().numpy(), b.logq(pts), rtol=0, atol=1e-11)
    xs = b._sample_device(300)
    assert isinstance(xs, torch.Tensor) and xs.shape == (300, 5) and bool(torch.isfinite(xs).all())
    np.testing.assert_allclose(xs.cpu().numpy(), b.sample(300)[0], rtol=0, atol=1e-12)     # (the same Sobol points)
    with pytest.raises(ValueError):
        SIT().fit(ctx.tensor(x[:, :1]))                   # one variable: nothing to rotate
    with pytest.raises(ValueError):
        SIT().fit(ctx.tensor(x[:0]))
    c = SIT(n_iter=1, random_generator=4, mvn_generator=lambda m, c_, n: rng.normal(size=(n, m.size)))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        c.fit(x)
    assert c._sample_device(10) is None                   # a user's generator: the host route draws
