"""The drop-in seam (bayesfast_amd/integrate.py) under the REFERENCE's own code.

Container-only: these tests import the reference package (built from /root/reference into a scratch directory by
tests/golden/make_golden.py's recipe) and are skipped wherever it does not exist -- the reference never travels to the GPU
box.  There is no GPU here either, so the device entry points (DeviceChains, DeviceDensity, the device fit) are swapped, IN
THESE TESTS ONLY, for stand-ins backed by the CPU oracle (tests/helpers/oracle_standin.py); everything else -- the seam's
subclasses, adapters, trace conversion, sample(), TraceTuple -- is the shipped code, driven by the reference's Recipe,
Density, _get_step_size and _get_metric.  The GPU side of the same workload is tests/test_gpu_recipe.py."""
import copy
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'helpers'))
REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'bayesfast')), reason='needs the reference at /root/reference')


@pytest.fixture(scope='module')
def bf():
    import make_golden
    return make_golden.prepare_reference(REF, os.environ.get('BF_REF_WORK', '/tmp/bfref'))


@pytest.fixture()
def seam(bf, monkeypatch):
    import oracle_standin
    from bayesfast_amd import integrate
    oracle_standin.install(monkeypatch)
    unpatch = integrate.patch(bf)
    yield integrate.reference_classes(bf)
    unpatch()


def test_subclasses_pass_the_reference_isinstance_gates(bf, seam):
    """core/recipe.py:52-61 (steps), core/density.py:306-310 (Density.surrogate_list), deepcopy / dill (core/recipe.py:822,1163)."""
    import dill
    pm = seam.PolyModel('quadratic', input_size=3, output_size=1, input_vars='x', output_vars='y')
    assert isinstance(pm, bf.core.module.Surrogate) and isinstance(pm, bf.modules.PolyModel)
    assert bf.modules.PolyModel is seam.PolyModel and bf.core.recipe.sample is not None  # patched names
    step = bf.recipe.SampleStep(surrogate_list=pm, alpha_n=2)           # raises ValueError for anything that is not a Surrogate
    assert step.n_eval == 2 * pm.n_param
    like = seam.GaussianLikelihood(1., 2., input_vars='y', output_vars='logp')
    assert isinstance(like, bf.core.module.Module)
    den = bf.Density(module_list=[bf.Module(fun=lambda x: np.sum(x**2, -1, keepdims=True), input_vars='x', output_vars='y'), like],
                     input_shapes=[3], input_vars='x', density_name='logp')
    den.surrogate_list = pm                                             # raises ValueError likewise
    # the likelihood module is the analytic function it claims to be
    np.testing.assert_allclose(den.logp(np.array([1., 2., 2.]), use_surrogate=False), -0.5 * 2. * (9. - 1.)**2)
    for clone in (copy.deepcopy(pm), dill.loads(dill.dumps(pm))):
        assert isinstance(clone, bf.core.module.Surrogate) and clone.n_param == pm.n_param
    with pytest.raises(ValueError):
        bf.recipe.SampleStep(surrogate_list=[object()])


def test_polymodel_fit_on_the_seam_equals_the_reference_fit(bf, seam):
    """seam.PolyModel.fit (coefficients from the fit behind integrate._device_fit, bound statistics by the reference's own
    _set_bound) against the reference class's fit on the same data: same coefficients, bound, evaluation."""
    ref_cls = [c for c in seam.PolyModel.__mro__ if c.__module__.endswith('modules.poly') and c is not seam.PolyModel][0]
    rng = np.random.default_rng(5)
    x = rng.normal(size=(60, 3))
    y = (x[:, 0] * x[:, 1] - 0.3 * x[:, 2]**2 + x[:, 0] + 0.01 * rng.normal(size=60))[:, None]
    a = seam.PolyModel('quadratic', input_size=3, output_size=1)
    b = ref_cls('quadratic', input_size=3, output_size=1)
    a.fit(x, y, y[:, 0])
    b.fit(x, y, y[:, 0])
    iu = np.triu_indices(3)
    np.testing.assert_allclose(a.configs[0]._coef, b.configs[0]._coef, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(a.configs[1]._coef[0][iu], b.configs[1]._coef[0][iu], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose([a._alpha, a._f_mu[0]], [b._alpha, b._f_mu[0]], rtol=1e-9)
    for xt in (0.1 * np.ones(3), 30. * np.ones(3)):   # inside / outside the bound
        fa, ja = a.fun_and_jac(xt)
        fb, jb = b.fun_and_jac(xt)
        np.testing.assert_allclose(fa[0], fb[0], rtol=1e-9)
        np.testing.assert_allclose(ja[0], jb[0], rtol=1e-8, atol=1e-10)


def test_reference_recipe_runs_config1_on_the_seam(bf, seam):
    """BASELINE config 1: the reference's Recipe.run() -- OptimizeStep with its sampling, ten SampleSteps, PostStep -- with
    the seam's PolyModel / GaussianLikelihood in its Density and the seam's sample() behind core/recipe.py:979-981,1160-1173.
    It must reach `finished`, print the optimiser trace of examples/2d-donut.ipynb:110-111, and land on the ring the
    reference's own run lands on (tests/golden/recipe.npz, fixture vi), within the T2 bounds of SURVEY section 8c."""
    import donut
    z = np.load(os.path.join(HERE, 'golden', 'recipe.npz'))
    rec = donut.build_recipe(bf, poly_model=seam.PolyModel,
                             likelihood=seam.GaussianLikelihood(donut.A, 2. / donut.B, input_vars='m', output_vars='logp'))
    rec.run()
    rt = rec.recipe_trace
    assert tuple(rt.finished) == (True, True, True)
    opt = rt.results.optimize
    # the optimiser's log is deterministic given the fit (Laplace + the reference's Sobol points): equal to the fixture's
    np.testing.assert_allclose([r.f_max.logp_trans for r in opt[:-1]], z['opt.logp_trans'], rtol=1e-6)
    np.testing.assert_allclose([r.f_max.logq_trans for r in opt[:-1]], z['opt.logq_trans'], rtol=1e-6)
    assert abs(z['opt.logp'][0] - (-1.870)) < 5e-4 and abs(z['opt.logp'][1] - (-1.497)) < 5e-4   # the notebook's printed lines
    steps = rt.results.sample
    assert len(steps) == 10
    ring = np.array([donut.ring_statistics(r.samples) for r in steps])
    # every step's result is a reference TraceTuple of reference NTraces, accepted by the reference's warm-start helpers
    tt = steps[-1].sample_trace
    assert isinstance(tt, bf.samplers.TraceTuple) and all(isinstance(t, bf.samplers.NTrace) for t in tt)
    ss = bf.samplers._get_step_size(tt)
    assert 0.01 < ss < 2.
    m = bf.samplers._get_metric(tt, 'diag')
    assert m.shape == (2,) and np.all(m > 1.)
    assert bf.samplers._get_metric(tt, 'full', from_samples=False).shape == (2, 2)
    assert tt.n_call == sum(t.n_call for t in tt)
    # T2: the ring of the last four steps against the reference's last four (radius 5.02 +- 0.49 there)
    want, got = z['ring'][-4:], ring[-4:]
    assert abs(got[:, 0].mean() - want[:, 0].mean()) < 0.06, (got[:, 0], want[:, 0])
    assert abs(got[:, 1].mean() - want[:, 1].mean()) < 0.05, (got[:, 1], want[:, 1])
    assert got[:, 2].max() < 0.15            # the whole ring is covered (mean resultant length of the angle)
    # ... and the progress towards it: the first step sits on the arc the optimiser found, as the reference's does
    assert abs(ring[0, 0] - z['ring'][0, 0]) < 0.15 and ring[0, 2] > 0.95
    res = rec.get()
    assert res.samples.shape == (2000, 2) and res.n_call == int(z['n_call'])


def test_sample_on_the_seam_takes_reference_traces_and_continues(bf, seam):
    """integrate.sample: dict / NTrace in, reference TraceTuple out; a second call with that tuple continues the chains
    (core/sample.py:94-96); what the device path does not cover is refused, or left to the reference with fallback=True."""
    from bayesfast_amd import integrate
    pm = seam.PolyModel('quadratic', input_size=2, output_size=1, input_vars='x', output_vars='logp')
    den = bf.Density(module_list=[bf.Module(fun=lambda x: -0.5 * np.sum(x**2, -1, keepdims=True), input_vars='x', output_vars='logp')],
                     input_shapes=[2], input_vars='x', density_name='logp', surrogate_list=pm)
    x = np.random.default_rng(1).normal(size=(40, 2)) * 2.
    pm.fit(x, -0.5 * np.sum(x**2, -1, keepdims=True), -0.5 * np.sum(x**2, -1))
    den.use_surrogate = True
    trace = bf.samplers.NTrace(n_chain=3, n_iter=80, n_warmup=50, random_generator=4)
    tt = bf.sample(den, trace, n_run=60, verbose=False)          # bf.sample is the patched entry point
    assert isinstance(tt, bf.samplers.TraceTuple) and tt.i_iter == 60 and not tt.finished
    tt2 = bf.sample(den, tt, verbose=False)
    assert tt2.i_iter == 80 and tt2.finished
    assert np.array_equal(tt2.samples[:, :60], tt.samples)
    assert tt2.get().shape == (3 * 30, 2) and tt2.get(return_type='logp', flatten=False).shape == (3, 30)
    assert np.isfinite(bf.samplers._get_step_size(tt2))
    # the per-chain traces carry a CONSISTENT adapted metric (random() reads _inv_std, velocity() reads _var:
    # samplers/hmc_utils/metrics.py:60-86), one object per chain
    for t in tt2:
        np.testing.assert_allclose(t._metric._inv_std**-2, t._metric._var, rtol=1e-14)
        np.testing.assert_allclose(t._metric._std**2, t._metric._var, rtol=1e-14)
    assert len({id(t._metric) for t in tt2}) == 3 and len({id(t._step_size) for t in tt2}) == 3
    # a template that already carries INSTANCES (a warm start built by hand): every chain gets its own copy, the caller's
    # objects are not written to
    from bayesfast.samplers.hmc_utils.metrics import QuadMetricDiagAdapt
    from bayesfast.samplers.hmc_utils.step_size import DualAverageAdaptation
    met = QuadMetricDiagAdapt(2, np.zeros(2), np.ones(2), 10)
    ssz = DualAverageAdaptation(0.7, 0.8, 0.05, 0.75, 10., True)
    tr3 = bf.samplers.NTrace(n_chain=3, n_iter=40, n_warmup=30, random_generator=5, metric=met, step_size=ssz)
    tt3 = bf.sample(den, tr3, verbose=False)
    assert len({id(t._metric) for t in tt3}) == 3 and all(t._metric is not met for t in tt3)
    assert len({id(t._step_size) for t in tt3}) == 3 and all(t._step_size is not ssz for t in tt3)
    np.testing.assert_array_equal(met._var, np.ones(2))
    assert len({float(t._step_size._log_step) for t in tt3}) == 3
    den.use_surrogate = False
    with pytest.raises(NotImplementedError):
        integrate.sample(den, {'n_chain': 2, 'n_iter': 20, 'n_warmup': 10})


def test_des_shaped_pipeline_on_the_seam(bf, seam):
    """SURVEY 8f-1 through the seam: a reference ``Density`` with module_list = [model -> m (22 outputs), GaussianLikelihood of m,
    GaussianPrior(like, x)] and the seam's multi-output PolyModel as the surrogate of the first module
    (examples/des-y1-w-cosmosis.ipynb cells 12-18 in small).  The seam's analytic modules are real ``bayesfast.Module``s: the
    reference evaluates the pipeline with them (use_surrogate=True), ``as_surrogate_density`` turns it into the pipeline
    density the device samples, both give the same logp / grad, and ``bf.sample`` runs NUTS on it."""
    from bayesfast_amd import integrate
    from bayesfast_amd.core.density import Chi2PipelineDensity
    from oracle import oracle as orc
    rng = np.random.default_rng(91)
    d, m = 6, 22
    lo, hi = -1. - rng.uniform(size=d), 1.5 + rng.uniform(size=d)
    para_range = np.stack([lo, hi], 1)
    nonlinear = np.array([0, 2, 3])
    W1 = rng.normal(size=(m, d)) * 0.6
    W2 = rng.normal(size=(m, 3, 3)) * 0.25
    dvec = rng.normal(size=m) * 0.3

    def model(x):
        z = x[nonlinear]
        return W1 @ x + np.einsum('ojk,j,k->o', W2, z, z) + 0.05 * np.sin(2. * z[0])

    mod0 = bf.Module(fun=model, input_vars='x', output_vars='m')
    like = seam.GaussianLikelihood(dvec, logp0=-1.5, input_vars='m', output_vars='like')            # identity precision
    post = seam.GaussianPrior(d, indices=[1, 4, 5], mu=[0.1, -0.2, 0.05], sigma=[0.3, 0.4, 0.25], c0=0.7, input_vars=['like', 'x'],
                              output_vars='logp')
    su = seam.PolyModel([bf.modules.PolyConfig('linear'), bf.modules.PolyConfig('quadratic', input_mask=nonlinear)], input_size=d,
                        output_size=m, input_vars='x', output_vars='m', input_scales=para_range)
    den = bf.Density(density_name='logp', module_list=[mod0, like, post], surrogate_list=[su], input_vars='x', input_shapes=d,
                     input_scales=para_range, hard_bounds=True)
    xf = lo + (hi - lo) * (0.5 + 0.2 * rng.normal(size=(4 * int(su.n_param), d))).clip(0.03, 0.97)
    den.fit([den.fun(x, original_space=True, use_surrogate=False) for x in xf])                     # the device fit (stand-in here)
    den.use_surrogate = True
    ours = integrate.as_surrogate_density(den)
    assert isinstance(ours, Chi2PipelineDensity) and ours.input_size == d
    spec = ours.spec()
    assert spec['chi2']['prec_diag'].shape == (m,) and spec['prior']['prec_diag'][0] == 0. and spec['prior']['prec_diag'][1] > 0.
    xo = lo + (hi - lo) * rng.uniform(0.05, 0.95, size=(12, d))
    xt = np.array([den.from_original(x) for x in xo])
    for sp, pts in ((True, xo), (False, xt)):
        ref = [den.logp_and_grad(x, original_space=sp, use_surrogate=True) for x in pts]
        lp, g = orc.logp_and_grad(spec, pts, original_space=sp)
        np.testing.assert_allclose(lp, [r[0] for r in ref], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(g, [r[1] for r in ref], rtol=1e-10, atol=1e-10)
    tt = bf.sample(den, bf.samplers.NTrace(n_chain=4, n_iter=60, n_warmup=40, x_0=xo[:4], random_generator=3), verbose=False)
    assert isinstance(tt, bf.samplers.TraceTuple) and tt.samples.shape == (4, 60, d)
    s = tt.get()
    assert np.all(s > lo) and np.all(s < hi) and np.all(np.isfinite(tt.get(return_type='logp')))
    # a full precision matrix and a likelihood the kernel does not know
    A = rng.normal(size=(m, m)) * 0.1
    like2 = seam.GaussianLikelihood(dvec, prec=np.eye(m) + A @ A.T, input_vars='m', output_vars='logp')
    den2 = bf.Density(density_name='logp', module_list=[mod0, like2], surrogate_list=[su], input_vars='x', input_shapes=d)
    den2.use_surrogate = True
    sp2 = integrate.as_surrogate_density(den2).spec()
    assert sp2['chi2']['prec'].shape == (m, m) and sp2['prior'] is None
    ref = [den2.logp_and_grad(x, original_space=True, use_surrogate=True) for x in xo[:5]]
    lp, g = orc.logp_and_grad(sp2, xo[:5], original_space=True)
    np.testing.assert_allclose(lp, [r[0] for r in ref], rtol=1e-11)
    np.testing.assert_allclose(g, [r[1] for r in ref], rtol=1e-9, atol=1e-9)
    den3 = bf.Density(density_name='logp', module_list=[mod0, bf.Module(fun=lambda mm: -0.5 * np.sum(mm**2, keepdims=True), input_vars='m',
                                                                           output_vars='logp')],
                      surrogate_list=[su], input_vars='x', input_shapes=d)
    den3.use_surrogate = True
    with pytest.raises(NotImplementedError):
        integrate.as_surrogate_density(den3)


def test_tempered_nuts_on_the_seam(bf, seam):
    """core/sample.py:83-84: ``sample(density, TNTrace(density_base=..., logxi=...))`` -- the reference's own TNTrace, with the
    seam's GaussianBaseDensity (a real DensityLite, so the trace's type check passes) as the base density; the result is a
    reference TraceTuple whose chains are TNTraces carrying u and the weights (samplers/sample_trace.py:540-583)."""
    pm = seam.PolyModel('quadratic', input_size=3, output_size=1, input_vars='x', output_vars='logp')
    den = bf.Density(module_list=[bf.Module(fun=lambda x: -0.5 * np.sum(x**2, -1, keepdims=True), input_vars='x', output_vars='logp')],
                     input_shapes=[3], input_vars='x', density_name='logp', surrogate_list=pm)
    x = np.random.default_rng(1).normal(size=(60, 3)) * 2.
    pm.fit(x, -0.5 * np.sum(x**2, -1, keepdims=True), -0.5 * np.sum(x**2, -1))
    den.use_surrogate = True
    base = seam.GaussianBaseDensity(np.zeros(3), 1.5 * np.eye(3))
    np.testing.assert_allclose(base.logp(np.ones(3)), -0.5 * 3 / 1.5 - 1.5 * np.log(2 * np.pi * 1.5), rtol=1e-12)
    tr = bf.samplers.TNTrace(density_base=base, logxi=0.3, n_chain=3, n_iter=50, n_warmup=30, random_generator=8)
    tt = bf.sample(den, tr, verbose=False)
    assert isinstance(tt, bf.samplers.TraceTuple) and tt.i_iter == 50 and tt.samples.shape == (3, 50, 3)
    chains = list(tt)
    assert all(isinstance(t, bf.samplers.TNTrace) for t in chains)
    assert chains[0].u.shape == (50,) and chains[0].weights.shape == (50,) and np.all(chains[0].weights > 0)
    assert len(chains[0].stats._tree_size) == 50 and np.isfinite(bf.samplers._get_step_size(tt))
    with pytest.raises(NotImplementedError):   # an arbitrary Python base density cannot run inside the kernel
        bf.sample(den, bf.samplers.TNTrace(density_base=bf.DensityLite(logp=lambda x: -0.5 * np.sum(x**2), input_size=3), n_chain=2,
                                           n_iter=10, n_warmup=5), verbose=False)


def test_reference_recipe_runs_the_des_shaped_pipeline_on_the_seam(bf, seam):
    """The reference's own ``Recipe.run()`` in the shape of examples/des-y1-w-cosmosis.ipynb cells 14-21 -- OptimizeStep with a
    LINEAR multi-output surrogate, two SampleSteps with the block-quadratic one (``reuse_samples=1``), PostStep with truncated
    importance sampling -- on a Density = [model (22 outputs), seam GaussianLikelihood, seam GaussianPrior] with input scales
    and hard bounds: every fit goes through the device fit, every ``sample`` through the pipeline density of the seam.  It must
    reach ``finished``, count its true-model calls as the notebook's does (a few hundred), and put the posterior on the truth."""
    rng = np.random.default_rng(91)
    d, m = 6, 22
    lo, hi = -1. - rng.uniform(size=d), 1.5 + rng.uniform(size=d)
    para_range = np.stack([lo, hi], 1)
    nonlinear = np.array([0, 2, 3])
    W1 = rng.normal(size=(m, d)) * 1.5
    W2 = rng.normal(size=(m, 3, 3)) * 0.5
    x_true = lo + (hi - lo) * rng.uniform(0.4, 0.6, size=d)

    def model(x):
        z = x[nonlinear]
        return W1 @ x + np.einsum('ojk,j,k->o', W2, z, z) + 0.05 * np.sin(2. * z[0])

    dvec = model(x_true) + rng.normal(size=m)
    bf.utils.random.set_generator(27)
    bf.utils.parallel.set_backend(4)
    mod0 = bf.Module(fun=model, input_vars='x', output_vars='m')
    like = seam.GaussianLikelihood(dvec, logp0=-1.5, input_vars='m', output_vars='like')
    post = seam.GaussianPrior(d, indices=[1, 4, 5], mu=x_true[[1, 4, 5]], sigma=[0.3, 0.4, 0.25], c0=0.7, input_vars=['like', 'x'],
                              output_vars='logp')
    den = bf.Density(density_name='logp', module_list=[mod0, like, post], input_vars='x', input_shapes=d, input_scales=para_range,
                     hard_bounds=True)
    su0 = seam.PolyModel('linear', input_size=d, output_size=m, input_vars='x', output_vars='m', input_scales=para_range)
    su1 = seam.PolyModel([bf.modules.PolyConfig('linear'), bf.modules.PolyConfig('quadratic', input_mask=nonlinear)], input_size=d,
                         output_size=m, input_vars='x', output_vars='m', input_scales=para_range)
    tr = {'n_chain': 4, 'n_iter': 600, 'n_warmup': 300}
    x_0 = bf.utils.sobol.multivariate_normal(x_true, np.diag(((hi - lo) / 50)**2), 40)
    rec = bf.recipe.Recipe(density=den, optimize=bf.recipe.OptimizeStep(surrogate_list=su0, alpha_n=2, x_0=x_0, sample_trace=dict(tr)),
                           sample=[bf.recipe.SampleStep(surrogate_list=su1, alpha_n=2, reuse_samples=1, sample_trace=dict(tr)),
                                   bf.recipe.SampleStep(surrogate_list=su1, alpha_n=2, reuse_samples=1, sample_trace=dict(tr))],
                           post=bf.recipe.PostStep(n_is=400, k_trunc=0.25))
    rec.run()
    assert tuple(rec.recipe_trace.finished) == (True, True, True)
    res = rec.get()
    assert res.samples.shape == (400, d) and np.isfinite(res.weights_trunc).all()
    assert 300 < res.n_call < 900                      # 40 + the fit points of three steps + 400 importance-sampling calls
    tt = rec.recipe_trace.results.sample[-1].sample_trace
    assert isinstance(tt, bf.samplers.TraceTuple) and tt.samples.shape == (4, 600, d)
    s = res.samples
    assert np.all(np.abs(s.mean(0) - x_true) < 4. * s.std(0)), (s.mean(0) - x_true) / s.std(0)
    assert np.all(s > lo) and np.all(s < hi)
