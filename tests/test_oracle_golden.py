"""Pins the CPU oracle (oracle/bf_oracle.c + oracle/oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc
from specio import rebuild_spec, rebuild_poly

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


@pytest.fixture(scope='module')
def kern():
    return np.load(os.path.join(G, 'poly_kernels.npz'))


@pytest.mark.parametrize('n', [2, 3, 5, 8])
@pytest.mark.parametrize('order', ['quadratic', 'cubic_2', 'cubic_3'])
def test_poly_kernels(kern, n, order):
    """_poly.pyx value / Jacobian / design block / coefficient scatter, bitwise or to 1 ulp."""
    L = orc.lib()
    k = 'n%d.%s.' % (n, order)
    x = np.ascontiguousarray(kern['n%d.x' % n])
    xs = np.ascontiguousarray(kern['n%d.xs' % n])
    a, coef = kern[k + 'a'], np.ascontiguousarray(kern[k + 'coef'])
    m = a.shape[0]
    # scatter
    shape = coef.shape[1:]
    for i in range(m):
        c = np.zeros(shape)
        getattr(L, 'bfo_set_' + order)(_p(np.ascontiguousarray(a[i])), _p(c), n)
        assert np.array_equal(c, coef[i])
    f = np.empty(m)
    j = np.empty((m, n))
    getattr(L, 'bfo_%s_f' % order)(_p(x), _p(coef), _p(f), m, n)
    getattr(L, 'bfo_%s_j' % order)(_p(x), _p(coef), _p(j), m, n)
    np.testing.assert_allclose(f, kern[k + 'f'], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(j, kern[k + 'j'], rtol=1e-14, atol=1e-15)
    A = np.empty_like(kern[k + 'lsq'])
    if A.size:
        getattr(L, 'bfo_lsq_' + order)(_p(xs), _p(A), xs.shape[0], n)
    assert np.array_equal(A, kern[k + 'lsq'])


def test_constraint():
    z = np.load(os.path.join(G, 'constraint.npz'))
    L = orc.lib()
    ranges, hb = np.ascontiguousarray(z['ranges']), np.ascontiguousarray(z['hard_bounds'])
    n = ranges.shape[0]
    hbp = hb.ctypes.data_as(C.POINTER(C.c_uint8))
    for i, xt in enumerate(z['x_trans']):
        xt = np.ascontiguousarray(xt)
        for nm in ('f', 'j', 'jj'):
            o = np.empty(n)
            getattr(L, 'bfo_to_original_' + nm)(_p(xt), _p(ranges), _p(o), hbp, n)
            np.testing.assert_allclose(o, z['to_' + nm][i], rtol=1e-14, atol=0)
        xo = np.ascontiguousarray(z['to_f'][i])
        for nm in ('f', 'j', 'jj'):
            o = np.empty(n)
            assert getattr(L, 'bfo_from_original_' + nm)(_p(xo), _p(ranges), _p(o), hbp, n) == 0
            np.testing.assert_allclose(o, z['from_' + nm][i], rtol=1e-13, atol=0)
    # out of bound is reported (the reference raises ValueError, _constraint.pyx:27-28)
    bad = np.ascontiguousarray(z['to_f'][0]).copy()
    bad[0] = ranges[0, 1] + 1.
    assert L.bfo_from_original_f(_p(bad), _p(ranges), _p(np.empty(n)), hbp, n) == 1


def _independent(order, coef):
    """The entries the kernels read.  PolyConfig._set fills an np.empty block (modules/poly.py:146), so the
    other entries of the reference's dense coefficient arrays are uninitialised memory."""
    n = coef.shape[-1]
    if order == 'quadratic':
        return coef[..., np.triu(np.ones((n, n), bool))]
    if order == 'cubic-3':
        i, j, k = np.meshgrid(*[np.arange(n)] * 3, indexing='ij')
        return coef[..., (i < j) & (j < k)]
    return coef


def test_polymodel_eval_and_fit():
    z = np.load(os.path.join(G, 'polymodel.npz'))
    poly = rebuild_poly(z)
    f, j = orc.poly_fun_and_jac(poly, z['x_eval'])
    np.testing.assert_allclose(f, z['f'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(j, z['j'], rtol=1e-11, atol=1e-12)
    # fit restatement: same lstsq, same scatter, same bound statistics
    fit = orc.poly_fit(poly, z['x_fit'], z['y_fit'], z['logp_fit'], bound_options=dict(alpha_p=float(z['alpha_p'])))
    for c0, c1 in zip(poly['configs'], fit['configs']):
        np.testing.assert_allclose(_independent(c0['order'], c1['coef']), _independent(c0['order'], c0['coef']),
                                   rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(fit['mu'], poly['mu'], rtol=1e-13)
    np.testing.assert_allclose(fit['hess'], poly['hess'], rtol=1e-10)
    np.testing.assert_allclose(fit['alpha'], poly['alpha'], rtol=1e-12)
    np.testing.assert_allclose(fit['f_mu'], poly['f_mu'], rtol=1e-9)
    assert sum(orc.a_size(c['order'], len(c['input_mask'])) for c in poly['configs']) == int(z['n_param'])
    # weighted single-output fit
    pw = rebuild_poly(z, 'w.poly.')
    fw = orc.poly_fit(pw, z['w.x_fit'], z['w.y_fit'], z['w.y_fit'][:, 0], w=z['w.w'])
    for c0, c1 in zip(pw['configs'], fw['configs']):
        np.testing.assert_allclose(_independent(c0['order'], c1['coef']), _independent(c0['order'], c0['coef']),
                                   rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(fw['alpha'], pw['alpha'], rtol=1e-12)
    with pytest.raises(ValueError):
        orc.poly_fit(pw, z['w.x_fit'][:5], z['w.y_fit'][:5])


@pytest.mark.parametrize('case', ['plain', 'decay', 'scales', 'su', 'full', 'd64'])
def test_density_logp_and_grad(case):
    z = np.load(os.path.join(G, 'density.npz'))
    spec = rebuild_spec(z, case + '.')
    lp, g = orc.logp_and_grad(spec, z[case + '.x_trans'])
    np.testing.assert_allclose(lp, z[case + '.logp_trans'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g, z[case + '.grad_trans'], rtol=1e-11, atol=1e-11)
    if case + '.x_orig' in z.files:
        lp, g = orc.logp_and_grad(spec, z[case + '.x_orig'], original_space=True)
        np.testing.assert_allclose(lp, z[case + '.logp_orig'], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(g, z[case + '.grad_orig'], rtol=1e-11, atol=1e-11)


@pytest.fixture(scope='module')
def samp():
    return np.load(os.path.join(G, 'sampler.npz'))


@pytest.mark.parametrize('name', ['full5', 'plain16', 'd64'])
def test_leapfrog(samp, name):
    spec = rebuild_spec(samp, name + '.')
    var, eps = samp[name + '.lf.var'], samp[name + '.lf.eps']
    for a, b, e in (('s0', 's1', eps[0]), ('s1', 's2', eps[1])):
        s = {f: samp['%s.lf.%s.%s' % (name, a, f)] for f in ('q', 'p', 'q_grad')}
        r = orc.leapfrog(spec, var, e, s['q'], s['p'], s['q_grad'])
        for f, k in (('q', 'q'), ('p', 'p'), ('velocity', 'v'), ('q_grad', 'grad'), ('energy', 'energy'), ('logp', 'logp')):
            np.testing.assert_allclose(r[k], samp['%s.lf.%s.%s' % (name, b, f)], rtol=1e-11, atol=1e-12)


def _replay_nuts(samp, name, c, **kw):
    spec = rebuild_spec(samp, name + '.')
    k = '%s.nuts%d.' % (name, c)
    x0 = samp[name + '.x0'][c]
    n_iter = samp[k + 'samples'].shape[0]
    n_warmup = int(np.sum(samp[k + 'warmup']))
    chain = orc.Chain(x0, **kw)
    rng = orc.make_rng('replay', normals=samp[k + 'normals'], uniforms=samp[k + 'uniforms'])
    mc = kw.pop('max_change', None)
    return orc.nuts_run(spec, chain, rng, n_iter, n_warmup), chain, rng, k


@pytest.mark.parametrize('name,c', [('plain16', 0), ('plain16', 1), ('d64', 0), ('d64', 1), ('full5', 0),
                                     ('full5', 1), ('full5', 2)])
def test_nuts_replay(samp, name, c):
    """T1: feeding the reference's logged draws to the restatement reproduces its trajectory."""
    (samples, st), chain, rng, k = _replay_nuts(samp, name, c)
    for f in ('tree_depth', 'tree_size', 'diverging', 'warmup'):
        assert np.array_equal(st[f], samp[k + f]), f
    # every logged draw consumed, none missing: the draw ORDER of the recursion is reproduced
    assert rng[0].i_uniform == samp[k + 'uniforms'].size and rng[0].i_normal == samp[k + 'normals'].size
    np.testing.assert_allclose(samples, samp[k + 'samples'], rtol=1e-9, atol=1e-9)
    for f in ('logp', 'energy', 'mean_tree_accept', 'step_size', 'step_size_bar', 'energy_change', 'max_energy_change'):
        np.testing.assert_allclose(st[f], samp[k + f], rtol=1e-7, atol=1e-7, err_msg=f)
    np.testing.assert_allclose(chain.vec('var'), samp[k + 'final_var'], rtol=1e-9)
    # n_call accounting of NTrace (samplers/sample_trace.py:529-530)
    assert int(st['tree_size'][1:].sum()) + samples.shape[0] + 1 == int(samp[k + 'n_call'])


def test_nuts_replay_divergent(samp):
    """Huge step size: divergences (|dE| >= max_change) and immediate U-turns, nuts.py:119-132."""
    spec = rebuild_spec(samp, 'div5.')
    k = 'div5.nuts0.'
    chain = orc.Chain(samp['div5.x0'][0], step_size=40.)
    rng = orc.make_rng('replay', normals=samp[k + 'normals'], uniforms=samp[k + 'uniforms'])
    samples, st = orc.nuts_run(spec, chain, rng, 30, 10, max_change=50.)
    assert st['diverging'].sum() >= 1
    for f in ('tree_depth', 'tree_size', 'diverging'):
        assert np.array_equal(st[f], samp[k + f]), f
    np.testing.assert_allclose(samples, samp[k + 'samples'], rtol=1e-9, atol=1e-9)
    assert rng[0].i_uniform == samp[k + 'uniforms'].size


@pytest.mark.parametrize('name,c', [('plain16', 0), ('d64', 1), ('full5', 0)])
def test_hmc_replay(samp, name, c):
    spec = rebuild_spec(samp, name + '.')
    k = '%s.hmc%d.' % (name, c)
    chain = orc.Chain(samp[name + '.x0'][c])
    rng = orc.make_rng('replay', normals=samp[k + 'normals'], uniforms=samp[k + 'uniforms'])
    n_iter = samp[k + 'samples'].shape[0]
    samples, st = orc.hmc_run(spec, chain, rng, n_iter, int(samp[k + 'warmup'].sum()), n_int_step=8)
    for f in ('accepted', 'diverging', 'n_int_step'):
        assert np.array_equal(st[f], samp[k + f]), f
    assert rng[0].i_uniform == samp[k + 'uniforms'].size
    # Early warm-up runs 8-step trajectories at step sizes far beyond the leapfrog stability limit, so
    # summation-order differences (1e-14 at iteration 2) grow exponentially: tight on the head, loose overall.
    np.testing.assert_allclose(samples[:8], samp[k + 'samples'][:8], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(samples, samp[k + 'samples'], rtol=1e-4, atol=1e-4)
    for f in ('logp', 'energy', 'accept_stat', 'step_size', 'step_size_bar', 'energy_change'):
        np.testing.assert_allclose(st[f][:8], samp[k + f][:8], rtol=1e-8, atol=1e-8, err_msg=f)
        np.testing.assert_allclose(st[f], samp[k + f], rtol=1e-3, atol=1e-3, err_msg=f)


def test_xoshiro_reference_vectors():
    """xoshiro256++ known answers: state {1,2,3,4} (first outputs of the public reference implementation)."""
    L = orc.lib()
    s = (C.c_uint64 * 4)(1, 2, 3, 4)
    got = [L.bfo_xoshiro_next(s) for _ in range(3)]
    assert got == [41943041, 58720359, 3588806011781223]


# ---- full-rank metric (QuadMetricFull / QuadMetricFullAdapt): fixture sampler_fullmetric.npz ----------------

@pytest.fixture(scope='module')
def fullm():
    return np.load(os.path.join(G, 'sampler_fullmetric.npz'))


def test_full_metric_leapfrog(fullm):
    """CpuLeapfrogIntegrator with QuadMetricFull: velocity = cov p (metrics.py:113-115)."""
    spec = rebuild_spec(fullm, 'fm8.')
    cov = fullm['fm8.cov0']
    for a, b, eps in (('s0', 's1', fullm['fm8.lf.eps'][0]), ('s1', 's2', fullm['fm8.lf.eps'][1])):
        r = orc.leapfrog_full(spec, cov, eps, fullm['fm8.lf.%s.q' % a], fullm['fm8.lf.%s.p' % a], fullm['fm8.lf.%s.q_grad' % a])
        for f, g in (('q', 'q'), ('p', 'p'), ('v', 'velocity'), ('grad', 'q_grad'), ('energy', 'energy'), ('logp', 'logp')):
            np.testing.assert_allclose(r[f], fullm['fm8.lf.%s.%s' % (b, g)], rtol=1e-11, atol=1e-11, err_msg=f)


def _replay_full(fullm, key, chain_i, sampler, **chain_kw):
    spec = rebuild_spec(fullm, 'fm8.')
    chain = orc.Chain(fullm['fm8.x0'][chain_i], **chain_kw)
    rng = orc.make_rng('replay', normals=fullm[key + 'normals'], uniforms=fullm[key + 'uniforms'])
    n_iter, n_warmup = int(fullm['fm8.n_iter']), int(fullm['fm8.n_warmup'])
    if sampler == 'NUTS':
        out = orc.nuts_run(spec, chain, rng, n_iter, n_warmup)
    else:
        out = orc.hmc_run(spec, chain, rng, n_iter, n_warmup, n_int_step=8)
    return out, chain, rng


@pytest.mark.parametrize('c', [0, 1])
def test_full_metric_adaptive_nuts_replay(fullm, c):
    """NTrace(metric='full'): Welford covariance windows, per-iteration Cholesky, window switch and doubling
    (metrics.py:240-330,374-417) under the reference's own logged draws."""
    k = 'fm8.nuts_adapt%d.' % c
    (samples, st), chain, rng = _replay_full(fullm, k, c, 'NUTS', metric='full', adapt_window=8)
    for f in ('tree_depth', 'tree_size', 'diverging', 'warmup'):
        assert np.array_equal(st[f], fullm[k + f]), f
    assert rng[0].i_uniform == fullm[k + 'uniforms'].size and rng[0].i_normal == fullm[k + 'normals'].size
    # BLAS / LAPACK summation orders (dgemv, dtrtrs, dpotrf) differ from the plain loops here at the 1e-16 level and
    # the warm-up trajectories (up to 63 leapfrogs, covariance still rough) amplify that: tight on the head of the
    # run, loose overall, discrete fields exact over the whole horizon (above)
    np.testing.assert_allclose(samples[:7], fullm[k + 'samples'][:7], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(samples, fullm[k + 'samples'], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(chain.mat('cov'), fullm[k + 'final_cov'], rtol=1e-3, atol=1e-3)
    for f in ('energy', 'step_size', 'mean_tree_accept'):
        np.testing.assert_allclose(st[f][:7], fullm[k + f][:7], rtol=1e-8, atol=1e-8, err_msg=f)
        np.testing.assert_allclose(st[f], fullm[k + f], rtol=5e-3, atol=5e-3, err_msg=f)


def test_full_metric_fixed_nuts_and_hmc_replay(fullm):
    cov0 = fullm['fm8.cov0']
    k = 'fm8.nuts_fixed0.'
    (samples, st), chain, rng = _replay_full(fullm, k, 0, 'NUTS', metric=cov0, adapt_metric=False)
    for f in ('tree_depth', 'tree_size', 'diverging'):
        assert np.array_equal(st[f], fullm[k + f]), f
    np.testing.assert_allclose(samples, fullm[k + 'samples'], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(chain.mat('cov'), cov0, rtol=0, atol=0)
    # HMC at step sizes near the stability limit amplifies the rounding-level differences (cf. test_hmc_replay)
    k = 'fm8.hmc_fixed1.'
    (samples, st), chain, rng = _replay_full(fullm, k, 1, 'HMC', metric=cov0, adapt_metric=False)
    assert np.array_equal(st['accepted'], fullm[k + 'accepted'])
    np.testing.assert_allclose(samples[:10], fullm[k + 'samples'][:10], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(samples, fullm[k + 'samples'], rtol=1e-2, atol=1e-2)
    k = 'fm8.hmc_adapt0.'
    (samples, st), chain, rng = _replay_full(fullm, k, 0, 'HMC', metric=cov0, adapt_window=8)
    assert np.array_equal(st['accepted'], fullm[k + 'accepted'])
    np.testing.assert_allclose(samples[:10], fullm[k + 'samples'][:10], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(samples, fullm[k + 'samples'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(chain.mat('cov'), fullm[k + 'final_cov'], rtol=1e-4, atol=1e-4)


def test_full_metric_rejects_indefinite_covariance():
    with pytest.raises(ValueError):
        orc.Chain(np.zeros(3), metric=np.diag([1., -1., 1.]))


def test_tuned_baseline_evaluation_agrees_with_the_faithful_one():
    """bench.py's CPU baseline times bf_cpu_tuned.c (symmetrised dense matvec, AVX2); it must be the same density: equal
    to rounding inside the bound and outside (where it extrapolates by linearity instead of a second evaluation)."""
    from bayesfast_amd.workloads import correlated_gaussian_spec
    spec, cov = correlated_gaussian_spec(64)
    rng = np.random.default_rng(3)
    L = np.linalg.cholesky(cov)
    x = np.concatenate((rng.normal(size=(200, 64)) @ L.T, 6. * rng.normal(size=(20, 64)) @ L.T))
    f0, g0 = orc.logp_and_grad(spec, x)
    f1, g1 = orc.logp_and_grad(spec, x, tuned=True)
    np.testing.assert_allclose(f1, f0, rtol=1e-12, atol=1e-11)
    np.testing.assert_allclose(g1, g0, rtol=1e-11, atol=1e-11)
    f2, _ = orc.logp_and_grad(spec, x)  # and the registration does not outlive the call
    assert np.array_equal(f2, f0)


def _random_single_output_spec(d, rng, cubic, decay, transform, su, link):
    """A single-output surrogate density with the chosen features (test input for the tuned evaluation)."""
    cfgs = [dict(order='linear', input_mask=np.arange(d), output_mask=np.arange(1), coef=rng.normal(size=(1, d + 1))),
            dict(order='quadratic', input_mask=np.arange(d), output_mask=np.arange(1), coef=np.triu(rng.normal(size=(d, d)))[None] * -0.3)]
    if cubic:
        m2, m3 = np.sort(rng.choice(d, 5, replace=False)), np.sort(rng.choice(d, 6, replace=False))
        a3 = np.zeros((1, 6, 6, 6))
        for j in range(6):
            for k in range(j + 1, 6):
                for l in range(k + 1, 6):
                    a3[0, j, k, l] = 0.05 * rng.normal()
        cfgs += [dict(order='cubic-2', input_mask=m2, output_mask=np.arange(1), coef=0.05 * rng.normal(size=(1, 5, 5))),
                 dict(order='cubic-3', input_mask=m3, output_mask=np.arange(1), coef=a3)]
    xs = rng.normal(size=(40 * d, d))
    poly = dict(input_size=d, output_size=1, configs=cfgs, use_bound=False)
    poly.update(orc.set_bound(poly, xs, rng.normal(size=xs.shape[0]), dict(alpha_p=80.)))
    spec = dict(d=d, poly=poly)
    if decay == 2:   # the decay term on the bound's own points: its centre and Hessian ARE the bound's arrays (SurrogateDensity.fit's case)
        spec.update(orc.set_decay(xs, alpha_p=90.))
        assert np.array_equal(spec['decay_hess'], poly['hess']) and np.array_equal(spec['decay_mu'], poly['mu'])
    elif decay:
        spec.update(orc.set_decay(xs * 0.7, alpha_p=90.))
    if transform:
        spec['ranges'] = np.stack((-4. - rng.uniform(size=d), 4. + rng.uniform(size=d)), axis=1)
        hb = np.zeros((d, 2), np.uint8)
        hb[: d // 3] = 1
        hb[d // 3: d // 2, 0] = 1
        spec['hard_bounds'] = hb
    if su:
        spec['su_lo'], spec['su_diff'] = rng.normal(size=d) * 0.1, rng.uniform(0.7, 1.5, size=d)
    if link:
        spec['link'] = dict(kind='gaussian', y=0.3, prec=0.7, logp0=-1.2)
    return spec


@pytest.mark.parametrize('cubic,decay,transform,su,link', [(0, 1, 0, 0, 0), (1, 0, 0, 0, 0), (1, 1, 1, 1, 0), (0, 0, 1, 0, 1), (0, 1, 0, 1, 0), (1, 0, 0, 0, 1),
                                                          (0, 2, 0, 0, 0), (1, 2, 1, 0, 0), (0, 2, 0, 1, 1)])
def test_tuned_evaluation_of_every_single_output_feature_set(cubic, decay, transform, su, link):
    """Round 6: the tuned evaluation covers decay, points outside the bound (by linearity, or a second evaluation with cubic
    configs), cubic configs, transforms, input scaling and the Gaussian link -- the config blocks' CPU baselines all run it."""
    d = 12
    rng = np.random.default_rng(100 + 16 * cubic + 8 * decay + 4 * transform + 2 * su + link)
    spec = _random_single_output_spec(d, rng, cubic, decay, transform, su, link)
    for original_space in (False, True):
        x = rng.normal(size=(300, d)) * np.repeat([0.6, 2.5], 150)[:, None]
        if transform and original_space:
            x = np.clip(x, -3.9, 3.9)
        f0, g0 = orc.logp_and_grad(spec, x, original_space=original_space)
        f1, g1 = orc.logp_and_grad(spec, x, original_space=original_space, tuned=True)
        assert np.all(np.isfinite(f0))
        np.testing.assert_allclose(f1, f0, rtol=1e-11, atol=1e-10)
        np.testing.assert_allclose(g1, g0, rtol=1e-10, atol=1e-10)
    # points on both sides of the bound were there
    xm = (x - spec.get('su_lo', 0.)) / spec.get('su_diff', 1.) - spec['poly']['mu'] if not transform else None
    if xm is not None:
        b = np.sqrt(np.einsum('ij,jk,ik->i', xm, spec['poly']['hess'], xm))
        assert 20 < np.sum(b > spec['poly']['alpha']) < 280


def test_tuned_evaluation_takes_hessians_that_are_symmetric_only_to_rounding():
    """np.linalg.inv of an ill-conditioned covariance is symmetric to ~1e-16 x cond, not bit for bit (the refitted banana of
    bench.py's headline, config 5's cond-1e4 target): the tuned evaluation symmetrises and must not decline such a density."""
    d = 16
    rng = np.random.default_rng(77)
    spec = _random_single_output_spec(d, rng, 0, 1, 0, 0, 0)
    sc = np.logspace(-3, 2, d)
    xs = rng.normal(size=(40 * d, d)) * sc
    spec['poly'].update(orc.set_bound(dict(spec['poly'], use_bound=False), xs, rng.normal(size=xs.shape[0]), dict(alpha_p=80.)))
    spec.update(orc.set_decay(xs, alpha_p=90.))
    h = spec['poly']['hess']
    assert np.max(np.abs(h - h.T)) > 0.   # (not exactly symmetric, or the test tests nothing)
    cs = orc.ChainSet(spec, xs[:2], 1, tuned=True)
    assert cs.tuned
    cs.close()
    x = rng.normal(size=(100, d)) * sc * np.repeat([0.5, 3.], 50)[:, None]
    f0, g0 = orc.logp_and_grad(spec, x)
    f1, g1 = orc.logp_and_grad(spec, x, tuned=True)
    np.testing.assert_allclose(f1, f0, rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(g1, g0, rtol=1e-8, atol=1e-8 * np.max(np.abs(g0)))


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_tuned_evaluation_of_the_des_shaped_pipeline_matches_the_reference(tag):
    """The tuned pipeline evaluation (monomial vector, two dense products) against the REFERENCE's values (pipeline_des.npz)."""
    from specio import rebuild_pipeline_des
    z = np.load(os.path.join(G, 'pipeline_des.npz'))
    spec = rebuild_pipeline_des(z, tag)
    for sp, key, pts in ((True, 'orig', z[tag + '.xo']), (False, 'trans', z[tag + '.xt'])):
        lp, g = orc.logp_and_grad(spec, pts, original_space=sp, tuned=True)
        np.testing.assert_allclose(lp, z['%s.logp_%s' % (tag, key)], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(g, z['%s.grad_%s' % (tag, key)], rtol=1e-10, atol=1e-10)


def test_tuned_evaluation_of_a_pipeline_with_full_precision_and_cubic_configs():
    from specio import rebuild_pipeline
    z = np.load(os.path.join(G, 'pipeline.npz'))
    spec = rebuild_pipeline(z)
    lp, g = orc.logp_and_grad(spec, z['xt'], original_space=True, tuned=True)
    np.testing.assert_allclose(lp, z['logp'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(g, z['grad'], rtol=1e-10, atol=1e-10)


# ---- multi-output surrogate + chi-square (+ prior) pipelines, SURVEY 8f-1: fixtures pipeline.npz, pipeline_des.npz ----

def test_pipeline_density_with_full_precision_and_cubic_configs():
    """[surrogate with linear, quadratic, cubic-2, cubic-3 configs and output masks; chi-square with a full precision
    matrix]: the oracle's chi2 stage equals the reference's Density.logp_and_grad(use_surrogate=True), inside and outside
    the bound (core/density.py:552-560, modules/poly.py:480-503)."""
    from specio import rebuild_pipeline
    z = np.load(os.path.join(G, 'pipeline.npz'))
    spec = rebuild_pipeline(z)
    lp, g = orc.logp_and_grad(spec, z['xt'], original_space=True)
    np.testing.assert_allclose(lp, z['logp'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g, z['grad'], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_des_shaped_pipeline_density(tag):
    """The three-module pipeline of examples/des-y1-w-cosmosis.ipynb in small: surrogate (linear + masked quadratic, with
    surrogate input scales) -> whitened chi-square -> like + Gaussian prior of some inputs, behind the Density's input
    scales and hard bounds, without ('a') and with ('b') the decay term; both spaces, points inside and outside the bound."""
    from specio import rebuild_pipeline_des
    z = np.load(os.path.join(G, 'pipeline_des.npz'))
    spec = rebuild_pipeline_des(z, tag)
    assert 0 < int(z[tag + '.n_outside_bound']) < z[tag + '.xo'].shape[0]
    for sp, key, pts in ((True, 'orig', z[tag + '.xo']), (False, 'trans', z[tag + '.xt'])):
        lp, g = orc.logp_and_grad(spec, pts, original_space=sp)
        np.testing.assert_allclose(lp, z['%s.logp_%s' % (tag, key)], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(g, z['%s.grad_%s' % (tag, key)], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize('tag,c', [('a', 0), ('a', 1), ('b', 0), ('b', 1)])
def test_des_shaped_pipeline_nuts_replay(tag, c):
    """The reference's NUTS on that density, replayed with its logged draws: exact tree statistics and draw counts."""
    from specio import rebuild_pipeline_des
    z = np.load(os.path.join(G, 'pipeline_des.npz'))
    spec = rebuild_pipeline_des(z, tag)
    k = '%s.nuts%d.' % (tag, c)
    chain = orc.Chain(z[tag + '.x0'][c])
    rng = orc.make_rng('replay', normals=z[k + 'normals'], uniforms=z[k + 'uniforms'])
    samples, st = orc.nuts_run(spec, chain, rng, int(z[tag + '.n_iter']), int(z[tag + '.n_warmup']))
    for f in ('tree_depth', 'tree_size', 'diverging', 'warmup'):
        assert np.array_equal(st[f], z[k + f]), f
    assert rng[0].i_uniform == z[k + 'uniforms'].size and rng[0].i_normal == z[k + 'normals'].size
    np.testing.assert_allclose(samples, z[k + 'samples'], rtol=1e-8, atol=1e-8)
    for f in ('logp', 'energy', 'step_size', 'step_size_bar'):
        np.testing.assert_allclose(st[f], z[k + f], rtol=1e-7, atol=1e-7, err_msg=f)
