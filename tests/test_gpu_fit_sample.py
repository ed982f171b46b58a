"""GPU parity of the surrogate fit (design blocks, MFMA Gram, Cholesky solve) and of the public
``PolyModel`` / ``SurrogateDensity`` / ``sample`` interfaces against the CPU oracle.

The fit solves the normal equations instead of LAPACK gelsd (modules/poly.py:570), so coefficients are
compared at a condition-number-dependent tolerance (1e-8 relative to the largest coefficient of a block) and
predictions at 1e-9 of the data scale."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _indep(order, coef):
    n = coef.shape[-1]
    if order == 'quadratic':
        return coef[..., np.triu(np.ones((n, n), bool))]
    if order == 'cubic-3':
        i, j, k = np.meshgrid(*[np.arange(n)] * 3, indexing='ij')
        return coef[..., (i < j) & (j < k)]
    return coef


def test_design_blocks_match_reference_packing():
    """bfhip_design_block vs the golden _lsq_* blocks of the reference (modules/_poly.pyx:143-177)."""
    import ctypes as C
    import torch
    from bayesfast_amd.device import get_context, _ptr
    from bayesfast_amd import _lib
    ctx = get_context(0)
    kern = np.load(os.path.join(G, 'poly_kernels.npz'))
    for n in (2, 3, 5, 8):
        xs = kern['n%d.xs' % n]
        xt = ctx.tensor(xs, torch.float64)
        for oi, order in ((1, 'quadratic'), (2, 'cubic_2'), (3, 'cubic_3')):
            want = kern['n%d.%s.lsq' % (n, order)]
            if want.shape[1] == 0:
                continue
            A = ctx.zeros((xs.shape[0], want.shape[1] + 3))
            _lib.check(ctx._lib.bfhip_design_block(ctx.handle, oi, xs.shape[0], n, _ptr(xt), None, _ptr(A), A.shape[1], 2))
            got = A.cpu().numpy()
            assert np.array_equal(got[:, 2:2 + want.shape[1]], want)
            assert not got[:, :2].any() and not got[:, -1].any()
        A = ctx.zeros((xs.shape[0], n + 1))
        w = ctx.tensor(np.arange(1., xs.shape[0] + 1.), torch.float64)
        _lib.check(ctx._lib.bfhip_design_block(ctx.handle, 0, xs.shape[0], n, _ptr(xt), _ptr(w), _ptr(A), n + 1, 0))
        assert np.array_equal(A.cpu().numpy(), np.concatenate([np.ones((xs.shape[0], 1)), xs], 1) * np.arange(1., xs.shape[0] + 1.)[:, None])


def test_gram_and_solve_vs_numpy():
    import torch
    from bayesfast_amd.device import get_context, _ptr
    from bayesfast_amd import _lib
    ctx = get_context(0)
    rng = np.random.default_rng(0)
    for n, P, m in ((300, 70, 2), (1000, 333, 1), (130, 129, 3)):
        A = rng.normal(size=(n, P))
        B = rng.normal(size=(n, m))
        At, Bt = ctx.tensor(A, torch.float64), ctx.tensor(B, torch.float64)
        Gt, rt = ctx.empty((P, P)), ctx.empty((P, m))
        info = torch.ones((1,), dtype=torch.int32, device=ctx.device)
        _lib.check(ctx._lib.bfhip_gram(ctx.handle, n, P, m, _ptr(At), P, _ptr(Bt), _ptr(Gt), _ptr(rt)))
        np.testing.assert_allclose(Gt.cpu().numpy(), A.T @ A, rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(rt.cpu().numpy(), A.T @ B, rtol=1e-12, atol=1e-10)
        _lib.check(ctx._lib.bfhip_solve_spd(ctx.handle, P, m, _ptr(Gt), _ptr(rt), _ptr(info)))
        assert int(info.item()) == 0
        np.testing.assert_allclose(rt.cpu().numpy(), np.linalg.lstsq(A, B, rcond=None)[0], rtol=1e-8, atol=1e-9)
    # a singular system is reported, not silently solved
    A = rng.normal(size=(50, 10))
    A[:, 3] = A[:, 2]
    At = ctx.tensor(A, torch.float64)
    Gt, rt = ctx.empty((10, 10)), ctx.empty((10, 1))
    info = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
    _lib.check(ctx._lib.bfhip_gram(ctx.handle, 50, 10, 1, _ptr(At), 10, _ptr(ctx.tensor(A[:, :1].copy(), torch.float64)), _ptr(Gt), _ptr(rt)))
    _lib.check(ctx._lib.bfhip_solve_spd(ctx.handle, 10, 1, _ptr(Gt), _ptr(rt), _ptr(info)))
    assert int(info.item()) > 0


def test_polymodel_fit_matches_reference_fixture():
    """Masked multi-output model of the reference fixture: device fit vs the reference's own coefficients."""
    from bayesfast_amd import PolyModel, PolyConfig
    from specio import rebuild_poly
    z = np.load(os.path.join(G, 'polymodel.npz'))
    ref = rebuild_poly(z)
    configs = [PolyConfig(c['order'], c['input_mask'], c['output_mask']) for c in ref['configs']]
    pm = PolyModel(configs, input_size=6, output_size=3, bound_options=dict(alpha_p=float(z['alpha_p'])))
    assert pm.n_param == int(z['n_param'])
    pm.fit(z['x_fit'], z['y_fit'], z['logp_fit'])
    for c, r in zip(pm.configs, ref['configs']):
        a, b = _indep(c.order, c._coef), _indep(r['order'], r['coef'])
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-8 * np.abs(b).max())
    np.testing.assert_allclose(pm._alpha, ref['alpha'], rtol=1e-12)
    np.testing.assert_allclose(pm._f_mu, ref['f_mu'], rtol=1e-8)
    # evaluation inside and outside the bound, through the reference-style wrappers
    for x, f, j in zip(z['x_eval'][::3], z['f'][::3], z['j'][::3]):
        ff, jj = pm.fun_and_jac(x)
        np.testing.assert_allclose(ff[0], f, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(jj[0], j, rtol=1e-6, atol=1e-7)
    # weighted fit
    pw = rebuild_poly(z, 'w.poly.')
    p2 = PolyModel('quadratic', input_size=4, output_size=1)
    p2.fit(z['w.x_fit'], z['w.y_fit'], logp=z['w.y_fit'][:, 0], w=z['w.w'])
    for c, r in zip(p2.configs, pw['configs']):
        np.testing.assert_allclose(_indep(c.order, c._coef), _indep(r['order'], r['coef']), rtol=0, atol=1e-9 * np.abs(r['coef']).max() + 1e-12)


def test_fit_headline_size_recovers_exact_quadratic():
    """d = 64, n = 2 P = 4290 points, P = 2145 parameters: exact quadratic target."""
    from bayesfast_amd import PolyModel
    from bayesfast_amd.workloads import correlated_gaussian_spec
    spec, cov = correlated_gaussian_spec(64)
    P = np.linalg.inv(cov)
    rng = np.random.default_rng(7)
    x = rng.normal(size=(4290, 64))
    y = -0.5 * np.einsum('ij,jk,ik->i', x, P, x)
    pm = PolyModel('quadratic', input_size=64, output_size=1)
    pm.fit(x, y[:, None], y)
    want = spec['poly']['configs'][1]['coef'][0]
    got = pm.configs[1]._coef[0]
    np.testing.assert_allclose(got[np.triu_indices(64)], want[np.triu_indices(64)], rtol=0, atol=1e-10)
    np.testing.assert_allclose(pm.configs[0]._coef[0], 0., atol=1e-9)
    np.testing.assert_allclose(pm._alpha, spec['poly']['alpha'], rtol=1e-12)
    xt = rng.normal(size=64)
    ff, jj = pm.fun_and_jac(xt)
    np.testing.assert_allclose(ff[0][0], -0.5 * xt @ P @ xt, rtol=1e-10)
    np.testing.assert_allclose(jj[0][0], -P @ xt, rtol=1e-9, atol=1e-10)


def test_sample_end_to_end_matches_oracle_and_resumes():
    """SurrogateDensity.fit + sample(): scales, hard bounds, decay and surrogate scales; trajectories of the
    public entry point equal the oracle's for the same streams; a second sample() call continues the chains."""
    from bayesfast_amd import PolyModel, SurrogateDensity, sample, NTrace
    from bayesfast_amd.samplers import _get_step_size, _get_metric
    from oracle import oracle as orc
    d = 5
    rng = np.random.default_rng(21)
    Pm = np.eye(d) + 0.3 * rng.normal(size=(d, d)) / np.sqrt(d)
    Pm = Pm @ Pm.T
    scales = np.stack([-6. - rng.uniform(size=d), 7. + rng.uniform(size=d)], 1)
    hb = np.zeros((d, 2), int)
    hb[0], hb[1], hb[2] = (1, 1), (1, 0), (0, 1)
    su = PolyModel('quadratic', input_size=d, output_size=1,
                   input_scales=np.stack([-2. - rng.uniform(size=d), 3. + rng.uniform(size=d)], 1))
    den = SurrogateDensity(su, input_scales=scales, hard_bounds=hb, decay_options=dict(use_decay=True))
    xf = rng.normal(size=(80, d))
    yf = -0.5 * np.einsum('ij,jk,ik->i', xf, Pm, xf) - 0.02 * np.sum(xf**3, 1)
    den.fit(xf, yf)
    spec = den.spec()
    x0_orig = rng.normal(size=(6, d)) * 0.5
    tr = NTrace(n_chain=6, n_iter=30, n_warmup=12, x_0=x0_orig, random_generator=5)
    tt = sample(den, tr, n_run=20, verbose=False)
    assert tt.i_iter == 20 and not tt.finished
    tt = sample(den, tt, verbose=False)
    assert tt.finished and tt.samples.shape == (6, 30, d)
    x0_t = den.from_original(x0_orig)
    for i in (0, 3, 5):
        ch = orc.Chain(x0_t[i])
        so, sto = orc.nuts_run(spec, ch, orc.make_rng('xoshiro', seed=5, stream=i), 30, 12)
        assert np.array_equal(tt[i].stats._tree_size, sto['tree_size'].astype(int))
        np.testing.assert_allclose(tt[i].samples[:8], so[:8], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(tt[i].samples, so, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tt.samples_original, den.to_original(tt.samples))
    lp, _ = orc.logp_and_grad(spec, tt.samples_original.reshape(-1, d), original_space=True)
    np.testing.assert_allclose(tt.get(include_warmup=True, return_type='logp'), lp, rtol=1e-8, atol=1e-8)
    assert tt.get().shape == (6 * 18, d)
    assert _get_step_size(tt) > 0 and _get_metric(tt, 'diag').shape == (d,)
    with pytest.raises(ValueError):  # non-finite logp at x_0 (base_hmc.py:42-46)
        bad = den.to_original(np.full((6, d), 40.))
        sample(den, NTrace(n_chain=6, n_iter=5, n_warmup=2, x_0=np.full((6, d), np.nan)), verbose=False)
