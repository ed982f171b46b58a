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
    for n, P, m in ((300, 70, 2), (1000, 333, 1), (130, 129, 3), (500, 200, 6), (60, 50, 1)):   # (m = 6: two passes of the sweeps)
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


@pytest.mark.parametrize('n,P', [(700, 520), (1500, 1024), (333, 777), (2600, 1300)])
def test_gram_in_128_column_blocks_equals_the_one_wave_kernel_bit_for_bit(n, P):
    """bf_gram128_kernel (four waves on a 128 x 128 block, the panels of a 16-row stage through LDS once) against bf_gram_kernel (one
    wave per 64 x 64 block): the same split-K partials in the same order, so the same Gram matrix bit for bit -- an odd and an even
    number of 64-column blocks, a ragged last block, fewer rows than columns -- and A^T A to rounding."""
    import torch
    from bayesfast_amd.device import get_context, _ptr
    from bayesfast_amd import _lib
    ctx = get_context(0)
    A = np.random.default_rng(n + P).normal(size=(n, P))
    B = A[:, :1].copy()
    At, Bt = ctx.tensor(A, torch.float64), ctx.tensor(B, torch.float64)
    out = {}
    try:
        for one_wave in (1, 0):
            _lib.debug_set('gram_one_wave', one_wave)
            Gt, rt = ctx.empty((P, P)), ctx.empty((P, 1))
            Gt.fill_(float('nan'))
            _lib.check(ctx._lib.bfhip_gram(ctx.handle, n, P, 1, _ptr(At), P, _ptr(Bt), _ptr(Gt), _ptr(rt)))
            out[one_wave] = Gt.cpu().numpy()
    finally:
        _lib.debug_set('gram_one_wave', 0)
    assert np.array_equal(out[0], out[1])
    np.testing.assert_allclose(out[0], A.T @ A, rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize('n,P,m', [(1500, 700, 2), (2400, 1100, 1), (1300, 1283, 3)])
def test_cholesky_with_two_panels_per_pass_solves_like_the_one_panel_form(n, P, m):
    """bfhip_solve_spd with two panels per pass over the trailing matrix (delayed rank-128 update; the second panel's block column
    brought up to date by a column launch first) against one panel per pass and against NumPy: an odd and an even number of panels,
    a ragged last block, a nearly square system; the rank-deficiency report still names the first bad pivot."""
    import torch
    from bayesfast_amd.device import get_context, _ptr
    from bayesfast_amd import _lib
    ctx = get_context(0)
    rng = np.random.default_rng(n + P)
    A = rng.normal(size=(n, P)) * np.exp(rng.normal(size=P))[None]   # (columns on different scales: the equilibration matters)
    B = rng.normal(size=(n, m))
    At, Bt = ctx.tensor(A, torch.float64), ctx.tensor(B, torch.float64)
    out = {}
    try:
        for one in (1, 0):
            _lib.debug_set('chol_one_panel', one)
            Gt, rt = ctx.empty((P, P)), ctx.empty((P, m))
            info = torch.ones((1,), dtype=torch.int32, device=ctx.device)
            _lib.check(ctx._lib.bfhip_gram(ctx.handle, n, P, m, _ptr(At), P, _ptr(Bt), _ptr(Gt), _ptr(rt)))
            _lib.check(ctx._lib.bfhip_solve_spd(ctx.handle, P, m, _ptr(Gt), _ptr(rt), _ptr(info)))
            assert int(info.item()) == 0
            out[one] = rt.cpu().numpy()
        ref = np.linalg.lstsq(A, B, rcond=None)[0]
        np.testing.assert_allclose(out[0], out[1], rtol=1e-9, atol=1e-11 * np.abs(ref).max())
        np.testing.assert_allclose(out[0], ref, rtol=1e-6, atol=1e-8 * np.abs(ref).max())
        # a duplicated column in the second panel of a pair: reported with its index, as by the one-panel form
        A2 = A.copy()
        A2[:, 100] = A2[:, 99]
        At2 = ctx.tensor(A2, torch.float64)
        bad = {}
        for one in (1, 0):
            _lib.debug_set('chol_one_panel', one)
            Gt, rt = ctx.empty((P, P)), ctx.empty((P, m))
            info = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
            _lib.check(ctx._lib.bfhip_gram(ctx.handle, n, P, m, _ptr(At2), P, _ptr(Bt), _ptr(Gt), _ptr(rt)))
            _lib.check(ctx._lib.bfhip_solve_spd(ctx.handle, P, m, _ptr(Gt), _ptr(rt), _ptr(info)))
            bad[one] = int(info.item())
        assert bad[0] == bad[1] == 101
    finally:
        _lib.debug_set('chol_one_panel', 0)


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


def _compare_chains_with_oracle(tt, den, seed, chains, n_iter, n_warmup, **orc_kw):
    from oracle import oracle as orc
    spec = den.spec()
    x0_t = den.from_original(tt._trace.x_0) if not tt._trace.x_0_transformed else tt._trace.x_0
    for i in chains:
        ch = orc.Chain(x0_t[i], **{k: v for k, v in orc_kw.items() if k in ('target_accept', 'step_size')})
        so, sto = orc.nuts_run(spec, ch, orc.make_rng('xoshiro', seed=seed, stream=i), n_iter, n_warmup)
        assert np.array_equal(tt[i].stats._tree_size, sto['tree_size'].astype(int)), i
        assert np.array_equal(tt[i].stats._tree_depth, sto['tree_depth'].astype(int)), i
        assert np.array_equal(np.array(tt[i].stats._diverging, dtype=float), sto['diverging']), i
        np.testing.assert_allclose(tt[i].samples[:6], so[:6], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(tt[i].samples, so, rtol=1e-3, atol=1e-3)


def test_config3_banana_quadratic_surrogate_with_refit_cycle():
    """BASELINE config 3: 64-d rotated banana (examples/banana-gbs.ipynb cell 3 at D = 64, Q = 0.01), quadratic
    PolyModel fitted on 2 P points, NUTS, then ONE refit cycle on points drawn from the first round.  The
    first-round quadratic surrogate is indefinite, so chains live on the extrapolation bound with deep trees:
    a stress case for the fit, the out-of-bound branch and the tree logic (checked against the oracle)."""
    from scipy.stats import special_ortho_group
    from bayesfast_amd import PolyModel, SurrogateDensity, sample, NTrace
    D, Q = 64, 0.01
    np.random.seed(0)
    A = special_ortho_group.rvs(D)

    def logp(x):
        x = x @ A.T
        return -np.sum((x[..., ::2]**2 - x[..., 1::2])**2 / Q + (x[..., ::2] - 1)**2, axis=-1)

    su = PolyModel('quadratic', input_size=D, output_size=1)
    den = SurrogateDensity(su)
    P = su.n_param
    assert P == 2145
    rng = np.random.default_rng(1)
    x_fit = rng.normal(size=(2 * P, D))
    den.fit(x_fit, logp(x_fit))
    # the device fit solves the same least-squares problem as the reference's lstsq
    resid = su.fun(x_fit[0])[0][0] - logp(x_fit[0])
    from oracle import oracle as orc
    ref = orc.poly_fit(su.poly_spec(use_bound=False), x_fit, logp(x_fit)[:, None], logp(x_fit))
    np.testing.assert_allclose(su.configs[1]._coef[0][np.triu_indices(D)], ref['configs'][1]['coef'][0][np.triu_indices(D)],
                               rtol=0, atol=1e-7 * np.abs(ref['configs'][1]['coef']).max())
    tr = NTrace(n_chain=256, n_iter=24, n_warmup=12, x_0=x_fit[:256], random_generator=3)
    tt = sample(den, tr, verbose=False)
    assert tt.stat('tree_depth').max() >= 6
    _compare_chains_with_oracle(tt, den, 3, (0, 100, 255), 24, 12)
    # refit cycle: 2 P points spread evenly over the first-round samples, true logp, refit, sample again
    pool = np.unique(tt.get(include_warmup=True).reshape(-1, D), axis=0)  # rejected iterations repeat a sample
    pool = pool[np.all(np.abs(pool) < 50., axis=1)]
    n_new = min(P, pool.shape[0])
    x_new = np.concatenate([pool[np.linspace(0, pool.shape[0] - 1, n_new).astype(int)], x_fit[:2 * P - n_new]])
    den.fit(x_new, logp(x_new))
    tr2 = NTrace(n_chain=64, n_iter=16, n_warmup=8, x_0=x_new[:64], random_generator=4)
    tt2 = sample(den, tr2, verbose=False)
    _compare_chains_with_oracle(tt2, den, 4, (0, 63), 16, 8)
    assert abs(resid) < 1e6


def test_config4_funnel_target_accept_095():
    """BASELINE config 4 (one shard): 64-d funnel (examples/funnel-gbs.ipynb cell 3: a = 1, b = 0.5), quadratic
    surrogate, target_accept = 0.95; chain results depend on the GLOBAL chain index only (first_stream)."""
    from bayesfast_amd import PolyModel, SurrogateDensity, sample, NTrace
    D, a, b = 64, 1., 0.5

    def logp(x):
        return (-x[..., 0]**2 / (2 * a**2) - np.sum(x[..., 1:]**2, axis=-1) / (2 * np.exp(2 * b * x[..., 0])) -
                (D - 1) * b * x[..., 0])

    su = PolyModel('quadratic', input_size=D, output_size=1)
    den = SurrogateDensity(su, decay_options=dict(use_decay=True))
    rng = np.random.default_rng(2)
    x_fit = rng.normal(size=(2 * su.n_param, D))
    den.fit(x_fit, logp(x_fit))
    tr = NTrace(n_chain=96, n_iter=30, n_warmup=20, x_0=x_fit[:96] * 0.5, random_generator=9, target_accept=0.95)
    tt = sample(den, tr, verbose=False)
    _compare_chains_with_oracle(tt, den, 9, (0, 50, 95), 30, 20, target_accept=0.95)
    acc = tt.stat('mean_tree_accept')[:, 20:].mean()
    assert acc > 0.85, acc


@pytest.mark.parametrize('kernel', ['latency', 'pipelined'])
def test_decay_term_on_the_bounds_matrix_against_the_oracle(kernel):
    """The regime of config 3's second round at an oracle-affordable size: the banana's quadratic surrogate with the decay term (its
    centre and Hessian the bound's), chains started outside the bound so that their leaves are extrapolated (modules/poly.py:480-503) and
    the decay term is on -- the two-matrix forms of the latency kernel and of the pipelined kernel against the oracle's chains (which
    evaluate the decay term with its own matrix, as the reference does): trees, divergences, positions."""
    from bayesfast_amd import PolyModel, SurrogateDensity, sample, NTrace, _lib
    from bayesfast_amd._lib import debug_set
    from bayesfast_amd.workloads import banana_logp
    D = 64
    logp = banana_logp(D)
    su = PolyModel('quadratic', input_size=D, output_size=1)
    den = SurrogateDensity(su, decay_options=dict(use_decay=True))
    rng = np.random.default_rng(8)
    x_fit = rng.normal(size=(2 * su.n_param, D))
    den.fit(x_fit, logp(x_fit))
    assert np.array_equal(den._hess, su._hess)
    x0 = x_fit[:64] * 1.8                                       # beyond the alpha-ellipsoid of the fit points
    xm = x0 - su._mu
    assert (np.sqrt(np.einsum('ij,jk,ik->i', xm, su._hess, xm)) > su._alpha).mean() > 0.9
    try:
        debug_set('no_group', 1)
        if kernel == 'pipelined':
            debug_set('lone', 0)
            debug_set('wave_cpg', 16)
        tt = sample(den, NTrace(n_chain=64, n_iter=14, n_warmup=8, x_0=x0, random_generator=21), verbose=False, layout='wave')
        name = _lib.last_kernel()
    finally:
        for k, v in (('lone', 1), ('wave_cpg', 0), ('no_group', 0)):
            debug_set(k, v)
    assert name.startswith('bf_lone_kernel<4, false, 2,' if kernel == 'latency' else 'bf_nuts_pipe_kernel<4, false, 2,'), name
    _compare_chains_with_oracle(tt, den, 21, (0, 31, 63), 14, 8)
    assert tt.stat('tree_size').max() >= 15


@pytest.mark.parametrize('n_chain', [70, 600])
def test_decay_term_on_the_bounds_matrix(n_chain):
    """SurrogateDensity.fit takes the decay term's centre and Hessian from the same points, by the same statements, as the bound's
    (core/density.py:796-811, modules/poly.py:262-276): the upload sees identical arrays, and the wave-layout kernels then run TWO
    matrices (H_d (x - mu_d) is the bound's product, the decay radius the bound's; the plain surrogate's K-split).  The pipelined
    kernel and the latency kernel (a launch's tail, small launches) give the same numbers bit for bit in that form; the
    three-matrix form (debug switch) sums the bound's product in another order: the same chains to rounding while their trees
    agree."""
    from bayesfast_amd import PolyModel, SurrogateDensity, _lib
    from bayesfast_amd._lib import debug_set
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import banana_logp
    from bayesfast_amd.device import get_context
    ctx = get_context(0)
    D = 64
    logp = banana_logp(D)
    su = PolyModel('quadratic', input_size=D, output_size=1)
    den = SurrogateDensity(su, decay_options=dict(use_decay=True))
    rng = np.random.default_rng(3)
    x_fit = rng.normal(size=(2 * su.n_param, D))
    den.fit(x_fit, logp(x_fit))
    assert np.array_equal(den._hess, su._hess) and np.array_equal(den._mu, su._mu)   # (what the upload compares)
    x0 = rng.normal(size=(n_chain, D)) * 0.7
    out = {}
    try:
        debug_set('no_group', 1)
        for key, lone, off in (('pipe', 0, 0), ('lone', 2, 0), ('three', 0, 1)):
            if key == 'lone' and n_chain > 256:
                continue
            debug_set('no_decay_shared', off)
            debug_set('lone', lone)
            debug_set('wave_cpg', 16 if lone == 0 else 0)
            dc = DeviceChains(den.device(ctx), x0, seed=5)
            s_a, st_a = dc.run(40, 'NUTS', n_warmup=30, layout='wave')
            s_b, st_b = dc.run(25, 'NUTS', n_warmup=30, layout='wave')
            out[key] = [t.cpu().numpy() for t in (s_a, st_a, s_b, st_b)] + [dc.total_leapfrog, _lib.last_kernel()]
    finally:
        for k, v in (('no_decay_shared', 0), ('lone', 1), ('wave_cpg', 0), ('no_group', 0)):
            debug_set(k, v)
    assert out['pipe'][5] == 'bf_nuts_pipe_kernel<4, false, 2, 0>' and out['three'][5] == 'bf_nuts_pipe_kernel<4, false, 1, 0>'
    if 'lone' in out:
        assert out['lone'][5].startswith('bf_lone_kernel<4, false, 2,')
        for a, b in zip(out['pipe'][:4], out['lone'][:4]):
            np.testing.assert_array_equal(a, b)
        assert out['pipe'][4] == out['lone'][4]
    ts = _lib.NSTATS.index('tree_size')
    assert (out['pipe'][1][:, :, ts] > 1).any() and out['pipe'][4] > 65 * n_chain
    # two against three matrices: the first iterations agree to rounding (later ones as long as the trees do: chaotic dynamics)
    np.testing.assert_array_equal(out['pipe'][1][:, :3, ts], out['three'][1][:, :3, ts])
    np.testing.assert_allclose(out['pipe'][0][:, :3], out['three'][0][:, :3], rtol=1e-9, atol=1e-10)
    same = (out['pipe'][1][:, :, ts] == out['three'][1][:, :, ts]).mean()
    assert same > 0.9, same


@pytest.mark.parametrize('ipl,pipeline,n_rank', [(0, False, 2), (7, False, 2), (0, True, 2), (7, False, 8)])
def test_sample_two_ranks_equals_one_rank(tmp_path, ipl, pipeline, n_rank):
    """(ipl = 7: nine launches per round under layout 'auto' -- the layout of a launch is a pure function of the launches
    before it, decided from the trees of ALL ranks' chains, so it is the same for one rank and for two.  n_rank = 8: the node's
    launch rehearsed on one GPU -- ragged shards 3, 3, 3, 3, 3, 3, 2, 2 of the 22 chains, a layout vote per launch.)
    sample() under torch.distributed (two ranks sharing the box's GPU over gloo, ragged shards 11 + 11 of 22 chains) returns
    exactly what one rank returns: x_0 and the xoshiro streams follow the global chain index, the trace's seed is resolved
    once and broadcast, and the warm start of the next round (_get_step_size, _get_metric) reduces over all ranks."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    helper = os.path.join(root, 'tests', 'helpers', 'sample_ranks.py')
    env = dict(os.environ, BF_TEST_SEED='17', BFHIP_SHARE_DEVICE='1', BF_TEST_IPL=str(ipl))
    if pipeline:   # (the same on the pipeline density: Chi2PipelineDensity shards like any SurrogateDensity)
        env['BF_TEST_PIPELINE'] = '1'
    one, two = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    r = subprocess.run([sys.executable, helper, one], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    port = 29500 + os.getpid() % 150
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_rank), '--master-addr',
                        '127.0.0.1', '--master-port', str(port), helper, two], cwd=root, env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    a, b = np.load(one), np.load(two)
    assert a['s'].shape == (22, 60, 6)
    for k in ('ts', 's', 'logp'):
        assert np.array_equal(a[k], b[k]), k
    # the reductions over ranks sum in a different order than one rank does: the warm start agrees to rounding, and so
    # does the round that starts from it
    np.testing.assert_allclose(b['step'], a['step'], rtol=1e-13)
    np.testing.assert_allclose(b['metric'], a['metric'], rtol=1e-12)
    np.testing.assert_allclose(b['s2'][:, :5], a['s2'][:, :5], rtol=1e-9, atol=1e-9)


def test_fit_ill_conditioned_designs_match_reference_lstsq():
    """PolyModel.fit against the reference's LAPACK gelsd (modules/poly.py:570) on fixtures recorded from it
    (tests/golden/fit_illcond.npz, make_golden.py:gen_fit_illcond): cubic designs whose column-equilibrated condition
    number is 2.1e5 and 1.0e7, and an exactly rank-deficient one.  The device solve is normal equations + two refinement
    steps on the true residual (include/bfhip.h: bfhip_lstsq): plain normal equations lose cond^2 eps (1e-4 of the
    coefficient scale at cond 1e7), the refined solve agrees with the orthogonal factorisation to 1e-9.  Designs whose
    equilibrated Cholesky pivots fall below 1e-11 are reported (RuntimeWarning) and ridge-regularised: what is pinned
    there is the fitted function, not the coefficients along the null directions."""
    import warnings
    from bayesfast_amd import PolyModel
    from specio import rebuild_poly
    z = np.load(os.path.join(G, 'fit_illcond.npz'))
    orders = ['linear', 'quadratic', 'cubic-2', 'cubic-3']
    for tag, lo, hi, plain_floor in (('ill5', 1e5, 3e5, 0.), ('ill7', 5e6, 2e7, 1e-6)):
        assert lo < float(z[tag + '.cond_equilibrated']) < hi
        ref = rebuild_poly(z, tag + '.poly.')
        scale = max(np.abs(_indep(r['order'], r['coef'])).max() for r in ref['configs'])

        def coef_err(pm):
            return max(np.abs(_indep(c.order, c._coef) - _indep(r['order'], r['coef'])).max()
                       for c, r in zip(pm.configs, ref['configs'])) / scale

        pm = PolyModel(orders, input_size=5, output_size=1, bound_options=dict(use_bound=False))
        with warnings.catch_warnings():
            warnings.simplefilter('error')  # silent: these designs are inside the stated range
            pm.fit(z[tag + '.x'], z[tag + '.y'])
        assert coef_err(pm) < 1e-9, (tag, coef_err(pm))
        f = np.array([pm.fun(x)[0] for x in z[tag + '.xe']])
        np.testing.assert_allclose(f, z[tag + '.f_eval'], rtol=1e-10, atol=1e-10)
        # the refinement is what buys it
        pm0 = PolyModel(orders, input_size=5, output_size=1, bound_options=dict(use_bound=False))
        pm0._N_REFINE = 0
        pm0.fit(z[tag + '.x'], z[tag + '.y'])
        assert coef_err(pm0) > max(plain_floor, 20. * coef_err(pm)), (tag, coef_err(pm0), coef_err(pm))
    # exactly rank deficient (input 3 duplicates input 0): gelsd returns the MINIMUM-NORM coefficients (modules/poly.py:570), and
    # so does the device fit (round 6: truncated eigen-solve of the Gram matrix + refinement) -- the reference's own coefficients
    # to 1e-8, not only its fitted values
    pr = PolyModel('quadratic', input_size=4, output_size=1, bound_options=dict(use_bound=False))
    with pytest.warns(RuntimeWarning, match='rank deficient'):
        pr.fit(z['rankdef.x'], z['rankdef.y'])
    f = np.array([pr.fun(x)[0] for x in z['rankdef.x'][:40]])
    np.testing.assert_allclose(f, z['rankdef.f_fit'][:40], rtol=0, atol=1e-9)
    ref = rebuild_poly(z, 'rankdef.poly.')
    scale = max(np.abs(np.asarray(r['coef'])).max() for r in ref['configs'])
    for c, r in zip(pr.configs, ref['configs']):
        np.testing.assert_allclose(_indep(c.order, c._coef), _indep(r['order'], r['coef']), rtol=0, atol=1e-8 * scale, err_msg=c.order)
    # and against SciPy's gelsd on a fresh rank-deficient design with a weighted fit: a constant input AND a duplicated one
    import scipy.linalg
    rng = np.random.default_rng(3)
    x = rng.normal(size=(60, 5))
    x[:, 4] = x[:, 1]
    x[:, 2] = 0.7
    y = rng.normal(size=(60, 2))
    w = rng.uniform(0.5, 2., size=60)
    p2 = PolyModel('quadratic', input_size=5, output_size=2, bound_options=dict(use_bound=False))
    with pytest.warns(RuntimeWarning, match='rank deficient'):
        p2.fit(x, y, w=w)
    from oracle import oracle as orc
    Ad = np.concatenate([np.ones((60, 1)), x, orc.design_block('quadratic', x)], axis=1) * w[:, None]
    sol = scipy.linalg.lstsq(Ad, y * w[:, None])[0]
    got = np.stack([np.concatenate([p2.configs[0]._coef[o], _indep('quadratic', p2.configs[1]._coef)[o].ravel()]) for o in range(2)], 1)
    np.testing.assert_allclose(got, sol, rtol=0, atol=1e-8 * np.abs(sol).max())


def test_fit_rank_deficient_design_warns_and_regularises():
    """Duplicate columns: gelsd would return a min-norm solution; the device fit warns and solves ridge-regularised
    normal equations (predictions on the fit points still match the data)."""
    from bayesfast_amd import PolyModel
    rng = np.random.default_rng(0)
    x = rng.normal(size=(60, 3))
    x[:, 2] = x[:, 1]  # x_2 == x_1: the columns x_1, x_2 (and their products) are collinear
    y = 1. + x[:, 0] - 2. * x[:, 1] + 0.5 * x[:, 0] * x[:, 1]
    pm = PolyModel('quadratic', input_size=3, output_size=1, bound_options=dict(use_bound=False))
    with pytest.warns(RuntimeWarning):
        pm.fit(x, y[:, None])
    pred = np.array([pm.fun(xx)[0][0] for xx in x[:10]])
    np.testing.assert_allclose(pred, y[:10], rtol=1e-5, atol=1e-5)


def test_config5_128d_cubic_cross_surrogate():
    """BASELINE config 5 shapes (one shard): d = 128, N(0, Sigma) with cond(Sigma) = 1e4 (log-uniform spectrum, random
    rotation, seed 18) plus a cubic perturbation on the first 16 coordinates; surrogate = linear + quadratic +
    cubic-2 + cubic-3 ("cubic-cross") on input_mask range(16): P = 129 + 8256 + 256 + 560 = 9201, the largest fit
    and the widest sampler instantiation (DP = 128, two dimensions per lane).  The target lies in the model
    family, so the device fit must recover it (cf. the reference's tests/test_poly.py:18-26), and NUTS chains on
    the fitted surrogate must follow the oracle."""
    from scipy.stats import special_ortho_group
    from bayesfast_amd import PolyModel, PolyConfig, SurrogateDensity, sample, NTrace
    d, m16 = 128, np.arange(16)
    rs = np.random.RandomState(18)
    R = special_ortho_group.rvs(d, random_state=rs)
    lam = np.exp(np.linspace(0., np.log(1e4), d))
    prec = (R * (1. / lam)) @ R.T                        # Sigma^-1
    c2 = rs.normal(size=(16, 16)) * 0.02
    c3 = rs.normal(size=(16, 16, 16)) * 0.02

    def logp(x):
        x = np.atleast_2d(x)
        z = x[:, :16]
        cub = np.einsum('ni,ij,nj->n', z**2, c2, z)
        for j in range(16):
            for k in range(j + 1, 16):
                for l in range(k + 1, 16):
                    cub += c3[j, k, l] * z[:, j] * z[:, k] * z[:, l]
        return -0.5 * np.einsum('ni,ij,nj->n', x, prec, x) + cub

    su = PolyModel([PolyConfig('linear'), PolyConfig('quadratic'), PolyConfig('cubic-2', input_mask=m16),
                    PolyConfig('cubic-3', input_mask=m16)], input_size=d, output_size=1)
    assert su.n_param == 9201
    den = SurrogateDensity(su)
    rng = np.random.default_rng(5)
    chol = np.linalg.cholesky((R * lam) @ R.T)
    x_fit = rng.normal(size=(su.n_param + 600, d)) @ chol.T
    den.fit(x_fit, logp(x_fit))
    x_new = rng.normal(size=(32, d)) @ chol.T * 0.8
    got = np.array([su.fun(x)[0][0] for x in x_new])
    np.testing.assert_allclose(got, logp(x_new), rtol=1e-6, atol=1e-6 * np.abs(logp(x_new)).max())
    tr = NTrace(n_chain=40, n_iter=12, n_warmup=8, x_0=x_fit[:40] * 0.3, random_generator=12)
    tt = sample(den, tr, verbose=False)
    _compare_chains_with_oracle(tt, den, 12, (0, 17, 39), 12, 8)


def test_sample_with_full_rank_metric_recovers_a_correlated_gaussian():
    """T2 for the full-rank metric through the public entry point: sample(density, NTrace(metric='full')) on an
    exactly quadratic, strongly correlated target; the adapted metric approaches the target covariance and the
    posterior moments come out right (a diagonal metric would need far longer trees here)."""
    from bayesfast_amd import PolyModel, SurrogateDensity, sample, NTrace
    from bayesfast_amd.samplers.sample_trace import _get_metric
    d = 12
    rng = np.random.default_rng(77)
    B = rng.normal(size=(d, d))
    cov = B @ B.T / d + 0.05 * np.eye(d)          # condition number of a few hundred
    prec = np.linalg.inv(cov)
    su = PolyModel('quadratic', input_size=d, output_size=1, bound_options=dict(use_bound=False))
    den = SurrogateDensity(su)
    x_fit = rng.normal(size=(3 * su.n_param, d)) @ np.linalg.cholesky(cov).T
    den.fit(x_fit, -0.5 * np.einsum('ni,ij,nj->n', x_fit, prec, x_fit))
    tr = NTrace(n_chain=512, n_iter=260, n_warmup=160, x_0=x_fit[:512], random_generator=21, metric='full')
    tt = sample(den, tr, verbose=False)
    draws = tt.get().reshape(-1, d)
    assert tt.stat('diverging')[:, 160:].sum() == 0
    assert tt.stat('tree_size')[:, 160:].mean() < 12       # a well-adapted full metric needs short trajectories
    sd = np.sqrt(np.diag(cov))
    n_eff = 512 * 100 / 4.
    assert np.all(np.abs(draws.mean(0)) < 5 * sd / np.sqrt(n_eff))
    emp = np.cov(draws, rowvar=False)
    assert np.max(np.abs(emp - cov) / np.sqrt(np.outer(np.diag(cov), np.diag(cov)))) < 0.04
    m = _get_metric(tt, 'full', from_samples=False)          # mean over chains of the adapted covariances
    assert m.shape == (d, d)
    assert np.max(np.abs(m - cov) / np.sqrt(np.outer(np.diag(cov), np.diag(cov)))) < 0.25


@pytest.mark.parametrize('d,m', [(6, 5), (40, 9), (64, 33)])
def test_multi_output_polymodel_batched_fun_and_jac(d, m):
    """PolyModel.fun / jac / fun_and_jac for output_size > 1 in ONE launch (modules/poly.py:430-503; the masked
    scatter of :474-477, the shared extrapolation bound of :480-503) against the oracle, inside and outside the
    bound; then through the fitted module's own entry points."""
    from bayesfast_amd import PolyModel, PolyConfig
    from bayesfast_amd.device import DevicePolyModel
    from oracle import oracle as orc
    rng = np.random.default_rng(d * 10 + m)
    im_q = np.sort(rng.choice(d, size=max(2, d // 2), replace=False))
    om_q = np.sort(rng.choice(m, size=max(1, m // 2), replace=False))
    om_q2 = np.setdiff1d(np.arange(m), om_q)[:max(1, m // 3)]
    nq = im_q.size
    cq = np.zeros((om_q.size, nq, nq))
    iu = np.triu_indices(nq)
    for q in range(om_q.size):
        cq[q][iu] = rng.normal(size=iu[0].size) * 0.2
    cq2 = np.zeros((om_q2.size, d, d))
    iud = np.triu_indices(d)
    for q in range(om_q2.size):
        cq2[q][iud] = rng.normal(size=iud[0].size) * 0.1
    xs = rng.normal(size=(200, d))
    mu = xs.mean(0)
    hess = np.linalg.inv(np.atleast_2d(np.cov(xs, rowvar=False)))
    alpha = float(np.sqrt(np.einsum('ij,jk,ik->i', xs - mu, hess, xs - mu)).max()) * 0.8
    poly = dict(input_size=d, output_size=m, use_bound=True, mu=mu, hess=hess, alpha=alpha, f_mu=rng.normal(size=m),
                configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.arange(m), coef=rng.normal(size=(m, d + 1))),
                         dict(order='quadratic', input_mask=im_q, output_mask=om_q, coef=cq),
                         dict(order='quadratic', input_mask=np.arange(d), output_mask=om_q2, coef=cq2)])
    x = np.concatenate([rng.normal(size=(70, d)) * 0.6, rng.normal(size=(19, d)) * 2.5])  # ragged tile, in and out of bound
    beta = np.sqrt(np.einsum('ij,jk,ik->i', x - mu, hess, x - mu))
    assert (beta > alpha).any() and (beta < alpha).any()
    f0, j0 = orc.poly_fun_and_jac(poly, x)
    f, j = DevicePolyModel(poly).fun_and_jac(x)
    sc = np.abs(f0).max()
    np.testing.assert_allclose(f.cpu().numpy(), f0, rtol=1e-11, atol=1e-11 * sc)
    np.testing.assert_allclose(j.cpu().numpy(), j0, rtol=1e-10, atol=1e-10 * np.abs(j0).max())
    f1, _ = DevicePolyModel(poly).fun_and_jac(x[3], jac=False)
    np.testing.assert_allclose(f1.cpu().numpy(), f0[3], rtol=1e-11, atol=1e-11 * sc)
    # a fitted multi-output module: fit (one factorisation per config group), then batched evaluation
    pm = PolyModel([PolyConfig('linear'), PolyConfig('quadratic', input_mask=im_q, output_mask=om_q)], input_size=d,
                   output_size=m)
    xf = rng.normal(size=(2 * pm.n_param + 20, d))
    yf = rng.normal(size=(xf.shape[0], m)) * 0.01 + xf[:, :1] * np.arange(1, m + 1)[None, :]
    yf[:, om_q] += (xf[:, im_q[0]] * xf[:, im_q[1]])[:, None]
    pm.fit(xf, yf, logp=yf[:, 0])
    fb, jb = pm.fun_and_jac_batch(x[:40])
    f_ref, j_ref = orc.poly_fun_and_jac(pm.poly_spec(), x[:40])
    np.testing.assert_allclose(fb.cpu().numpy(), f_ref, rtol=1e-10, atol=1e-10 * np.abs(f_ref).max())
    np.testing.assert_allclose(jb.cpu().numpy(), j_ref, rtol=1e-9, atol=1e-9 * np.abs(j_ref).max())
    f_one, j_one = pm.fun_and_jac(x[5])
    np.testing.assert_allclose(f_one[0], f_ref[5], rtol=1e-10, atol=1e-10 * np.abs(f_ref).max())
    np.testing.assert_allclose(j_one[0], j_ref[5], rtol=1e-9, atol=1e-9 * np.abs(j_ref).max())


def test_surrogate_plus_chi2_pipeline_matches_reference_density():
    """SURVEY 8f-1 against a fixture recorded from the reference (tests/golden/pipeline.npz, make_golden.py:gen_pipeline):
    ``Density(module_list=[model, chi2], surrogate_list=[PolyModel(linear, quadratic, cubic-2, cubic-3)])``, evaluated
    with ``use_surrogate=True`` (core/density.py:487-566).  (1) The reference's fitted coefficients through
    bfhip_polymodel_eval + bfhip_chi2_stage: logp and grad of the pipeline, inside and outside the bound.  (2) Our own
    fit of the same data gives the same pipeline density."""
    import os
    from specio import rebuild_poly
    from bayesfast_amd import PolyModel, PolyConfig, Chi2PipelineDensity
    from bayesfast_amd.device import DevicePolyModel
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pipeline.npz'))
    poly = rebuild_poly(z, 'poly.')
    d, m = poly['input_size'], poly['output_size']
    f, j = DevicePolyModel(poly).fun_and_jac(z['xt'])
    np.testing.assert_allclose(f.cpu().numpy(), z['su_f'], rtol=1e-10, atol=1e-10)

    class _Fixed(PolyModel):  # a PolyModel carrying the reference's coefficients
        def poly_spec(self, use_bound=None):
            return poly

    cfgs = [PolyConfig('linear'), PolyConfig('quadratic'), PolyConfig('cubic-2', input_mask=[0, 2, 3, 5], output_mask=[0, 1, 3]),
            PolyConfig('cubic-3', input_mask=[1, 2, 4, 5, 6], output_mask=[1, 2, 4])]
    ref = Chi2PipelineDensity(_Fixed(cfgs, input_size=d, output_size=m), z['ydat'], prec=z['prec'])
    lp, g = ref.logp_and_grad(z['xt'])
    np.testing.assert_allclose(lp, z['logp'], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(g, z['grad'], rtol=1e-9, atol=1e-9 * np.abs(z['grad']).max())
    # the first implementation (f and the (n, m, d) Jacobians, then the chi-square stage) is a second, independent route
    lpd, gd = ref.logp_and_grad_device(z['xt'])
    np.testing.assert_allclose(lpd.cpu().numpy(), lp, rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(gd.cpu().numpy(), g, rtol=1e-10, atol=1e-10 * np.abs(z['grad']).max())
    lp1, g1 = ref.logp_and_grad(z['xt'][30])
    np.testing.assert_allclose([lp1], [z['logp'][30]], rtol=1e-10)
    np.testing.assert_allclose(g1, z['grad'][30], rtol=1e-9, atol=1e-9 * np.abs(z['grad']).max())
    # our fit of the same (x, y) data: same density up to the fit's accuracy
    su = PolyModel(cfgs, input_size=d, output_size=m)
    su.fit(z['x_fit'], z['y_fit'], z['logp_fit'])
    lp2, g2 = Chi2PipelineDensity(su, z['ydat'], prec=z['prec']).logp_and_grad(z['xt'])
    np.testing.assert_allclose(lp2, z['logp'], rtol=1e-6, atol=1e-6 * np.abs(z['logp']).max())
    np.testing.assert_allclose(g2, z['grad'], rtol=1e-5, atol=1e-6 * np.abs(z['grad']).max())


def test_sample_accepts_a_reference_style_density_and_fit_takes_var_dicts():
    """The Python seam as Recipe uses it: Density.fit(var_dicts) (core/density.py:813-838), and sample() on
    a density object that only exposes the reference's attribute names (read through bayesfast_amd/adapters.py)."""
    from types import SimpleNamespace
    from bayesfast_amd import PolyModel, SurrogateDensity, sample
    from test_host_api import _RefDensity
    d = 5
    rng = np.random.default_rng(12)
    Pm = np.eye(d) + 0.2 * rng.normal(size=(d, d)) / np.sqrt(d)
    Pm = Pm @ Pm.T
    xf = rng.normal(size=(90, d)) * 1.5
    lp = -0.5 * np.einsum('ij,jk,ik->i', xf, Pm, xf)
    den = SurrogateDensity(PolyModel('quadratic', input_size=d, output_size=1), input_scales=np.stack([-9. * np.ones(d), 9. * np.ones(d)], 1),
                           hard_bounds=True, decay_options=dict(use_decay=True))
    den.fit(xf, lp)
    den2 = SurrogateDensity(PolyModel('quadratic', input_size=d, output_size=1), input_scales=np.stack([-9. * np.ones(d), 9. * np.ones(d)], 1),
                            hard_bounds=True, decay_options=dict(use_decay=True))
    # var_dicts: x under the input variable's name, the true log-density under density_name
    den2.input_vars, den2.density_name = ('x',), 'logp'
    den2.fit([SimpleNamespace(_fun={'x': x, 'logp': np.array([l])}) for x, l in zip(xf, lp)])
    a, b = den.spec(), den2.spec()
    for c0, c1 in zip(a['poly']['configs'], b['poly']['configs']):
        assert np.array_equal(c0['coef'], c1['coef'])
    assert np.array_equal(a['decay_hess'], b['decay_hess']) and a['decay_alpha2'] == b['decay_alpha2']
    opts = {'n_chain': 12, 'n_iter': 40, 'n_warmup': 25, 'random_generator': 3}
    t0 = sample(den, dict(opts), verbose=False)
    t1 = sample(_RefDensity(den), dict(opts), verbose=False)
    assert np.array_equal(t0.samples, t1.samples) and np.array_equal(t0.samples_original, t1.samples_original)
    assert np.array_equal(t0.get(return_type='logp'), t1.get(return_type='logp'))
