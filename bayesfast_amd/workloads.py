"""Synthetic workloads of BASELINE.json (shapes, distributions and seeds fixed in SURVEY.md section 8d).

Only NumPy: these build the density *spec* (plain dict) that both the device path and the test oracle consume.
"""
import numpy as np

__all__ = ['correlated_gaussian_spec', 'banana_logp', 'funnel_logp', 'planck_like_logp', 'des_like_pipeline', 'random_pipeline_spec', 'sobol_normal',
           'B_STEP_BYTES', 'flops_per_leapfrog', 'flops_per_leapfrog_spec']


def B_STEP_BYTES(d):
    """Algorithmic HBM bytes of one leapfrog step of one chain: read q, p, grad and write them back
    (48 d) plus logp/energy in and out (32).  SURVEY.md section 8(d)."""
    return 48 * d + 32


def flops_per_leapfrog(d, use_bound=True):
    """Algorithmic flops of one leapfrog step of one chain: S x (2 d^2) for value+gradient of the
    quadratic form and H (x - mu) (2 d^2) for the bound test; the O(d) tail is ignored."""
    return (4 if use_bound else 2) * d * d


def correlated_gaussian_spec(d=64, seed=123, n_fit_mult=2, fit_seed=7, fit_scale=1.5, scales=None):
    """Config-2 family of SURVEY.md section 8(d) at dimension d: target logp = -x^T P x / 2 with P = L L^T,
    L = I + 0.3 tril(G, -1) / sqrt(d), G ~ N(0, 1) from default_rng(seed).

    The surrogate is ``PolyModel('quadratic')`` (configs linear + quadratic, bound on with alpha_p = 100,
    modules/poly.py:185-186,232-260).  The target is exactly quadratic, so the least-squares solution is
    known in closed form (linear part 0, upper-triangular a[j,k] from -P/2); the bound statistics
    mu, H = inv(cov), alpha = max Mahalanobis radius (modules/poly.py:268-276) are those of the
    n_fit_mult * P fit points x ~ N(0, fit_scale^2 I) from default_rng(fit_seed), and f_mu is the surrogate at the
    fit point of largest logp (center_max).

    fit_scale = 1.5 makes the training set broader than the posterior, as the first rounds of a recipe are: the
    alpha-ellipsoid then contains the posterior with a wide margin.  With a training set as tight as the
    posterior (fit_scale = 1) the linear extrapolation outside the ellipsoid (modules/poly.py:480-503) wins in 64
    dimensions: its radial density beta^63 exp(-c beta) still rises just outside the bound, chains leak out and
    stay there (99 % of 4096 chains after 900 iterations, every leapfrog then paying the second evaluation at the
    projected point) -- a property of the reference's bound without its decay term, not a sampling workload."""
    rng = np.random.default_rng(seed)
    L = np.eye(d) + 0.3 * np.tril(rng.normal(size=(d, d)), -1) / np.sqrt(d)
    P = L @ L.T
    if scales is not None:  # x_i -> x_i / s_i: standard deviations spread by the given factors (anisotropic target)
        sinv = 1. / np.asarray(scales, dtype=np.float64)
        P = P * np.outer(sinv, sinv)
    A = -0.5 * P
    quad = np.zeros((d, d))
    iu = np.triu_indices(d)
    quad[iu] = A[iu] * np.where(iu[0] == iu[1], 1., 2.)
    n_param = 1 + d + d * (d + 1) // 2
    x = fit_scale * np.random.default_rng(fit_seed).normal(size=(n_fit_mult * n_param, d))
    if scales is not None:
        x = x * np.asarray(scales, dtype=np.float64)
    mu = np.mean(x, axis=0)
    hess = np.linalg.inv(np.atleast_2d(np.cov(x, rowvar=False)))
    beta = np.einsum('ij,jk,ik->i', x - mu, hess, x - mu)**0.5
    alpha = float(np.max(beta))
    logp_fit = -0.5 * np.einsum('ij,jk,ik->i', x, P, x)
    xm = x[np.argmax(logp_fit)]
    f_mu = float(-0.5 * xm @ P @ xm)
    poly = dict(input_size=d, output_size=1, use_bound=True, mu=mu, hess=hess, alpha=alpha, f_mu=np.array([f_mu]),
                configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.array([0]), coef=np.zeros((1, d + 1))),
                         dict(order='quadratic', input_mask=np.arange(d), output_mask=np.array([0]), coef=quad[None])])
    spec = dict(d=d, ranges=None, hard_bounds=None, su_lo=None, su_diff=None, poly=poly, use_decay=False)
    return spec, np.linalg.inv(P)


def banana_logp(d=64, q=0.01, seed=0):
    """Config 3 (headline) target of SURVEY.md section 8(d): the banana of examples/banana-gbs.ipynb cell 3 at
    D = d, Q = q, rotated by ``special_ortho_group.rvs(d)`` drawn after ``np.random.seed(seed)``.
    Returns a vectorised ``logp(x (..., d)) -> (...)`` evaluated on the host (the "true model")."""
    from scipy.stats import special_ortho_group
    A = special_ortho_group.rvs(d, random_state=np.random.RandomState(seed))

    def logp(x):
        z = np.asarray(x, dtype=np.float64) @ A.T
        return -np.sum((z[..., ::2]**2 - z[..., 1::2])**2 / q + (z[..., ::2] - 1)**2, axis=-1)

    return logp


def funnel_logp(d=64, a=1., b=0.5):
    """Config 4 target of SURVEY.md section 8(d): the funnel of examples/funnel-gbs.ipynb cell 3 (a = 1, b = 0.5, D = d)."""
    def logp(x):
        x = np.asarray(x, dtype=np.float64)
        return (-x[..., 0]**2 / (2 * a**2) - np.sum(x[..., 1:]**2, axis=-1) / (2 * np.exp(2 * b * x[..., 0])) -
                (d - 1) * b * x[..., 0])

    return logp


def planck_like_logp(d=128, seed=18, n_cubic=16, amp=0.02):
    """Config 5 target of SURVEY.md section 8(d) (the reference has no Planck likelihood; examples/planck_18_sterile.ipynb is a
    stub): N(0, Sigma) with cond(Sigma) = 1e4 (log-uniform spectrum, random rotation, RandomState(seed)) plus a cubic
    perturbation -- cubic-2 and cubic-3 ("cubic-cross") terms -- on the first n_cubic coordinates.  Returns
    (logp(x (..., d)) -> (...), chol) with chol chol^T = Sigma (to draw fit points from the Gaussian part)."""
    from scipy.stats import special_ortho_group
    rs = np.random.RandomState(seed)
    R = special_ortho_group.rvs(d, random_state=rs)
    lam = np.exp(np.linspace(0., np.log(1e4), d))
    prec = (R * (1. / lam)) @ R.T
    c2 = rs.normal(size=(n_cubic, n_cubic)) * amp
    c3 = rs.normal(size=(n_cubic,) * 3) * amp
    j, k, l = np.meshgrid(*[np.arange(n_cubic)] * 3, indexing='ij')
    c3 = np.where((j < k) & (k < l), c3, 0.)

    def logp(x):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        z = x[:, :n_cubic]
        cub = np.einsum('ni,ij,nj->n', z**2, c2, z) + np.einsum('jkl,nj,nk,nl->n', c3, z, z, z, optimize=True)
        return -0.5 * np.einsum('ni,ij,nj->n', x, prec, x) + cub

    return logp, np.linalg.cholesky((R * lam) @ R.T)


def des_like_pipeline(d=27, m=457, n_nonlinear=9, n_prior=13, seed=1):
    """The shape of examples/des-y1-w-cosmosis.ipynb (cells 9-18) with a synthetic theory model (the notebook's is CosmoSIS,
    which nothing here can run): d = 27 parameters in a box ``para_range`` with hard bounds, a theory vector of m = 457
    whitened data points that is linear in all parameters and quadratic (plus a small non-polynomial term, so that the
    surrogate is not exact) in ``nonlinear`` = 9 of them, like = -|theory - data|^2 / 2 + norm, and a Gaussian prior on 13
    of the parameters.  Returns a dict: para_range (d, 2), nonlinear, model(x (n, d)) -> (n, m), data (m,), norm, prior_mu /
    prior_prec (d,), prior_c0, x_true, and logp(x) the true log-density."""
    rng = np.random.default_rng(seed)
    lo = -1. - rng.uniform(size=d)
    hi = 1. + rng.uniform(size=d)
    para_range = np.stack([lo, hi], 1)
    nonlinear = np.sort(rng.choice(d, n_nonlinear, replace=False))
    A = rng.normal(size=(m, d))
    B = rng.normal(size=(m, n_nonlinear, n_nonlinear)) * 0.5
    t0 = rng.normal(size=m) * 3.
    u_true = rng.uniform(0.35, 0.65, size=d)

    def model(x):
        u = (np.atleast_2d(np.asarray(x, dtype=np.float64)) - lo) / (hi - lo)
        z = u[:, nonlinear]
        return t0 + u @ A.T + np.einsum('ojk,nj,nk->no', B, z, z, optimize=True) + 0.1 * np.sin(3. * z[:, :1])

    x_true = lo + u_true * (hi - lo)
    data = model(x_true)[0] + rng.normal(size=m)
    norm = -0.5 * m * np.log(2 * np.pi)
    pidx = np.sort(rng.choice(d, n_prior, replace=False))
    prior_mu, prior_prec = np.zeros(d), np.zeros(d)
    sig = 0.05 * (hi - lo)[pidx]
    prior_mu[pidx] = x_true[pidx] + 0.3 * sig * rng.normal(size=n_prior)
    prior_prec[pidx] = 1. / sig**2
    prior_c0 = float(-0.5 * np.sum(np.log(2 * np.pi * sig**2)))

    def logp(x):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        r = model(x) - data
        return -0.5 * np.sum(r * r, axis=1) + norm + prior_c0 - 0.5 * np.sum(prior_prec * (x - prior_mu)**2, axis=1)

    return dict(d=d, m=m, para_range=para_range, nonlinear=nonlinear, model=model, data=data, norm=float(norm), prior_mu=prior_mu,
                prior_prec=prior_prec, prior_c0=prior_c0, x_true=x_true, logp=logp)


def random_pipeline_spec(m, d, nq, seed=0, bound=True, transform=True):
    """A pipeline-density spec (DeviceDensity / the oracle take it) with random coefficients, no fit: m outputs, linear in all d
    inputs and quadratic in nq of them, unit precision, a prior on every second input, optionally behind the box transform
    with hard bounds and surrogate input scales; the bound's ellipsoid from a random cloud (bound=False: far away)."""
    rng = np.random.default_rng(seed)
    mask = np.sort(rng.choice(d, nq, replace=False))
    lin = rng.normal(size=(m, d + 1))
    quad = np.zeros((m, nq, nq))
    iu = np.triu_indices(nq)
    quad[:, iu[0], iu[1]] = rng.normal(size=(m, iu[0].size)) * 0.3
    x = rng.normal(size=(400, d)) * 0.3 + 0.5
    mu = x.mean(0)
    hess = np.linalg.inv(np.cov(x, rowvar=False))
    alpha = float(np.max(np.einsum('ij,jk,ik->i', x - mu, hess, x - mu)**0.5)) * (1. if bound else 1e6)
    poly = dict(input_size=d, output_size=m, use_bound=True, mu=mu, hess=hess, alpha=alpha, f_mu=np.zeros(m),
                configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.arange(m), coef=lin),
                         dict(order='quadratic', input_mask=mask, output_mask=np.arange(m), coef=quad)])
    u0 = np.full(d, 0.5)
    f0 = lin[:, 0] + lin[:, 1:] @ u0 + np.einsum('ojk,j,k->o', quad, u0[mask], u0[mask])
    rg = np.stack([-np.ones(d), np.ones(d)], 1) * 2.
    return dict(d=d, ranges=rg if transform else None, hard_bounds=np.ones((d, 2), np.uint8) if transform else None,
                su_lo=rg[:, 0] if transform else None, su_diff=(rg[:, 1] - rg[:, 0]) if transform else None, poly=poly, use_decay=False,
                chi2=dict(y=f0 + rng.normal(size=m), prec_diag=np.ones(m), logp0=0.),
                prior=dict(mu=np.zeros(d), prec_diag=np.where(np.arange(d) % 2, 4., 0.), c0=0.))


def decay_shares_bound(spec):
    """The decay term's Hessian and centre are the bound's arrays, bit for bit (what ``bfhip_density_upload`` compares)."""
    poly = spec.get('poly') or {}
    if not (spec.get('use_decay') and poly.get('use_bound')) or poly.get('hess') is None or spec.get('decay_hess') is None:
        return False
    return bool(np.array_equal(np.asarray(spec['decay_hess']), np.asarray(poly['hess'])) and
                np.array_equal(np.asarray(spec['decay_mu']), np.asarray(poly['mu'])))


def flops_per_leapfrog_spec(spec):
    """Algorithmic flops of one leapfrog step on a density spec: one d x d matvec (2 d^2) each for S x, the bound's
    H (x - mu) and the decay term's H_d (x - mu_d) -- unless the decay term's matrix and centre ARE the bound's (``decay_shares_bound``:
    the usual case, both come from the fit points; one product then serves both) -- plus the cubic configs' contractions on their
    masked inputs (cubic-2: two n2 x n2 matvecs; cubic-3: one n3^3 contraction); the O(d) tail is ignored."""
    d = int(spec['d'])
    poly = spec['poly']
    if decay_shares_bound(spec):
        spec = dict(spec, use_decay=False)
    orders = {c['order']: np.asarray(c['input_mask']).size for c in poly['configs']}
    if spec.get('chi2') is not None:
        # pipeline density: f = C phi and C^T r, C (m, n_monomials) with the configs' masks (bfhip_pld.h), plus the matvecs of
        # the bound and of the decay term; a full precision matrix is folded into C at upload and costs nothing per step
        nf = 1
        for c in poly['configs']:
            n = np.asarray(c['input_mask']).size
            nf += {'linear': n, 'quadratic': n * (n + 1) // 2, 'cubic-2': n * n, 'cubic-3': n * (n - 1) * (n - 2) // 6}[c['order']]
        f = 4 * int(poly['output_size']) * nf
        if poly.get('use_bound'):
            f += 2 * d * d
        if spec.get('use_decay'):
            f += 2 * d * d
        return f
    f = 0
    if 'quadratic' in orders:
        f += 2 * d * d
    if poly.get('use_bound') and len(orders) > (1 if 'linear' in orders else 0):
        f += 2 * d * d
    if spec.get('use_decay'):
        f += 2 * d * d
    if 'cubic-2' in orders:
        f += 4 * orders['cubic-2']**2
    if 'cubic-3' in orders:
        f += 2 * orders['cubic-3']**3
    return f


def sobol_normal(n, d, seed=0):
    """n rows of a scrambled Sobol sequence mapped to N(0, I_d) (the reference's fit points and chain starts,
    core/sample.py:111-112, come from its own Sobol generator; any low-discrepancy normal set serves)."""
    from scipy.stats import qmc, norm
    m = int(np.ceil(np.log2(max(n, 2))))
    u = qmc.Sobol(d, scramble=True, seed=seed).random_base2(m)[:n]
    return norm.ppf(np.clip(u, 1e-12, 1 - 1e-12))
