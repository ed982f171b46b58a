"""BASELINE config 1: the 2-d donut of the reference's examples/2d-donut.ipynb (cell 4) with 4 chains --
logp = -(|x| - 5)^2 / 0.5; a linear surrogate of m = |x| in the OptimizeStep, then a quadratic one without the bound in ten
SampleSteps, decay term on.  ``build_recipe(bf, ...)`` writes it against whatever ``bayesfast`` package it is given: the
reference as imported (tests/golden/make_golden.py: fixture vi) or the reference patched by bayesfast_amd.integrate
(tests/test_integrate_reference.py)."""
import numpy as np

A, B = 5., 0.5


def f_0(x):
    return np.linalg.norm(x, 2, -1)


def f_1(m):
    return -(m - A)**2 / B


def j_1(m):
    return -2 * (m - A) / B


def true_logp(x):
    return f_1(f_0(np.asarray(x, dtype=np.float64)))


def build_recipe(bf, n_chain=4, n_iter=1000, n_warmup=500, poly_model=None, likelihood=None, sample_repeat=(5, 5),
                 n_backend=4):
    bf.utils.random.set_generator(2)
    bf.utils.parallel.set_backend(n_backend)
    poly_model = poly_model or bf.modules.PolyModel
    module_0 = bf.Module(fun=f_0, input_vars='x', output_vars='m')
    module_1 = likelihood or bf.Module(fun=f_1, jac=j_1, input_vars='m', output_vars='logp')
    density = bf.Density(module_list=[module_0, module_1], input_shapes=[2], input_vars='x', density_name='logp')
    density.set_decay_options(use_decay=True)
    surro_0 = poly_model('linear', input_size=2, output_size=1, input_vars='x', output_vars='m')
    surro_1 = poly_model('quadratic', input_size=2, output_size=1, input_vars='x', output_vars='m')
    surro_1.set_bound_options(use_bound=False)
    x_0 = bf.utils.sobol.multivariate_normal([10, 10], np.eye(2), 20)
    trace = {'n_chain': n_chain, 'n_iter': n_iter, 'n_warmup': n_warmup}
    opt_0 = bf.recipe.OptimizeStep(surrogate_list=surro_0, x_0=x_0, sample_trace=dict(trace))
    sam_0 = bf.recipe.SampleStep(surrogate_list=surro_1, alpha_n=5, reuse_samples=0, sample_trace=dict(trace), logp_cutoff=False)
    sam_1 = bf.recipe.SampleStep(surrogate_list=surro_1, alpha_n=5, reuse_samples=1, sample_trace=dict(trace), logp_cutoff=False)
    return bf.recipe.Recipe(density=density, optimize=opt_0, sample=[sam_0, sam_1], post={}, sample_repeat=list(sample_repeat))


def ring_statistics(samples):
    """Per-step summary of (n, 2) samples of the ring: radius mean / sd, angular coverage (mean resultant length of the
    angle: 0 for a uniformly covered ring), mean of x."""
    s = np.asarray(samples, dtype=np.float64).reshape(-1, 2)
    r = np.linalg.norm(s, axis=-1)
    th = np.arctan2(s[:, 1], s[:, 0])
    return np.array([r.mean(), r.std(), np.hypot(np.cos(th).mean(), np.sin(th).mean()), s[:, 0].mean(), s[:, 1].mean()])


def own_refit_loop(x_fit0, step_size0=None, n_steps=10, n_chain=4, n_iter=1000, n_warmup=500, alpha_n=5, seed=2):
    """The SampleSteps of the recipe above through THIS package's own API (no reference, no Recipe): fit the quadratic
    surrogate of m = |x| (bound off, decay on) on the points handed over, sample logp = -(m - 5)^2 / 0.5 of it on the device
    (the likelihood is chained in the kernel: GaussianLink), pick alpha_n * n_param points of the result by their logq
    (SystematicResampler, core/recipe.py:1074), evaluate the true model there, refit, warm-start step size and metric from
    the round before (core/recipe.py:1033-1045).  Returns the per-step ring statistics (n_steps, 5) and the last TraceTuple."""
    import bayesfast_amd as bfa
    from bayesfast_amd.samplers import _get_step_size, _get_metric
    su = bfa.PolyModel('quadratic', input_size=2, output_size=1, bound_options=dict(use_bound=False))
    den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True), link=bfa.GaussianLink(A, 2. / B))
    n_eval = alpha_n * su.n_param
    x_fit = np.asarray(x_fit0, dtype=np.float64)
    rings, tt, pool = [], None, x_fit
    for i in range(n_steps):
        den.fit(x_fit, true_logp(x_fit), y=f_0(x_fit))
        kw = dict(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, x_0=pool, random_generator=seed + i)
        if tt is not None:
            kw.update(step_size=_get_step_size(tt), metric=_get_metric(tt, 'diag'))
        elif step_size0 is not None:
            kw.update(step_size=step_size0)
        tt = bfa.sample(den, kw, verbose=False)
        tt.gather()
        rings.append(ring_statistics(tt.get()))
        x_fit, _, _ = bfa.select_fit_points(tt, None, true_logp, n_eval, logp_cutoff=False)
        pool = tt.get()
    return np.array(rings), tt
