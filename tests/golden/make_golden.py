#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference (h3jia/bayesfast).

Runs only in the build container, where /root/reference exists.  It copies the reference package to a
scratch directory, builds its four Cython extensions with a plain setuptools script, installs two
environment shims (removed NumPy aliases; a stub numdifftools) and then records inputs/outputs of the
reference's own functions on the hot path.  Nothing of the reference's source enters this repository:
the .npz files hold data only (inputs, coefficients the reference fitted, outputs, logged random draws).

Usage:  python tests/golden/make_golden.py [--ref /root/reference] [--work /tmp/bfref]
"""
import argparse
import os
import shutil
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from specio import flatten_spec, flatten_poly  # noqa: E402

BUILD_EXT = r'''
import numpy as np
from setuptools import setup, Extension
from Cython.Build import cythonize
names = ['bayesfast/modules/_poly', 'bayesfast/transforms/_constraint', 'bayesfast/utils/_sobol',
         'bayesfast/utils/_cubic']
exts = [Extension(n.replace('/', '.'), [n + '.pyx'], include_dirs=[np.get_include()],
                  extra_compile_args=['-fopenmp', '-O3'], extra_link_args=['-fopenmp']) for n in names]
setup(ext_modules=cythonize(exts, language_level='3'))
'''


def prepare_reference(ref, work):
    pkg = os.path.join(work, 'bayesfast')
    if not os.path.exists(os.path.join(pkg, 'modules')) or not any(
            f.endswith('.so') for f in os.listdir(os.path.join(pkg, 'modules'))):
        os.makedirs(work, exist_ok=True)
        if os.path.exists(pkg):
            shutil.rmtree(pkg)
        shutil.copytree(os.path.join(ref, 'bayesfast'), pkg)
        with open(os.path.join(work, 'build_ext.py'), 'w') as f:
            f.write(BUILD_EXT)
        subprocess.check_call([sys.executable, 'build_ext.py', 'build_ext', '--inplace'], cwd=work,
                              stdout=subprocess.DEVNULL)
    # shim 1: NumPy aliases removed in 1.24 that the reference still uses
    for n, t in (('int', int), ('float', float)):
        if not hasattr(np, n):
            setattr(np, n, t)
    # shim 2: numdifftools is not installed; only Laplace (OptimizeStep) and the reference's tests use it
    nd = types.ModuleType('numdifftools')

    def _central(f, x, h=1e-6):
        x = np.atleast_1d(np.asarray(x, dtype=float))
        f0 = np.atleast_1d(f(x))
        J = np.empty((f0.size, x.size))
        for i in range(x.size):
            e = np.zeros_like(x)
            e[i] = h
            J[:, i] = (np.atleast_1d(f(x + e)) - np.atleast_1d(f(x - e))) / (2 * h)
        return J

    class Gradient:
        def __init__(self, f, *a, **k):
            self.f = f

        def __call__(self, x):
            return _central(self.f, x)[0]

    class Jacobian(Gradient):
        def __call__(self, x):
            return _central(self.f, x)

    class Hessian(Gradient):
        def __call__(self, x):
            return _central(lambda y: _central(self.f, y)[0], x, 1e-4)

    class Hessdiag(Gradient):
        def __call__(self, x):
            return np.diag(Hessian(self.f)(x))

    nd.Gradient, nd.Jacobian, nd.Hessian, nd.Hessdiag = Gradient, Jacobian, Hessian, Hessdiag
    sys.modules['numdifftools'] = nd
    sys.path.insert(0, work)
    import bayesfast  # noqa: F401
    return bayesfast


# ---------------------------------------------------------------------------------------------------

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bayesfast_amd.adapters import poly_spec_from_reference, density_spec_from_reference  # noqa: E402  (the shipped adapter)


class LoggingGenerator:
    """Wraps a np.random.Generator and records the draws the samplers consume."""

    def __init__(self, g):
        self._g = g
        self.normals = []
        self.uniforms = []

    def normal(self, loc=0., scale=1., size=None):
        v = self._g.normal(loc, scale, size=size)
        self.normals.extend(np.atleast_1d(v).tolist())
        return v

    def uniform(self):
        v = self._g.uniform()
        self.uniforms.append(float(v))
        return v

    def integers(self, *a, **k):
        return self._g.integers(*a, **k)

    @property
    def bit_generator(self):
        return self._g.bit_generator


def make_density(bf, d, seed, scales=False, decay=False, cubic=0.1, n_fit_mult=3, surrogate='quadratic',
                 su_scales=False):
    """A reference Density whose single module is replaced by a fitted PolyModel logp surrogate."""
    from bayesfast.modules import PolyModel
    rng = np.random.default_rng(seed)
    Pm = np.eye(d) + 0.3 * rng.normal(size=(d, d)) / np.sqrt(d)
    Pm = Pm @ Pm.T

    def f(x):
        return -0.5 * x @ Pm @ x - cubic * np.sum(x**3) / d

    def j(x):
        return (-(Pm @ x) - 3 * cubic * x**2 / d)[None]

    kw = {}
    if scales:
        lo, hi = -6. - rng.uniform(size=d), 7. + rng.uniform(size=d)
        kw['input_scales'] = np.stack([lo, hi], 1)
        hb = np.zeros((d, 2), int)
        hb[0] = (1, 1)
        hb[1 % d] = (1, 0)
        hb[2 % d] = (0, 1)
        kw['hard_bounds'] = hb
    skw = {}
    if su_scales:
        skw['input_scales'] = np.stack([-2. - rng.uniform(size=d), 3. + rng.uniform(size=d)], 1)
    su = PolyModel(surrogate, input_size=d, output_size=1, input_vars='x', output_vars='logp', **skw)
    den = bf.Density(density_name='logp', module_list=[bf.Module(fun=f, jac=j, input_vars='x', output_vars='logp')],
                     surrogate_list=[su], input_vars='x', input_shapes=d,
                     decay_options=dict(use_decay=bool(decay)), **kw)
    xs = rng.normal(size=(n_fit_mult * int(su.n_param), d))
    vds = [den.fun(x, original_space=True, use_surrogate=False) for x in xs]
    den.fit(vds)
    return den, rng


def gen_poly_kernels(bf, out):
    from bayesfast.modules import _poly
    rng = np.random.default_rng(11)
    z = {}
    for n in (2, 3, 5, 8):  # n = 1 would overflow the reference's size_t loop bound n - 2 (_poly.pyx:98)
        m = 2
        x = rng.normal(size=n)
        z['n%d.x' % n] = x
        xs = rng.normal(size=(4, n))
        z['n%d.xs' % n] = xs
        for order, npk, shape in (('quadratic', n * (n + 1) // 2, (n, n)), ('cubic_2', n * n, (n, n)),
                                  ('cubic_3', n * (n - 1) * (n - 2) // 6, (n, n, n))):
            a = rng.normal(size=(m, npk))
            coef = np.zeros((m,) + shape)
            for i in range(m):
                getattr(_poly, '_set_' + order)(np.ascontiguousarray(a[i]), coef[i], n)
            f = np.empty(m)
            jj = np.empty((m, n))
            getattr(_poly, '_' + order + '_f')(x, coef, f, m, n)
            getattr(_poly, '_' + order + '_j')(x, coef, jj, m, n)
            A = np.empty((4, npk))
            getattr(_poly, '_lsq_' + order)(xs, A, 4, n)
            k = 'n%d.%s.' % (n, order)
            z[k + 'a'], z[k + 'coef'], z[k + 'f'], z[k + 'j'], z[k + 'lsq'] = a, coef, f, jj, A
    np.savez_compressed(os.path.join(out, 'poly_kernels.npz'), **z)


def gen_constraint(bf, out):
    from bayesfast.transforms import _constraint as cs
    rng = np.random.default_rng(12)
    n = 8
    ranges = np.stack([-1 - rng.uniform(size=n), 2 + rng.uniform(size=n)], 1)
    hb = np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * 2, dtype=np.uint8)
    xt = rng.normal(size=(6, n)) * 2
    z = dict(ranges=ranges, hard_bounds=hb, x_trans=xt)
    for nm in ('f', 'j', 'jj'):
        o = np.empty_like(xt)
        getattr(cs, '_to_original_%s2' % nm)(xt, ranges, o, hb, n, xt.shape[0])
        z['to_' + nm] = o
    xo = z['to_f']
    for nm in ('f', 'j', 'jj'):
        o = np.empty_like(xo)
        getattr(cs, '_from_original_%s2' % nm)(np.ascontiguousarray(xo), ranges, o, hb, n, xo.shape[0])
        z['from_' + nm] = o
    np.savez_compressed(os.path.join(out, 'constraint.npz'), **z)


def gen_polymodel(bf, out):
    """Masked multi-output PolyModel: fit + fun/jac inside and outside the bound (poly.py:440-589)."""
    from bayesfast.modules import PolyModel, PolyConfig
    rng = np.random.default_rng(13)
    d, m = 6, 3
    configs = [PolyConfig('linear'), PolyConfig('quadratic', input_mask=[0, 2, 3, 5], output_mask=[0, 2]),
               PolyConfig('cubic-2', input_mask=[1, 2, 4], output_mask=[1]),
               PolyConfig('cubic-3', input_mask=[0, 1, 2, 3, 4], output_mask=[0, 1, 2])]
    pm = PolyModel(configs, input_size=d, output_size=m, bound_options=dict(alpha_p=90.))
    npts = 120
    x = rng.normal(size=(npts, d))
    W = rng.normal(size=(d, m))
    y = np.tanh(x @ W) + 0.1 * (x**2) @ np.abs(W) + 0.05 * rng.normal(size=(npts, m))
    logp = -0.5 * np.sum(x**2, 1)
    pm.fit(x, y, logp)
    xe = np.concatenate([rng.normal(size=(6, d)) * 0.5, rng.normal(size=(6, d)) * 4.0])
    f = np.array([pm._fun(xx) for xx in xe])
    j = np.array([pm._jac(xx) for xx in xe])
    beta = np.array([np.dot(np.dot(xx - pm._mu, pm._hess), xx - pm._mu)**0.5 for xx in xe])
    assert (beta > pm._alpha).any() and (beta < pm._alpha).any()
    z = dict(x_fit=x, y_fit=y, logp_fit=logp, x_eval=xe, f=f, j=j, beta=beta, n_param=int(pm.n_param),
             alpha_p=90.)
    z.update(flatten_poly(poly_spec_from_reference(pm)))
    # weighted fit of a single-output quadratic model
    pm2 = PolyModel('quadratic', input_size=4, output_size=1)
    x2 = rng.normal(size=(40, 4))
    y2 = np.sum(x2**2, 1, keepdims=True) + x2[:, :1] * x2[:, 1:2] + 0.01 * rng.normal(size=(40, 1))
    w2 = rng.uniform(0.5, 1.5, size=40)
    pm2.fit(x2, y2, logp=y2[:, 0], w=w2)
    z.update(flatten_poly(poly_spec_from_reference(pm2), 'w.poly.'))
    z.update({'w.x_fit': x2, 'w.y_fit': y2, 'w.w': w2})
    np.savez_compressed(os.path.join(out, 'polymodel.npz'), **z)


def gen_density(bf, out):
    """Density.logp_and_grad with every feature combination (density.py:724-754)."""
    z = {}
    cases = dict(plain=dict(), decay=dict(decay=True), scales=dict(scales=True), su=dict(su_scales=True),
                 full=dict(scales=True, decay=True, su_scales=True))
    for name, kw in cases.items():
        den, rng = make_density(bf, 5, 21, **kw)
        spec = density_spec_from_reference(den)
        # points near and far (far ones leave the surrogate bound and the decay radius)
        xt = np.concatenate([rng.normal(size=(5, 5)) * 0.3, rng.normal(size=(5, 5)) * (1.5 if kw.get('scales') else 4.)])
        lt, gt = zip(*[den.logp_and_grad(x, original_space=False, use_surrogate=True) for x in xt])
        xo = den.to_original(xt)
        lo, go = zip(*[den.logp_and_grad(x, original_space=True, use_surrogate=True) for x in xo])
        z.update(flatten_spec(spec, name + '.'))
        z.update({name + '.x_trans': xt, name + '.logp_trans': np.array(lt), name + '.grad_trans': np.array(gt),
                  name + '.x_orig': xo, name + '.logp_orig': np.array(lo), name + '.grad_orig': np.array(go)})
    # headline-shaped case: d = 64 quadratic logp surrogate, bound on, no scales
    den, rng = make_density(bf, 64, 22, n_fit_mult=2)
    spec = density_spec_from_reference(den)
    xt = np.concatenate([rng.normal(size=(6, 64)), rng.normal(size=(2, 64)) * 3.])
    lt, gt = zip(*[den.logp_and_grad(x, original_space=False, use_surrogate=True) for x in xt])
    z.update(flatten_spec(spec, 'd64.'))
    z.update({'d64.x_trans': xt, 'd64.logp_trans': np.array(lt), 'd64.grad_trans': np.array(gt)})
    np.savez_compressed(os.path.join(out, 'density.npz'), **z)


def gen_sampler(bf, out):
    """Leapfrog states and logged-RNG NUTS/HMC trajectories (integration.py, nuts.py, hmc.py, base_hmc.py)."""
    from bayesfast.samplers import NUTS, HMC, NTrace, HTrace
    from bayesfast.samplers.hmc_utils.integration import CpuLeapfrogIntegrator
    from bayesfast.samplers.hmc_utils.metrics import QuadMetricDiag
    z = {}
    for name, d, kw, n_iter, n_warmup, n_chain in (('full5', 5, dict(scales=True, decay=True, su_scales=True), 40, 25, 3),
                                                   ('plain16', 16, dict(), 40, 25, 2),
                                                   ('d64', 64, dict(n_fit_mult=2), 30, 20, 2)):
        den, rng = make_density(bf, d, 31 + d, **kw)
        spec = density_spec_from_reference(den)
        z.update(flatten_spec(spec, name + '.'))

        def lg(x):
            return den.logp_and_grad(x, original_space=False, use_surrogate=True)

        # (iv) single leapfrog steps
        var = rng.uniform(0.5, 2., size=d)
        integ = CpuLeapfrogIntegrator(QuadMetricDiag(var), lg)
        q0 = rng.normal(size=d) * 0.5
        p0 = rng.normal(size=d)
        s0 = integ.compute_state(q0, p0)
        s1 = integ.step(0.13, s0)
        s2 = integ.step(-0.07, s1)
        for tag, s in (('s0', s0), ('s1', s1), ('s2', s2)):
            for fld in ('q', 'p', 'velocity', 'q_grad', 'energy', 'logp'):
                z['%s.lf.%s.%s' % (name, tag, fld)] = np.asarray(getattr(s, fld))
        z[name + '.lf.var'] = var
        z[name + '.lf.eps'] = np.array([0.13, -0.07])

        # (v) NUTS trajectories with logged draws
        x0 = rng.normal(size=(n_chain, d)) * 0.5
        z[name + '.x0'] = x0
        z[name + '.n_iter'], z[name + '.n_warmup'] = np.asarray(n_iter), np.asarray(n_warmup)
        for c in range(n_chain):
            t = NTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, x_0=x0.copy(), random_generator=1234)
            t._init_chain(c)
            log = LoggingGenerator(t._random_generator)
            t._random_generator = log
            NUTS(logp_and_grad=lg, sample_trace=t).run(verbose=False)
            k = '%s.nuts%d.' % (name, c)
            z[k + 'samples'] = t.samples
            z[k + 'normals'] = np.array(log.normals)
            z[k + 'uniforms'] = np.array(log.uniforms)
            for si in t.stats.stats_items:
                z[k + si] = np.array(getattr(t.stats, '_' + si), dtype=float)
            z[k + 'final_var'] = np.array(t.metric._var)
            z[k + 'n_call'] = np.asarray(t.n_call)
        # HMC
        for c in range(2):
            t = HTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, n_int_step=8, x_0=x0.copy(),
                       random_generator=4321)
            t._init_chain(c)
            log = LoggingGenerator(t._random_generator)
            t._random_generator = log
            HMC(logp_and_grad=lg, sample_trace=t).run(verbose=False)
            k = '%s.hmc%d.' % (name, c)
            z[k + 'samples'] = t.samples
            z[k + 'normals'] = np.array(log.normals)
            z[k + 'uniforms'] = np.array(log.uniforms)
            for si in t.stats.stats_items:
                z[k + si] = np.array(getattr(t.stats, '_' + si), dtype=float)
    # a deliberately hard case: huge step size => divergences and early U-turns
    den, rng = make_density(bf, 5, 77)
    spec = density_spec_from_reference(den)
    z.update(flatten_spec(spec, 'div5.'))
    x0 = rng.normal(size=(1, 5))
    t = NTrace(n_chain=1, n_iter=30, n_warmup=10, x_0=x0.copy(), random_generator=99, step_size=40.,
               max_change=50.)
    t._init_chain(0)
    log = LoggingGenerator(t._random_generator)
    t._random_generator = log
    NUTS(logp_and_grad=lambda x: den.logp_and_grad(x, original_space=False, use_surrogate=True),
         sample_trace=t).run(verbose=False)
    z['div5.x0'] = x0
    z['div5.nuts0.samples'] = t.samples
    z['div5.nuts0.normals'] = np.array(log.normals)
    z['div5.nuts0.uniforms'] = np.array(log.uniforms)
    for si in t.stats.stats_items:
        z['div5.nuts0.' + si] = np.array(getattr(t.stats, '_' + si), dtype=float)
    np.savez_compressed(os.path.join(out, 'sampler.npz'), **z)


def gen_sampler_fullmetric(bf, out):
    """Full-rank metric (QuadMetricFull / QuadMetricFullAdapt, samplers/hmc_utils/metrics.py:94-132,240-330,
    374-417): leapfrog states and logged-RNG NUTS / HMC trajectories."""
    from bayesfast.samplers import NUTS, HMC, NTrace, HTrace
    from bayesfast.samplers.hmc_utils.integration import CpuLeapfrogIntegrator
    from bayesfast.samplers.hmc_utils.metrics import QuadMetricFull
    z = {}
    d = 8
    den, rng = make_density(bf, d, 57)
    spec = density_spec_from_reference(den)
    z.update(flatten_spec(spec, 'fm8.'))

    def lg(x):
        return den.logp_and_grad(x, original_space=False, use_surrogate=True)

    B = rng.normal(size=(d, d)) * 0.3
    cov0 = np.eye(d) * 0.8 + B @ B.T
    z['fm8.cov0'] = cov0
    integ = CpuLeapfrogIntegrator(QuadMetricFull(cov0), lg)
    q0 = rng.normal(size=d) * 0.5
    p0 = rng.normal(size=d)
    s0 = integ.compute_state(q0, p0)
    s1 = integ.step(0.11, s0)
    s2 = integ.step(-0.06, s1)
    for tag, st in (('s0', s0), ('s1', s1), ('s2', s2)):
        for fld in ('q', 'p', 'velocity', 'q_grad', 'energy', 'logp'):
            z['fm8.lf.%s.%s' % (tag, fld)] = np.asarray(getattr(st, fld))
    z['fm8.lf.eps'] = np.array([0.11, -0.06])
    n_iter, n_warmup, n_chain = 40, 28, 2
    x0 = rng.normal(size=(n_chain, d)) * 0.5
    z['fm8.x0'] = x0
    z['fm8.n_iter'], z['fm8.n_warmup'] = np.asarray(n_iter), np.asarray(n_warmup)

    def record(t, sampler, key):
        log = LoggingGenerator(t._random_generator)
        t._random_generator = log
        sampler(logp_and_grad=lg, sample_trace=t).run(verbose=False)
        z[key + 'samples'] = t.samples
        z[key + 'normals'] = np.array(log.normals)
        z[key + 'uniforms'] = np.array(log.uniforms)
        for si in t.stats.stats_items:
            z[key + si] = np.array(getattr(t.stats, '_' + si), dtype=float)
        z[key + 'final_cov'] = np.array(t.metric._cov)

    # adaptive full metric from the identity (metric='full'), NUTS: small adapt_window so that the window switch
    # and the doubling both happen inside the warm-up
    for c in range(n_chain):
        t = NTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, x_0=x0.copy(), random_generator=777,
                   metric='full', adapt_window=8)
        t._init_chain(c)
        record(t, NUTS, 'fm8.nuts_adapt%d.' % c)
    # fixed full metric (adapt_metric=False): NUTS and HMC
    t = NTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, x_0=x0.copy(), random_generator=778,
               metric=cov0.tolist(), adapt_metric=False)
    t._init_chain(0)
    record(t, NUTS, 'fm8.nuts_fixed0.')
    t = HTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, n_int_step=8, x_0=x0.copy(),
               random_generator=779, metric=cov0.tolist(), adapt_metric=False)
    t._init_chain(1)
    record(t, HMC, 'fm8.hmc_fixed1.')
    # adaptive from a given covariance, HMC
    t = HTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, n_int_step=8, x_0=x0.copy(),
               random_generator=780, metric=cov0.tolist(), adapt_window=8)
    t._init_chain(0)
    record(t, HMC, 'fm8.hmc_adapt0.')
    np.savez_compressed(os.path.join(out, 'sampler_fullmetric.npz'), **z)


def gen_refit(bf, out):
    """Refit glue: SystematicResampler.run (utils/misc.py:61-108) on arrays with and without ties, for
    several node / weight settings, and the truncated importance weights of PostStep (core/recipe.py:1289-1296)."""
    from bayesfast.utils.misc import SystematicResampler
    rng = np.random.default_rng(77)
    z = {}
    cases = [dict(m=5000, n=400, nodes=(1., 100.), weights=None),
             dict(m=20011, n=2145, nodes=(1., 100.), weights=None),
             dict(m=4096, n=333, nodes=(0., 50., 90., 100.), weights=(1., 2., 3.)),
             dict(m=1000, n=1000, nodes=(0., 100.), weights=None),
             dict(m=3000, n=7, nodes=(5., 25., 100.), weights=(0.2, 0.8))]
    z['n_case'] = np.asarray(len(cases))
    for i, c in enumerate(cases):
        a = rng.normal(size=c['m']) * 3. - 40.
        r = SystematicResampler(nodes=c['nodes'], weights=c['weights'])
        idx = r.run(a, c['n'])
        z['c%d.a' % i] = a
        z['c%d.n' % i] = np.asarray(c['n'])
        z['c%d.nodes' % i] = np.asarray(c['nodes'], dtype=np.float64)
        z['c%d.weights' % i] = np.asarray(c['weights'] if c['weights'] is not None else [], dtype=np.float64)
        z['c%d.idx' % i] = np.asarray(idx, dtype=np.int64)
    # ties (repeated samples of a chain that stayed put): the selected VALUES are what is pinned
    a = np.repeat(rng.normal(size=700), 3)[rng.permutation(2100)]
    r = SystematicResampler(require_unique=False)
    idx = r.run(a, 150)
    z['ties.a'] = a
    z['ties.n'] = np.asarray(150)
    z['ties.values'] = a[idx]
    # importance weights with truncation, recipe.py:1289-1296
    logp = rng.normal(size=3000) * 2.
    logq = logp + rng.normal(size=3000) * 0.7
    for k_trunc in (0.25, -1.):
        w = np.exp(logp - logq)
        wt = w.copy() if k_trunc < 0 else np.clip(w, 0, np.mean(w) * w.size**k_trunc)
        z['iw.k%g.w' % k_trunc] = w
        z['iw.k%g.wt' % k_trunc] = wt
    z['iw.logp'] = logp
    z['iw.logq'] = logq
    np.savez_compressed(os.path.join(out, 'refit.npz'), **z)


def gen_evidence(bf, out):
    """Evidence path: kde.cdf, the cubic_spline arrays and kernels, a fitted SIT (rotations, splines,
    logq, forward / backward transforms) on funnel-shaped samples, and bridge() inputs / outputs."""
    if not hasattr(np, 'asscalar'):  # removed NumPy alias still used by utils/kde.py:347
        np.asscalar = lambda a: np.asarray(a).item()
    from bayesfast.utils.kde import kde
    from bayesfast.utils.cubic import cubic_spline
    from bayesfast.transforms import SIT
    from bayesfast.evidence.bridge import bridge
    from scipy.stats import norm
    rng = np.random.default_rng(2024)
    z = {}
    # -- 1-d kde + Gaussianizing spline of a skewed sample
    x1 = np.concatenate((rng.normal(size=3000), rng.normal(size=1500) * 0.4 + 2.5, rng.standard_t(3, size=500)))
    w1 = np.full(x1.size, 1. / x1.size)
    k = kde(x1, bw_factor=1., weights=w1)
    pts = np.linspace(-6, 6, 41)
    z['kde.x'], z['kde.w'], z['kde.pts'] = x1, w1, pts
    z['kde.h'] = np.asarray(float(np.asarray(k.covariance).item())**0.5)
    z['kde.cdf'] = k.cdf(pts)
    c = cubic_spline(x1, lambda xx: norm.ppf(k.cdf(xx)))
    z['spl.x'], z['spl.y'], z['spl.c'] = c._x, c._y, c._c
    tp = np.concatenate((np.linspace(x1.min() - 1, x1.max() + 1, 97), c._x[:5], [np.nan]))
    z['spl.pts'] = tp
    z['spl.evaluate'] = c.evaluate(tp)
    z['spl.derivative'] = c.derivative(tp)
    yp = np.concatenate((np.linspace(c._y[0] - 1, c._y[-1] + 1, 83), c._y[:4]))
    z['spl.ypts'] = yp
    z['spl.solve'] = c.solve(yp)
    # -- SIT on a 6-d funnel (examples/funnel-gbs.ipynb shape, a = 1, b = 0.5), 3 iterations
    D, n = 6, 4000
    x0 = rng.normal(size=n)
    data = np.concatenate((x0[:, None], rng.normal(size=(n, D - 1)) * np.exp(0.5 * x0)[:, None]), 1)
    sit = SIT(n_iter=3, parallel_backend=1, random_generator=11, mvn_generator=lambda m, c, s: rng.normal(size=(s, len(m))))
    sit.fit(data)
    z['sit.data'] = data
    z['sit.A'], z['sit.B'], z['sit.m'], z['sit.logdetA'] = sit._A, sit._B, sit._m, sit._logdetA
    for i in range(3):
        for j in range(D):
            cs = sit._cubic[i][j]
            z['sit.it%d.d%d.x' % (i, j)], z['sit.it%d.d%d.y' % (i, j)], z['sit.it%d.d%d.c' % (i, j)] = cs._x, cs._y, cs._c
    z['sit.data_final'] = sit._data
    xt = data[:200] * 1.1
    z['sit.xt'] = xt
    z['sit.logq'] = sit.logq(xt)
    yf, ljf = sit.forward_transform(xt)
    z['sit.forward_y'], z['sit.forward_logj'] = yf, ljf
    yb = rng.normal(size=(150, D))
    xb, ljb = sit.backward_transform(yb)
    z['sit.yb'], z['sit.backward_x'], z['sit.backward_logj'] = yb, xb, ljb
    # -- bridge: p samples as (chains, iterations) with autocorrelation, q samples flat
    n_c, n_i, n_q = 4, 600, 2000
    e = rng.normal(size=(n_c, n_i))
    for t in range(1, n_i):
        e[:, t] = 0.6 * e[:, t - 1] + 0.8 * e[:, t]
    lpp = -0.5 * e**2 - 3.
    lqp = -0.5 * (e / 1.2)**2 - np.log(1.2) + 0.05 * np.sin(e)
    q = rng.normal(size=n_q) * 1.2
    lpq = -0.5 * q**2 - 3.
    lqq = -0.5 * (q / 1.2)**2 - np.log(1.2) + 0.05 * np.sin(q)
    logr, err = bridge(lpp, lpq, lqp, lqq)
    z['br.lpp'], z['br.lpq'], z['br.lqp'], z['br.lqq'] = lpp, lpq, lqp, lqq
    z['br.logr'], z['br.err'] = np.asarray(logr), np.asarray(err)
    # -- integrated autocorrelation time (utils/acor.py:79-145) of correlated series: (walker, time, dimension), (time, dimension), (time,)
    from bayesfast.utils.acor import integrated_time
    rng2 = np.random.default_rng(31)
    e = rng2.normal(size=(3, 3000, 2))
    for t in range(1, e.shape[1]):
        e[:, t] = np.array([0.5, 0.8]) * e[:, t - 1] + e[:, t]
    z['acor.x'] = e
    z['acor.tau3'] = integrated_time(e)
    z['acor.tau2'] = integrated_time(e[0])
    z['acor.tau1'] = integrated_time(e[1, :, 0])
    np.savez_compressed(os.path.join(out, 'evidence.npz'), **z)


def gen_pipeline(bf, out):
    """a reference Density whose first module is replaced by a multi-output PolyModel surrogate with linear,
    quadratic and cubic configs, followed by an analytic chi-square module: fitted coefficients, and
    Density.logp_and_grad(use_surrogate=True) (core/density.py:487-566,724-754) at points inside and outside the bound."""
    rng = np.random.default_rng(31)
    d, m = 7, 5
    W1 = rng.normal(size=(m, d)) * 0.4
    ydat = rng.normal(size=m)
    cov = np.diag(rng.uniform(0.5, 2., m)) + 0.1
    prec = np.linalg.inv(cov)

    def f_model(x):
        return W1 @ x + 0.3 * np.tanh(x[:m]) * x[1:m + 1] + 0.05 * x[:m]**3

    def f_chi2(y):
        r = y - ydat
        return -0.5 * r @ prec @ r

    def j_chi2(y):
        return -(prec @ (y - ydat))[np.newaxis]

    mod0 = bf.Module(fun=f_model, input_vars='x', output_vars='y')
    mod1 = bf.Module(fun=f_chi2, jac=j_chi2, input_vars='y', output_vars='logp')
    cfgs = [bf.modules.PolyConfig('linear'), bf.modules.PolyConfig('quadratic'),
            bf.modules.PolyConfig('cubic-2', input_mask=[0, 2, 3, 5], output_mask=[0, 1, 3]),
            bf.modules.PolyConfig('cubic-3', input_mask=[1, 2, 4, 5, 6], output_mask=[1, 2, 4])]
    su = bf.modules.PolyModel(cfgs, input_size=d, output_size=m, input_vars='x', output_vars='y')
    den = bf.Density(module_list=[mod0, mod1], surrogate_list=[su], input_shapes=[d], input_vars='x', density_name='logp')
    xf = rng.normal(size=(3 * su.n_param, d))
    yf = np.array([f_model(x) for x in xf])
    lpf = np.array([f_chi2(y) for y in yf])
    su.fit(xf, yf, lpf)
    z = flatten_poly(poly_spec_from_reference(su), 'poly.')
    z['x_fit'], z['y_fit'], z['logp_fit'] = xf, yf, lpf
    z['ydat'], z['prec'] = ydat, prec
    xt = np.concatenate((rng.normal(size=(25, d)) * 0.7, rng.normal(size=(10, d)) * 3.))
    lg = [den.logp_and_grad(x, use_surrogate=True) for x in xt]
    z['xt'] = xt
    z['logp'] = np.array([v[0] for v in lg])
    z['grad'] = np.array([v[1] for v in lg])
    z['su_f'] = np.array([su.fun(x)[0] for x in xt]).reshape(len(xt), m)
    np.savez_compressed(os.path.join(out, 'pipeline.npz'), **z)


def gen_pipeline_des(bf, out):
    """SURVEY 8f-1 in the shape of examples/des-y1-w-cosmosis.ipynb (cells 12-18), scaled down: a THREE-module pipeline
    x -> m (multi-output model, replaced by a PolyModel surrogate = linear on all inputs + quadratic on a masked subset, with
    surrogate input_scales) -> like = -|m - d|^2 / 2 + norm (whitened chi-square) -> logp = like + Gaussian prior on some of
    the x, in a Density with input_scales and hard bounds and the decay term.  Recorded: the fitted surrogate, the density's
    state, Density.logp_and_grad(use_surrogate=True) in both spaces inside and outside the bound (core/density.py:487-566,
    724-754; modules/poly.py:466-503), and NUTS trajectories of the reference's own sampler on it with logged draws."""
    from bayesfast.samplers import NUTS, NTrace
    rng = np.random.default_rng(457)
    d, m = 9, 22
    lo = -1. - rng.uniform(size=d)
    hi = 1.5 + rng.uniform(size=d)
    para_range = np.stack([lo, hi], 1)
    nonlinear = np.array([0, 1, 2, 5])
    prior_idx = np.array([3, 4, 6, 7, 8])
    prior_mu = rng.normal(size=prior_idx.size) * 0.1
    prior_sig = rng.uniform(0.2, 0.5, size=prior_idx.size)
    prior_norm = -0.5 * np.sum(np.log(2 * np.pi * prior_sig**2))
    W1 = rng.normal(size=(m, d)) * 0.6
    W2 = rng.normal(size=(m, nonlinear.size, nonlinear.size)) * 0.25
    dvec = rng.normal(size=m) * 0.3
    norm_c = -3.21

    def model(x):
        z = x[nonlinear]
        return W1 @ x + np.einsum('ojk,j,k->o', W2, z, z) + 0.05 * np.sin(2. * z[0]) * np.ones(m)

    def chi2_f(mm):
        return np.atleast_1d(-0.5 * np.sum((mm - dvec)**2) + norm_c)

    def chi2_fj(mm):
        return np.atleast_1d(-0.5 * np.sum((mm - dvec)**2) + norm_c), -(mm - dvec)[np.newaxis]

    def prior_f(x):
        return -0.5 * np.sum(((x[prior_idx] - prior_mu) / prior_sig)**2) + prior_norm

    def post_f(like, x):
        return like + prior_f(x)

    def post_fj(like, x):
        pj = np.zeros((1, d))
        pj[0, prior_idx] = -(x[prior_idx] - prior_mu) / prior_sig**2
        return like + prior_f(x), np.concatenate((np.ones((1, 1)), pj), axis=-1)

    mod0 = bf.Module(fun=model, input_vars='x', output_vars='m')
    mod1 = bf.Module(fun=chi2_f, fun_and_jac=chi2_fj, input_vars='m', output_vars='like')
    mod2 = bf.Module(fun=post_f, fun_and_jac=post_fj, input_vars=['like', 'x'], output_vars='logp')
    su = bf.modules.PolyModel([bf.modules.PolyConfig('linear'), bf.modules.PolyConfig('quadratic', input_mask=nonlinear)],
                              input_size=d, output_size=m, input_vars='x', output_vars='m', input_scales=para_range)
    z = {}
    for tag, decay in (('a', False), ('b', True)):
        den = bf.Density(density_name='logp', module_list=[mod0, mod1, mod2], surrogate_list=[su], input_vars='x', input_shapes=d,
                         input_scales=para_range, hard_bounds=True, decay_options=dict(use_decay=decay))
        xf = lo + (hi - lo) * (0.5 + 0.22 * rng.normal(size=(3 * int(su.n_param), d))).clip(0.02, 0.98)
        vds = [den.fun(x, original_space=True, use_surrogate=False) for x in xf]
        den.fit(vds)
        z.update(flatten_poly(poly_spec_from_reference(su), tag + '.poly.'))
        z[tag + '.ranges'], z[tag + '.hard_bounds'] = para_range, np.ones((d, 2), np.uint8)
        z[tag + '.su_lo'], z[tag + '.su_diff'] = np.array(su._input_scales[:, 0]), np.array(su._input_scales_diff)
        z[tag + '.use_decay'] = np.asarray(decay)
        if decay:
            z[tag + '.decay_mu'], z[tag + '.decay_hess'] = np.array(den._mu), np.array(den._hess)
            z[tag + '.decay_alpha2'], z[tag + '.decay_gamma'] = np.asarray(den._alpha_2), np.asarray(den._gamma)
        z[tag + '.x_fit'] = xf
        z[tag + '.y_fit'] = np.array([vd._fun['m'] for vd in vds])
        z[tag + '.logp_fit'] = np.array([vd._fun['logp'][0] for vd in vds])
        # evaluation points: original space inside the range (near and far from the fit cloud), and their transformed images
        xo = lo + (hi - lo) * np.concatenate(((0.5 + 0.15 * rng.normal(size=(20, d))).clip(0.03, 0.97),
                                              rng.uniform(0.02, 0.98, size=(14, d))))
        xt = np.array([den.from_original(x) for x in xo])
        z[tag + '.xo'], z[tag + '.xt'] = xo, xt
        for sp, pts, key in ((True, xo, 'orig'), (False, xt, 'trans')):
            lg = [den.logp_and_grad(x, original_space=sp, use_surrogate=True) for x in pts]
            z['%s.logp_%s' % (tag, key)] = np.array([v[0] for v in lg])
            z['%s.grad_%s' % (tag, key)] = np.array([v[1] for v in lg])
        z[tag + '.su_f'] = np.array([su.fun((x - su._input_scales[:, 0]) / su._input_scales_diff)[0] for x in xo]).reshape(len(xo), m)
        beta = np.array([np.dot(np.dot(xs - su._mu, su._hess), xs - su._mu)**0.5
                         for xs in (xo - su._input_scales[:, 0]) / su._input_scales_diff])
        z[tag + '.n_outside_bound'] = np.asarray(int(np.sum(beta > su._alpha)))

        def lgt(x):
            return den.logp_and_grad(x, original_space=False, use_surrogate=True)

        n_chain, n_iter, n_warmup = 2, 36, 24
        x0 = xt[:n_chain].copy()
        z[tag + '.x0'] = x0
        z[tag + '.n_iter'], z[tag + '.n_warmup'] = np.asarray(n_iter), np.asarray(n_warmup)
        for c in range(n_chain):
            t = NTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, x_0=x0.copy(), random_generator=2718)
            t._init_chain(c)
            log = LoggingGenerator(t._random_generator)
            t._random_generator = log
            NUTS(logp_and_grad=lgt, sample_trace=t).run(verbose=False)
            k = '%s.nuts%d.' % (tag, c)
            z[k + 'samples'] = t.samples
            z[k + 'normals'] = np.array(log.normals)
            z[k + 'uniforms'] = np.array(log.uniforms)
            for si in t.stats.stats_items:
                z[k + si] = np.array(getattr(t.stats, '_' + si), dtype=float)
    z['chi2.y'], z['chi2.logp0'] = dvec, np.asarray(norm_c)
    z['prior.idx'], z['prior.mu'], z['prior.sig'], z['prior.c0'] = prior_idx, prior_mu, prior_sig, np.asarray(prior_norm)
    np.savez_compressed(os.path.join(out, 'pipeline_des.npz'), **z)


def gen_tempered(bf, out):
    """Tempered samplers: TCpuLeapfrogIntegrator states (integration.py:98-222) and TNUTS trajectories with
    logged draws (tnuts.py, base_hmc.py:220-262) on a surrogate target with a Gaussian base density.  (THMC is not
    recorded: the reference's THTrace constructor raises, samplers/sample_trace.py:600.)"""
    from bayesfast.samplers.sample_trace import TNTrace
    from bayesfast.samplers.tnuts import TNUTS
    from bayesfast.samplers.hmc_utils.integration import TCpuLeapfrogIntegrator
    from bayesfast.samplers.hmc_utils.metrics import QuadMetricDiag
    z = {}
    d = 6
    den, rng = make_density(bf, d, 57)
    spec = density_spec_from_reference(den)
    z.update(flatten_spec(spec, 't6.'))

    def lg(x):
        return den.logp_and_grad(x, original_space=False, use_surrogate=True)

    bmean = rng.normal(size=d) * 0.2
    L = np.eye(d) * 1.3 + 0.2 * np.tril(rng.normal(size=(d, d)), -1)
    bcov = L @ L.T
    bprec = np.linalg.inv(bcov)
    blogdet = np.linalg.slogdet(bcov)[1]

    def blogp(x):
        r = x - bmean
        return -0.5 * r @ bprec @ r - 0.5 * (d * np.log(2 * np.pi) + blogdet)

    base = bf.DensityLite(logp=blogp, grad=lambda x: -bprec @ (x - bmean), input_size=d)
    logxi = 0.37
    z['t6.base_mean'], z['t6.base_cov'], z['t6.logxi'] = bmean, bcov, np.asarray(logxi)
    # integrator states
    var = rng.uniform(0.5, 2., size=d)

    def lgb(x):
        lp, g = base.logp_and_grad(x, original_space=False)
        return lp + logxi, g

    integ = TCpuLeapfrogIntegrator(QuadMetricDiag(var), lg, lgb)
    q0, p0 = rng.normal(size=d) * 0.5, rng.normal(size=d)
    u0, v0 = 0.4, -0.7
    s0 = integ.compute_state(np.append(u0, q0), np.append(v0, p0))
    s1 = integ.step(0.11, s0)
    s2 = integ.step(-0.06, s1)
    for tag, st in (('s0', s0), ('s1', s1), ('s2', s2)):
        for fld in ('q', 'u', 'p', 'v', 'weight', 'energy', 'logp'):
            z['t6.lf.%s.%s' % (tag, fld)] = np.asarray(getattr(st, fld), dtype=float)
    z['t6.lf.var'], z['t6.lf.eps'] = var, np.array([0.11, -0.06])
    z['t6.lf.q0'], z['t6.lf.p0'], z['t6.lf.u0v0'] = q0, p0, np.array([u0, v0])
    # TNUTS trajectories
    n_chain, n_iter, n_warmup = 2, 40, 25
    x0 = rng.normal(size=(n_chain, d)) * 0.5
    z['t6.x0'] = x0
    z['t6.n_iter'], z['t6.n_warmup'] = np.asarray(n_iter), np.asarray(n_warmup)
    for c in range(n_chain):
        np.random.seed(100 + c)  # u_0 of the first iteration comes from numpy's global generator (base_hmc.py:241)
        u_first = np.random.normal(0, 1)
        np.random.seed(100 + c)
        t = TNTrace(density_base=base, logxi=logxi, n_chain=n_chain, n_iter=n_iter, n_warmup=n_warmup, x_0=x0.copy(),
                    random_generator=2468)
        t._init_chain(c)
        log = LoggingGenerator(t._random_generator)
        t._random_generator = log
        TNUTS(logp_and_grad=lg, sample_trace=t).run(verbose=False)
        k = 't6.tnuts%d.' % c
        z[k + 'u_first'] = np.asarray(u_first)
        z[k + 'samples'] = t.samples
        z[k + 'normals'] = np.array(log.normals)
        z[k + 'uniforms'] = np.array(log.uniforms)
        for si in t.stats.stats_items:
            z[k + si] = np.array(getattr(t.stats, '_' + si), dtype=float)
    np.savez_compressed(os.path.join(out, 'tempered.npz'), **z)


def gen_fit_illcond(bf, out):
    """PolyModel.fit (scipy.linalg.lstsq = LAPACK gelsd, modules/poly.py:566-587) on ill-conditioned and rank-deficient
    designs: a cubic model whose inputs sit far from the origin (the monomials are nearly collinear) at two offsets, and a
    quadratic model with a duplicated input column.  Recorded: the data, the reference's coefficients and predictions,
    and the condition number of the column-equilibrated design matrix."""
    from bayesfast.modules import PolyModel, PolyConfig
    from bayesfast.modules import _poly
    rng = np.random.default_rng(2145)
    z = {}

    def design(x, orders):
        n, d = x.shape
        blocks = [np.ones((n, 1)), x]
        if 'quadratic' in orders:
            a = np.empty((n, d * (d + 1) // 2)); _poly._lsq_quadratic(x, a, n, d); blocks.append(a)
        if 'cubic-2' in orders:
            a = np.empty((n, d * d)); _poly._lsq_cubic_2(x, a, n, d); blocks.append(a)
        if 'cubic-3' in orders:
            a = np.empty((n, d * (d - 1) * (d - 2) // 6)); _poly._lsq_cubic_3(x, a, n, d); blocks.append(a)
        return np.concatenate(blocks, 1)

    for tag, off, sc in (('ill5', 2.5, 0.3), ('ill7', 4.0, 0.11)):
        d = 5
        orders = ['linear', 'quadratic', 'cubic-2', 'cubic-3']
        pm = PolyModel(orders, input_size=d, output_size=1, bound_options=dict(use_bound=False))
        n = pm.n_param + 25
        x = off + sc * rng.normal(size=(n, d))
        y = (np.sin(x.sum(1)) + 0.3 * x[:, 0] * x[:, 1] * x[:, 2] - 0.1 * x[:, 3]**2 * x[:, 4] + 0.5 * x[:, 1]**2)[:, None]
        pm.fit(x, y)
        A = design(x, orders)
        sv = np.linalg.svd(A / np.linalg.norm(A, axis=0), compute_uv=False)
        xe = off + sc * rng.normal(size=(30, d))
        z[tag + '.x'], z[tag + '.y'], z[tag + '.xe'] = x, y, xe
        z[tag + '.f_fit'] = np.array([pm._fun(xx) for xx in x])
        z[tag + '.f_eval'] = np.array([pm._fun(xx) for xx in xe])
        z[tag + '.cond_equilibrated'] = np.asarray(sv[0] / sv[-1])
        z.update(flatten_poly(poly_spec_from_reference(pm), tag + '.poly.'))
    # rank deficient: input 3 duplicates input 0
    d = 4
    pm = PolyModel('quadratic', input_size=d, output_size=1, bound_options=dict(use_bound=False))
    n = 3 * pm.n_param
    x = rng.normal(size=(n, d))
    x[:, 3] = x[:, 0]
    y = (x[:, 0]**2 - x[:, 1] * x[:, 2] + 0.7 * x[:, 2] + 0.01 * rng.normal(size=n))[:, None]
    pm.fit(x, y)
    z['rankdef.x'], z['rankdef.y'] = x, y
    z['rankdef.f_fit'] = np.array([pm._fun(xx) for xx in x])
    z.update(flatten_poly(poly_spec_from_reference(pm), 'rankdef.poly.'))
    np.savez_compressed(os.path.join(out, 'fit_illcond.npz'), **z)


def gen_recipe(bf, out):
    """Fixture (vi) of SURVEY section 7 step 0: BASELINE config 1, the 2-d donut recipe of examples/2d-donut.ipynb (cell 4,
    n_chain = 4) run END TO END by the reference itself (its Recipe, its NUTS on 4 worker processes, its PolyModel.fit):
    the optimiser's log (what the notebook prints at :110-141), and per SampleStep the statistics of the ring and the
    surrogate it sampled.  The recipe's definition lives in tests/helpers/donut.py (written against the bayesfast API)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'helpers'))
    import donut
    rec = donut.build_recipe(bf)
    rec.run()
    rt = rec.recipe_trace
    z = {}
    opt = rt.results.optimize
    z['opt.logp'] = np.array([r.f_max.logp for r in opt[:-1]])
    z['opt.logp_trans'] = np.array([r.f_max.logp_trans for r in opt[:-1]])
    z['opt.logq_trans'] = np.array([r.f_max.logq_trans for r in opt[:-1]])
    z['opt.x_max'] = np.array([r.x_max.x for r in opt[:-1]])
    z['opt.chosen_logp_trans'] = np.array(opt[-1].f_max.logp_trans)
    z['opt.ring'] = donut.ring_statistics(opt[-1].samples)
    z['opt.x_0'] = np.array(rt._s_optimize.x_0)
    steps = rt.results.sample
    z['ring'] = np.array([donut.ring_statistics(r.samples) for r in steps])          # (10, 5): radius mean, sd, ...
    z['tree_size_mean'] = np.array([np.mean([np.mean(t.stats._tree_size) for t in r.sample_trace]) for r in steps])
    z['n_fit'] = np.array([len(r.var_dicts) for r in steps])
    z['step0.x_fit'] = np.array([np.atleast_1d(vd._fun['x']) for vd in steps[0].var_dicts])   # the 30 points SampleStep #0 fitted on
    z['step0.step_size'] = np.array(bf.samplers._get_step_size(opt[-1].sample_trace))
    z['last.samples'] = np.array(steps[-1].sample_trace.get(flatten=False))          # (4, 500, 2)
    z['last.logq'] = np.array(steps[-1].sample_trace.get(return_type='logp', flatten=False))
    su = steps[-1].surrogate_list[0]
    z.update(flatten_poly(poly_spec_from_reference(su), 'last.poly.'))
    z['n_call'] = np.array(rec.get().n_call)
    z['post.ring'] = donut.ring_statistics(rec.get().samples)
    np.savez_compressed(os.path.join(out, 'recipe.npz'), **z)


def gen_sobol(bf, out):
    """Default starting points of sample() (core/sample.py:106-113): Sobol-normal points, utils/sobol.py:49-61 (direction
    numbers new-joe-kuo-6.21201, Gray-code order, the first point skipped)."""
    from bayesfast.utils.sobol import multivariate_normal, uniform
    z = {}
    for d, n in ((2, 8), (5, 33), (64, 16), (128, 4)):
        z['normal_%d_%d' % (d, n)] = multivariate_normal(np.zeros(d), np.eye(d), n)
    z['uniform_3_10'] = uniform(np.zeros(3), np.ones(3), 10)
    rng = np.random.default_rng(3)
    a = rng.normal(size=(4, 4))
    z['cov'] = a @ a.T + np.eye(4)
    z['mean'] = rng.normal(size=4)
    z['normal_cov_4_12'] = multivariate_normal(z['mean'], z['cov'], 12)
    np.savez(os.path.join(out, 'sobol.npz'), **z)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--work', default='/tmp/bfref')
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    bf = prepare_reference(a.ref, a.work)
    gens = dict(poly_kernels=gen_poly_kernels, constraint=gen_constraint, polymodel=gen_polymodel,
                density=gen_density, sampler=gen_sampler, sampler_fullmetric=gen_sampler_fullmetric, refit=gen_refit, evidence=gen_evidence, pipeline=gen_pipeline, pipeline_des=gen_pipeline_des, tempered=gen_tempered,
                fit_illcond=gen_fit_illcond, recipe=gen_recipe, sobol=gen_sobol)
    for k, g in gens.items():
        if a.only and k != a.only:
            continue
        g(bf, HERE)
        print('wrote', k)


if __name__ == '__main__':
    main()
