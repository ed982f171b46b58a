"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  It wraps ``oracle/_build/libbf_oracle.so`` (C restatement of the reference's per-chain
arithmetic, see ``bf_oracle.h``) and adds the few NumPy/SciPy restatements of reference code that is
NumPy/SciPy in the reference too (``PolyModel.fit``, ``_set_bound``, ``_set_decay``).

Density "spec" consumed here (plain dict, NumPy arrays, float64 unless noted)::

    {'d': int,
     'ranges': None | (d,2), 'hard_bounds': None | (d,2) uint8,         # Density.input_scales / hard_bounds
     'su_lo': None | (d,), 'su_diff': None | (d,),                       # Surrogate.input_scales
     'poly': {'input_size': d, 'output_size': m,
              'configs': [{'order': 'linear'|'quadratic'|'cubic-2'|'cubic-3',
                           'input_mask': int array, 'output_mask': int array, 'coef': dense block}],
              'use_bound': bool, 'mu': (d,), 'hess': (d,d), 'alpha': float, 'f_mu': (m,)},
     'use_decay': bool, 'decay_mu': (d,), 'decay_hess': (d,d), 'decay_alpha2': float, 'decay_gamma': float}
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BF_ORACLE_LIB: another build of the same sources (the sanitizer build of `make -C oracle asan`, tests/test_sanitizers.py)
_LIB_PATH = os.environ.get('BF_ORACLE_LIB') or os.path.join(_HERE, '_build', 'libbf_oracle.so')

ORDERS = {'linear': 0, 'quadratic': 1, 'cubic-2': 2, 'cubic-3': 3}

NSTATS = ('logp', 'energy', 'tree_depth', 'tree_size', 'mean_tree_accept', 'step_size', 'step_size_bar',
          'warmup', 'energy_change', 'max_energy_change', 'diverging')
HSTATS = ('logp', 'energy', 'n_int_step', 'accept_stat', 'accepted', 'step_size', 'step_size_bar', 'warmup',
          'energy_change', 'diverging')

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u8p = C.POINTER(C.c_uint8)


class _PolyConfig(C.Structure):
    _fields_ = [('order', C.c_int), ('n_in', C.c_int), ('n_out', C.c_int), ('in_mask', _ip),
                ('out_mask', _ip), ('coef', _dp)]


class _PolyModel(C.Structure):
    _fields_ = [('input_size', C.c_int), ('output_size', C.c_int), ('n_config', C.c_int),
                ('configs', C.POINTER(_PolyConfig)), ('use_bound', C.c_int), ('mu', _dp), ('hess', _dp),
                ('alpha', C.c_double), ('f_mu', _dp)]


class _Density(C.Structure):
    _fields_ = [('d', C.c_int), ('ranges', _dp), ('hard_bounds', _u8p), ('su_lo', _dp), ('su_diff', _dp),
                ('poly', _PolyModel), ('use_decay', C.c_int), ('decay_mu', _dp), ('decay_hess', _dp),
                ('decay_alpha2', C.c_double), ('decay_gamma', C.c_double),
                ('link_kind', C.c_int), ('link_y', C.c_double), ('link_prec', C.c_double), ('link_logp0', C.c_double),
                ('chi2_y', _dp), ('chi2_prec', _dp), ('chi2_prec_diag', _dp), ('prior_mu', _dp), ('prior_prec', _dp),
                ('prior_c0', C.c_double)]


class _Rng(C.Structure):
    _fields_ = [('kind', C.c_int), ('s', C.c_uint64 * 4), ('normals', _dp), ('uniforms', _dp),
                ('n_normals', C.c_size_t), ('n_uniforms', C.c_size_t), ('i_normal', C.c_size_t),
                ('i_uniform', C.c_size_t), ('exhausted', C.c_int)]


class _Chain(C.Structure):
    _fields_ = [('log_step', C.c_double), ('log_bar', C.c_double), ('hbar', C.c_double), ('mu', C.c_double),
                ('target', C.c_double), ('gamma', C.c_double), ('k', C.c_double), ('t_0', C.c_double),
                ('count', C.c_long), ('adapt_step', C.c_int), ('adapt_metric', C.c_int), ('var', _dp),
                ('std', _dp), ('inv_std', _dp), ('fg_mean', _dp), ('fg_raw', _dp), ('bg_mean', _dp),
                ('bg_raw', _dp), ('fg_n', C.c_double), ('bg_n', C.c_double), ('initial_weight', C.c_double),
                ('n_samples', C.c_long), ('previous_update', C.c_long), ('adapt_window', C.c_long),
                ('update_window', C.c_long), ('doubling', C.c_int), ('q', _dp), ('i_iter', C.c_long),
                ('d', C.c_int), ('full', C.c_int), ('cov', _dp), ('chol', _dp), ('fg_cov', _dp), ('bg_cov', _dp),
                ('work', _dp)]


_lib = None


def build(force=False):
    """Compile the C restatement (building the checker is not using it)."""
    if os.environ.get('BF_ORACLE_LIB'):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ('bf_oracle.c', 'bf_oracle.h'))):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.bfo_poly_fun_and_jac.argtypes = [C.POINTER(_PolyModel), _dp, _dp, _dp]
        L.bfo_logp_and_grad.argtypes = [C.POINTER(_Density), _dp, C.c_int, _dp, _dp]
        L.bfo_chain_new.restype = C.POINTER(_Chain)
        L.bfo_chain_new.argtypes = [C.c_int, _dp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                                    C.c_double, _dp, C.c_int, _dp, C.c_double, C.c_long, C.c_long, C.c_int]
        L.bfo_chain_free.argtypes = [C.POINTER(_Chain)]
        L.bfo_leapfrog.argtypes = [C.POINTER(_Density), _dp, C.c_double] + [_dp] * 9
        L.bfo_leapfrog_full.argtypes = L.bfo_leapfrog.argtypes
        L.bfo_chain_set_full.restype = C.c_int
        L.bfo_chain_set_full.argtypes = [C.POINTER(_Chain), _dp]
        L.bfo_nuts_run.argtypes = [C.POINTER(_Density), C.POINTER(_Chain), C.POINTER(_Rng), C.c_long, C.c_long,
                                   C.c_int, C.c_double, _dp, _dp]
        L.bfo_hmc_run.argtypes = L.bfo_nuts_run.argtypes
        L.bfo_nuts_run_many.restype = C.c_long
        L.bfo_nuts_run_many.argtypes = [C.POINTER(_Density), C.c_int, _dp, C.c_uint64, C.c_uint64, C.c_long,
                                        C.c_long, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _dp]
        L.bfo_nuts_run_chains.restype = C.c_long
        L.bfo_nuts_run_chains.argtypes = [C.POINTER(_Density), C.c_int, C.POINTER(C.POINTER(_Chain)), C.POINTER(_Rng),
                                          C.c_long, C.c_long, C.c_int, C.c_double, C.c_int, _dp, _dp]
        L.bfo_max_threads.restype = C.c_int
        L.bfo_xoshiro_seed.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
        L.bfo_xoshiro_next.restype = C.c_uint64
        L.bfo_xoshiro_next.argtypes = [C.POINTER(C.c_uint64)]
        L.bfo_rng_uniform.restype = C.c_double
        L.bfo_rng_uniform.argtypes = [C.POINTER(_Rng)]
        L.bfo_rng_normal.argtypes = [C.POINTER(_Rng), _dp, C.c_int]
        for n in ('quadratic_f', 'quadratic_j', 'cubic_2_f', 'cubic_2_j', 'cubic_3_f', 'cubic_3_j'):
            getattr(L, 'bfo_' + n).argtypes = [_dp, _dp, _dp, C.c_int, C.c_int]
        for n in ('lsq_quadratic', 'lsq_cubic_2', 'lsq_cubic_3'):
            getattr(L, 'bfo_' + n).argtypes = [_dp, _dp, C.c_int, C.c_int]
        for n in ('set_quadratic', 'set_cubic_2', 'set_cubic_3'):
            getattr(L, 'bfo_' + n).argtypes = [_dp, _dp, C.c_int]
        for n in ('from_original_f', 'from_original_j', 'from_original_jj'):
            getattr(L, 'bfo_' + n).argtypes = [_dp, _dp, _dp, _u8p, C.c_size_t]
            getattr(L, 'bfo_' + n).restype = C.c_int
        for n in ('to_original_f', 'to_original_j', 'to_original_jj'):
            getattr(L, 'bfo_' + n).argtypes = [_dp, _dp, _dp, _u8p, C.c_size_t]
            getattr(L, 'bfo_' + n).restype = None
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


class _Keep:
    """Owns the NumPy buffers a C struct points into."""

    def __init__(self):
        self.refs = []

    def f64(self, a):
        a = _f64(a)
        self.refs.append(a)
        return _p(a)

    def i32(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        self.refs.append(a)
        return a.ctypes.data_as(_ip)

    def u8(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        self.refs.append(a)
        return a.ctypes.data_as(_u8p)


def _poly_struct(poly, keep):
    cfgs = (_PolyConfig * len(poly['configs']))()
    for i, cf in enumerate(poly['configs']):
        im = np.asarray(cf['input_mask'])
        om = np.asarray(cf['output_mask'])
        cfgs[i].order = ORDERS[cf['order']]
        cfgs[i].n_in = im.size
        cfgs[i].n_out = om.size
        cfgs[i].in_mask = keep.i32(im)
        cfgs[i].out_mask = keep.i32(om)
        cfgs[i].coef = keep.f64(cf['coef'])
    keep.refs.append(cfgs)
    pm = _PolyModel()
    pm.input_size = int(poly['input_size'])
    pm.output_size = int(poly['output_size'])
    pm.n_config = len(poly['configs'])
    pm.configs = cfgs
    all_linear = all(cf['order'] == 'linear' for cf in poly['configs'])
    pm.use_bound = int(bool(poly.get('use_bound', False)) and not all_linear)
    if pm.use_bound:
        pm.mu = keep.f64(poly['mu'])
        pm.hess = keep.f64(poly['hess'])
        pm.alpha = float(poly['alpha'])
        pm.f_mu = keep.f64(poly['f_mu'])
    return pm


def density_struct(spec):
    keep = _Keep()
    dn = _Density()
    dn.d = int(spec['d'])
    if spec.get('ranges') is not None:
        dn.ranges = keep.f64(spec['ranges'])
        hb = spec.get('hard_bounds')
        dn.hard_bounds = keep.u8(hb if hb is not None else np.zeros((dn.d, 2), np.uint8))
    if spec.get('su_lo') is not None:
        dn.su_lo = keep.f64(spec['su_lo'])
        dn.su_diff = keep.f64(spec['su_diff'])
    dn.poly = _poly_struct(spec['poly'], keep)
    chi2 = spec.get('chi2')
    if chi2 is None and dn.poly.output_size != 1:
        raise ValueError('the density surrogate must have output_size 1 (or a chi2 stage for its outputs).')
    if chi2 is not None:   # {'y' (m,), 'prec' (m,m) | 'prec_diag' (m,), 'logp0'}; optional spec['prior'] = {'mu', 'prec_diag' (d,), 'c0'}
        if spec.get('link') is not None:
            raise ValueError('a density has a link or a chi2 stage, not both.')
        m = dn.poly.output_size
        dn.link_kind = 2
        dn.chi2_y = keep.f64(np.asarray(chi2['y'], dtype=np.float64).reshape(m))
        if (chi2.get('prec') is None) == (chi2.get('prec_diag') is None):
            raise ValueError('give me exactly one of prec and prec_diag.')
        if chi2.get('prec') is not None:
            dn.chi2_prec = keep.f64(np.asarray(chi2['prec'], dtype=np.float64).reshape(m, m))
        else:
            dn.chi2_prec_diag = keep.f64(np.asarray(chi2['prec_diag'], dtype=np.float64).reshape(m))
        dn.link_logp0 = float(chi2.get('logp0', 0.))
        prior = spec.get('prior')
        if prior is not None:
            dn.prior_mu = keep.f64(np.asarray(prior['mu'], dtype=np.float64).reshape(dn.d))
            dn.prior_prec = keep.f64(np.asarray(prior['prec_diag'], dtype=np.float64).reshape(dn.d))
            dn.prior_c0 = float(prior.get('c0', 0.))
    elif spec.get('prior') is not None:
        raise ValueError('a prior stage needs a chi2 stage before it.')
    dn.use_decay = int(bool(spec.get('use_decay', False)))
    if dn.use_decay:
        dn.decay_mu = keep.f64(spec['decay_mu'])
        dn.decay_hess = keep.f64(spec['decay_hess'])
        dn.decay_alpha2 = float(spec['decay_alpha2'])
        dn.decay_gamma = float(spec['decay_gamma'])
    link = spec.get('link')
    if link is not None:  # {'kind': 'gaussian', 'y', 'prec', 'logp0'}: logp = logp0 - prec (m - y)^2 / 2 of the surrogate's output m
        if link['kind'] != 'gaussian':
            raise ValueError('unknown link kind.')
        dn.link_kind, dn.link_y, dn.link_prec = 1, float(link['y']), float(link['prec'])
        dn.link_logp0 = float(link.get('logp0', 0.))
    return dn, keep


def poly_fun_and_jac(poly, x):
    """PolyModel._fun_and_jac for one point or a batch: returns f (..., m), j (..., m, d)."""
    keep = _Keep()
    pm = _poly_struct(poly, keep)
    x = _f64(x)
    single = x.ndim == 1
    x2 = np.atleast_2d(x)
    m, d = pm.output_size, pm.input_size
    f = np.empty((x2.shape[0], m))
    j = np.empty((x2.shape[0], m, d))
    L = lib()
    for i in range(x2.shape[0]):
        L.bfo_poly_fun_and_jac(C.byref(pm), _p(x2[i]), _p(f[i]), _p(j[i]))
    return (f[0], j[0]) if single else (f, j)


def logp_and_grad(spec, x, original_space=False, tuned=False):
    """Density.logp_and_grad(x, original_space) for one point or a batch (tuned: through bf_cpu_tuned.c, the evaluation
    bench.py's CPU baseline times)."""
    dn, keep = density_struct(spec)
    if tuned:
        f = lib().bfo_tuned_prepare
        f.restype = C.c_int
        if f(C.byref(dn)) != 0:
            raise ValueError('the tuned evaluation does not cover this density')
    x = _f64(x)
    single = x.ndim == 1
    x2 = np.atleast_2d(x)
    logp = np.empty(x2.shape[0])
    grad = np.empty_like(x2)
    L = lib()
    for i in range(x2.shape[0]):
        L.bfo_logp_and_grad(C.byref(dn), _p(x2[i]), int(original_space), _p(logp[i:i + 1]), _p(grad[i]))
    if tuned:
        L.bfo_tuned_clear()
    return (logp[0], grad[0]) if single else (logp, grad)


def leapfrog(spec, var, eps, q, p, grad):
    """CpuLeapfrogIntegrator._step with a diagonal metric. Returns dict(q,p,v,grad,energy,logp)."""
    dn, keep = density_struct(spec)
    d = dn.d
    out = [np.empty(d) for _ in range(4)]
    e = np.empty(1)
    lp = np.empty(1)
    lib().bfo_leapfrog(C.byref(dn), _p(_f64(var)), float(eps), _p(_f64(q)), _p(_f64(p)), _p(_f64(grad)),
                       _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _p(e), _p(lp))
    return dict(q=out[0], p=out[1], v=out[2], grad=out[3], energy=e[0], logp=lp[0])


def leapfrog_full(spec, cov, eps, q, p, grad):
    """CpuLeapfrogIntegrator._step with QuadMetricFull(cov). Returns dict(q,p,v,grad,energy,logp)."""
    dn, keep = density_struct(spec)
    d = dn.d
    out = [np.empty(d) for _ in range(4)]
    e = np.empty(1)
    lp = np.empty(1)
    lib().bfo_leapfrog_full(C.byref(dn), _p(_f64(cov)), float(eps), _p(_f64(q)), _p(_f64(p)), _p(_f64(grad)),
                            _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _p(e), _p(lp))
    return dict(q=out[0], p=out[1], v=out[2], grad=out[3], energy=e[0], logp=lp[0])


def xoshiro_seed(seed, stream):
    s = (C.c_uint64 * 4)()
    lib().bfo_xoshiro_seed(int(seed), int(stream), s)
    return np.array(list(s), dtype=np.uint64)


class Chain:
    """Per-chain sampler state (step size, metric, position); mirrors one initialised _HTrace."""

    def __init__(self, x0, step_size=1., adapt_step_size=True, target_accept=0.8, gamma=0.05, k=0.75,
                 t_0=10., metric=None, adapt_metric=True, initial_mean=None, initial_weight=10.,
                 adapt_window=60, update_window=1, doubling=True):
        x0 = _f64(x0)
        self.d = x0.size
        # metric: None / 1-d variances (diagonal), or 'full' / 2-d covariance (QuadMetricFull(Adapt))
        cov0 = None
        full = isinstance(metric, str) and metric == 'full'
        if isinstance(metric, str):
            metric = None
        elif metric is not None and np.ndim(metric) == 2:
            cov0, metric, full = _f64(metric), None, True
        mv = None if metric is None else _f64(metric)
        im = None if initial_mean is None else _f64(initial_mean)
        self._c = lib().bfo_chain_new(self.d, _p(x0), float(step_size), int(adapt_step_size),
                                      float(target_accept), float(gamma), float(k), float(t_0),
                                      _p(mv) if mv is not None else None, int(adapt_metric),
                                      _p(im) if im is not None else None, float(initial_weight),
                                      int(adapt_window), int(update_window), int(doubling))
        if full:
            if lib().bfo_chain_set_full(self._c, _p(cov0) if cov0 is not None else None):
                raise ValueError('the input covariance is not positive definite.')

    def mat(self, name):
        return np.ctypeslib.as_array(getattr(self._c.contents, name), shape=(self.d, self.d)).copy()

    def __del__(self):
        try:
            lib().bfo_chain_free(self._c)
        except Exception:
            pass

    def vec(self, name):
        return np.ctypeslib.as_array(getattr(self._c.contents, name), shape=(self.d,)).copy()

    def scalar(self, name):
        return getattr(self._c.contents, name)


def make_rng(kind, seed=0, stream=0, normals=None, uniforms=None, state=None):
    r = _Rng()
    keep = []
    if kind == 'xoshiro':
        r.kind = 0
        s = xoshiro_seed(seed, stream) if state is None else np.asarray(state, dtype=np.uint64)
        for i in range(4):
            r.s[i] = int(s[i])
    elif kind == 'replay':
        r.kind = 1
        n = _f64(normals)
        u = _f64(uniforms)
        keep += [n, u]
        r.normals, r.uniforms = _p(n), _p(u)
        r.n_normals, r.n_uniforms = n.size, u.size
    else:
        raise ValueError(kind)
    return r, keep


def nuts_run(spec, chain, rng, n_run, n_warmup, max_treedepth=10, max_change=1000.):
    """Run n_run NUTS iterations of one chain. rng = (struct, keepalive) from make_rng."""
    dn, keep = density_struct(spec)
    samples = np.empty((n_run, chain.d))
    stats = np.empty((n_run, len(NSTATS)))
    rc = lib().bfo_nuts_run(C.byref(dn), chain._c, C.byref(rng[0]), int(n_run), int(n_warmup),
                            int(max_treedepth), float(max_change), _p(samples), _p(stats))
    if rc:
        raise RuntimeError({-1: 'bad initial energy', -2: 'replay stream exhausted',
                            -3: "logp can't be nan"}.get(rc, str(rc)))
    return samples, {k: stats[:, i].copy() for i, k in enumerate(NSTATS)}


def hmc_run(spec, chain, rng, n_run, n_warmup, n_int_step=32, max_change=1000.):
    dn, keep = density_struct(spec)
    samples = np.empty((n_run, chain.d))
    stats = np.empty((n_run, len(HSTATS)))
    rc = lib().bfo_hmc_run(C.byref(dn), chain._c, C.byref(rng[0]), int(n_run), int(n_warmup), int(n_int_step),
                           float(max_change), _p(samples), _p(stats))
    if rc:
        raise RuntimeError({-1: 'bad initial energy', -2: 'replay stream exhausted'}.get(rc, str(rc)))
    return samples, {k: stats[:, i].copy() for i, k in enumerate(HSTATS)}


def nuts_run_many(spec, x0, seed, n_run, n_warmup, first_stream=0, max_treedepth=10, max_change=1000.,
                  step_size=1., target_accept=0.8, n_threads=0):
    """Fresh default chains, xoshiro streams (seed, first_stream + i); one chain per OpenMP thread."""
    dn, keep = density_struct(spec)
    x0 = _f64(x0)
    n_chain, d = x0.shape
    samples = np.empty((n_chain, n_run, d))
    stats = np.empty((n_chain, n_run, len(NSTATS)))
    total = lib().bfo_nuts_run_many(C.byref(dn), n_chain, _p(x0), int(seed), int(first_stream), int(n_run),
                                    int(n_warmup), int(max_treedepth), float(max_change), float(step_size),
                                    float(target_accept), int(n_threads), _p(samples), _p(stats))
    if total < 0:
        raise RuntimeError('oracle chain failed with code %d' % total)
    return samples, {k: stats[:, :, i].copy() for i, k in enumerate(NSTATS)}, int(total)


# ---------------------------------------------------------------------------------------------------
# NumPy/SciPy restatements of reference code that is NumPy/SciPy in the reference as well
# ---------------------------------------------------------------------------------------------------

def a_size(order, n):
    """PolyConfig._a_shape, modules/poly.py:109-129."""
    return {'linear': n + 1, 'quadratic': n * (n + 1) // 2, 'cubic-2': n * n,
            'cubic-3': n * (n - 1) * (n - 2) // 6}[order]


def design_block(order, x):
    """One block of the design matrix, modules/poly.py:537-564 -> modules/_poly.pyx:143-177."""
    x = _f64(x)
    n_pts, n = x.shape
    out = np.empty((n_pts, a_size(order, n)))
    if order == 'linear':
        out[:, 0] = 1
        out[:, 1:] = x
    else:
        fn = getattr(lib(), 'bfo_lsq_' + order.replace('-', '_'))
        if out.size:
            fn(_p(x), _p(out), n_pts, n)
    return out


def dense_coef(order, a, n):
    """PolyConfig._set for one output, modules/poly.py:131-158 -> _poly.pyx:183-214 (zero background)."""
    a = _f64(a)
    if order == 'linear':
        return a.copy()
    shape = (n, n, n) if order == 'cubic-3' else (n, n)
    coef = np.zeros(shape)
    getattr(lib(), 'bfo_set_' + order.replace('-', '_'))(_p(a), _p(coef), n)
    return coef


def poly_fit(poly, x, y, logp=None, w=None, bound_options=None):
    """PolyModel.fit (modules/poly.py:505-589): returns a new poly spec with fitted coefficients.

    ``poly['configs']`` gives order/input_mask/output_mask (coef ignored). ``bound_options`` is a dict with
    use_bound, alpha, alpha_p, center_max (modules/poly.py:232-260)."""
    from scipy.linalg import lstsq
    x = _f64(x)
    y = _f64(y)
    d, m = int(poly['input_size']), int(poly['output_size'])
    cfgs = [dict(order=c['order'], input_mask=np.asarray(c['input_mask']), output_mask=np.asarray(c['output_mask']))
            for c in poly['configs']]
    n_param = sum(a_size(c['order'], c['input_mask'].size) for c in cfgs)
    if x.shape[0] < n_param:
        raise ValueError('I need at least {} points, but you only gave me {}.'.format(n_param, x.shape[0]))
    shapes = {'linear': lambda n: (n + 1,), 'quadratic': lambda n: (n, n), 'cubic-2': lambda n: (n, n),
              'cubic-3': lambda n: (n, n, n)}
    for c in cfgs:
        c['coef'] = np.zeros((c['output_mask'].size,) + shapes[c['order']](c['input_mask'].size))
    for ii in range(m):
        blocks, owners = [], []
        for order in ('linear', 'quadratic', 'cubic-2', 'cubic-3'):  # recipe column order, poly.py:294-338
            for c in cfgs:
                if c['order'] == order and ii in c['output_mask']:
                    blocks.append(design_block(order, x[:, c['input_mask']]))
                    owners.append(c)
        A = np.concatenate(blocks, axis=-1)
        b = np.copy(y[:, ii])
        if w is not None:
            b *= w
            A *= np.asarray(w)[:, np.newaxis]
        sol = lstsq(A, b)[0]
        k = 0
        for c, blk in zip(owners, blocks):
            qq = int(np.argwhere(c['output_mask'] == ii)[0, 0])
            c['coef'][qq] = dense_coef(c['order'], sol[k:k + blk.shape[1]], c['input_mask'].size)
            k += blk.shape[1]
    out = dict(input_size=d, output_size=m, configs=cfgs, use_bound=False)
    bo = dict(use_bound=True, alpha=None, alpha_p=100., center_max=True)
    bo.update(bound_options or {})
    all_linear = all(c['order'] == 'linear' for c in cfgs)
    if bo['use_bound'] and not all_linear:
        out.update(set_bound(out, x, logp, bo))
    return out


def _mu_hess_alpha(x, alpha, alpha_p):
    """Shared by PolyModel._set_bound (poly.py:268-276) and Density._set_decay (density.py:802-811)."""
    mu = np.mean(x, axis=0)
    hess = np.linalg.inv(np.cov(x, rowvar=False))
    if alpha_p is not None:
        beta = np.einsum('ij,jk,ik->i', x - mu, hess, x - mu)**0.5
        if alpha_p < 100.:
            alpha = np.percentile(beta, alpha_p)
        else:
            alpha = np.max(beta) * alpha_p / 100.
    return mu, hess, alpha


def set_bound(poly, x, logp, bo):
    """PolyModel._set_bound, modules/poly.py:262-292."""
    x = _f64(x)
    mu, hess, alpha = _mu_hess_alpha(x, bo.get('alpha'), bo.get('alpha_p', 100.))
    if bo.get('center_max', True) and logp is not None:
        mu_f = x[np.argmax(np.asarray(logp))]
    else:
        mu_f = mu
    tmp = dict(poly)
    tmp['use_bound'] = False
    f_mu, _ = poly_fun_and_jac(tmp, mu_f)
    return dict(use_bound=True, mu=mu, hess=hess, alpha=float(alpha), f_mu=f_mu)


def set_decay(x, alpha=None, alpha_p=150., gamma=0.1):
    """Density._set_decay, core/density.py:796-811 (+ set_decay_options :761-794)."""
    mu, hess, alpha = _mu_hess_alpha(_f64(x), alpha, alpha_p)
    return dict(use_decay=True, decay_mu=mu, decay_hess=hess, decay_alpha2=float(alpha)**2,
                decay_gamma=float(gamma))


class ChainSet:
    """Persistent chains + xoshiro streams for timing phases separately (bench.py cpu_baseline)."""

    def __init__(self, spec, x0, seed, first_stream=0, tuned=False, **chain_kw):
        self.dn, self._keep = density_struct(spec)
        self.tuned = False
        if tuned:  # bench.py's baseline: the same density through bf_cpu_tuned.c (one dense matvec pair, no allocation)
            f = lib().bfo_tuned_prepare
            f.restype = C.c_int
            self.tuned = f(C.byref(self.dn)) == 0
        x0 = _f64(x0)
        self.n_chain, self.d = x0.shape
        self.chains = [Chain(x0[i], **chain_kw) for i in range(self.n_chain)]
        self._cptr = (C.POINTER(_Chain) * self.n_chain)(*[c._c for c in self.chains])
        self._rngs = (_Rng * self.n_chain)()
        for i in range(self.n_chain):
            self._rngs[i].kind = 0
            s = xoshiro_seed(seed, first_stream + i)
            for k in range(4):
                self._rngs[i].s[k] = int(s[k])

    def run(self, n_run, n_warmup, max_treedepth=10, max_change=1000., n_threads=0):
        samples = np.empty((self.n_chain, n_run, self.d))
        stats = np.empty((self.n_chain, n_run, len(NSTATS)))
        total = lib().bfo_nuts_run_chains(C.byref(self.dn), self.n_chain, self._cptr, self._rngs, int(n_run),
                                          int(n_warmup), int(max_treedepth), float(max_change), int(n_threads),
                                          _p(samples), _p(stats))
        if total < 0:
            raise RuntimeError('oracle chain failed with code %d' % total)
        return samples, {k: stats[:, :, i].copy() for i, k in enumerate(NSTATS)}, int(total)


    def close(self):
        if self.tuned:
            lib().bfo_tuned_clear()
            self.tuned = False


def max_threads():
    return int(lib().bfo_max_threads())


# ---- evidence path ------------------------------------------------------------
def spline_apply(mode, c, x, y, pts):
    """utils/_cubic.pyx:188-336 on one spline: mode 'evaluate' | 'derivative' | 'solve'."""
    c, x, y, pts = _f64(c), _f64(x), _f64(y), _f64(pts)
    out = np.empty_like(pts)
    f = lib().bfo_spline_apply
    f.restype = None
    f(C.c_int({'evaluate': 0, 'derivative': 1, 'solve': 2}[mode]), _p(c), _p(x), _p(y), C.c_int(x.size), _p(pts), _p(out),
      C.c_size_t(pts.size))
    return out


def kde_cdf(data, weights, h, pts):
    """kde.cdf (utils/kde.py:322-354) of a 1-d weighted Gaussian KDE with bandwidth h."""
    data, w, pts = _f64(data), _f64(weights), _f64(np.atleast_1d(pts))
    out = np.empty_like(pts)
    f = lib().bfo_kde_cdf
    f.restype = None
    f(_p(data), _p(w), C.c_size_t(data.size), C.c_double(h), _p(pts), _p(out), C.c_size_t(pts.size))
    return out


def kde_bandwidth(data, weights, bw_factor=1.):
    """Bandwidth of the reference's 1-d kde (utils/kde.py:85-151): Scott's factor neff^(-1/5) times bw_factor times the
    square root of the weighted, unbiased variance."""
    w = np.asarray(weights, dtype=np.float64)
    w = w / w.sum()
    neff = 1. / np.sum(w**2)
    cov = float(np.cov(np.asarray(data, dtype=np.float64), bias=False, aweights=w))
    return np.sqrt(cov) * neff**(-1. / 5) * bw_factor


def bridge_score(logr, a, b):
    """evidence/bridge.py:44-49."""
    from scipy.special import logsumexp
    a, b = np.asarray(a), np.asarray(b)
    c = logsumexp(logr + a - logsumexp(np.array((logr + a, np.zeros_like(a))), axis=0))
    d = logsumexp(-logr + b - logsumexp(np.array((-logr + b, np.zeros_like(b))), axis=0))
    return c - d


# ---- tempered samplers ---------------------------------------------------------
def gaussian_base_spec(mean, cov, logz_offset=0.):
    """Density spec (no bound) of the Gaussian base density N(mean, cov) as a quadratic polynomial:
    logp = c0 + lin . x + sum_{j<=k} a[j,k] x_j x_k."""
    mean = np.asarray(mean, dtype=np.float64)
    prec = np.linalg.inv(np.atleast_2d(cov))
    d = mean.size
    lin = prec @ mean
    c0 = -0.5 * mean @ prec @ mean - 0.5 * (d * np.log(2 * np.pi) + np.linalg.slogdet(np.atleast_2d(cov))[1]) + logz_offset
    quad = np.zeros((d, d))
    iu = np.triu_indices(d)
    quad[iu] = (-0.5 * prec)[iu] * np.where(iu[0] == iu[1], 1., 2.)
    poly = dict(input_size=d, output_size=1, use_bound=False,
                configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.array([0]), coef=np.concatenate(([c0], lin))[None]),
                         dict(order='quadratic', input_mask=np.arange(d), output_mask=np.array([0]), coef=quad[None])])
    return dict(d=d, ranges=None, hard_bounds=None, su_lo=None, su_diff=None, poly=poly, use_decay=False)


def tempered_states(spec, base_spec, logxi, var, q0, p0, u0, v0, eps):
    """TState rows (q, p, u, v, weight, energy, logp) after compute_state and after each step (integration.py:132-222)."""
    dn, k1 = density_struct(spec)
    bs, k2 = density_struct(base_spec)
    var, q0, p0, eps = _f64(var), _f64(q0), _f64(p0), _f64(eps)
    d = q0.size
    out = np.empty((eps.size + 1, 2 * d + 5))
    f = lib().bfo_tempered_states
    f.restype = None
    f(C.byref(dn), C.byref(bs), C.c_double(logxi), _p(var), _p(q0), _p(p0), C.c_double(u0), C.c_double(v0), _p(eps), C.c_int(eps.size),
      _p(out))
    return dict(q=out[:, :d], p=out[:, d:2 * d], u=out[:, 2 * d], v=out[:, 2 * d + 1], weight=out[:, 2 * d + 2],
                energy=out[:, 2 * d + 3], logp=out[:, 2 * d + 4])


def tnuts_run(spec, base_spec, logxi, chain, rng, u0, n_run, n_warmup, max_treedepth=10, max_change=1000.):
    """TNUTS (samplers/tnuts.py, base_hmc.py:220-262).  Returns samples, stats dict (NUTS fields + 'u', 'weight'), last u."""
    dn, k1 = density_struct(spec)
    bs, k2 = density_struct(base_spec)
    samples = np.empty((n_run, chain.d))
    stats = np.empty((n_run, len(NSTATS)))
    st_t = np.empty((n_run, 2))
    u = C.c_double(u0)
    f = lib().bfo_tnuts_run
    f.restype = C.c_int
    rc = f(C.byref(dn), C.byref(bs), C.c_double(logxi), chain._c, C.byref(rng[0]), C.byref(u), C.c_long(int(n_run)),
           C.c_long(int(n_warmup)), C.c_int(int(max_treedepth)), C.c_double(max_change), _p(samples), _p(stats), _p(st_t))
    if rc:
        raise RuntimeError({-1: 'bad initial energy', -2: 'replay stream exhausted', -3: "logp can't be nan"}.get(rc, str(rc)))
    st = {k: stats[:, i].copy() for i, k in enumerate(NSTATS)}
    st['u'], st['weight'] = st_t[:, 0].copy(), st_t[:, 1].copy()
    return samples, st, u.value
