"""Device context and device-resident surrogate density (host side of include/bfhip.h).

PyTorch-ROCm is used for device memory and streams only; all arithmetic happens in libbfhip.so.
"""
import ctypes as C

import numpy as np

from . import _lib

__all__ = ['DeviceContext', 'DeviceDensity', 'DevicePolyModel', 'density_desc_from_spec', 'pipeline_desc_from_spec',
           'polymodel_desc_from_poly', 'get_context']

_ORDERS = ('linear', 'quadratic', 'cubic-2', 'cubic-3')


def _torch():
    import torch
    return torch


class DeviceContext:
    """One bfhip_ctx bound to a HIP device and stream (include/bfhip.h: bfhip_ctx_create)."""

    def __init__(self, device=0, stream=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError('bayesfast_amd needs a ROCm GPU (gfx950); none is visible.')
        self.device = torch.device('cuda', device if isinstance(device, int) else torch.device(device).index or 0)
        self._lib = _lib.lib()
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            s = stream if stream is not None else torch.cuda.current_stream(self.device)
            self.stream = s
            _lib.check(self._lib.bfhip_ctx_create(C.byref(self._ctx), self.device.index, C.c_void_p(s.cuda_stream)))

    def close(self):
        if self._ctx:
            self._lib.bfhip_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._ctx

    def set_stream(self, stream):
        self.stream = stream
        _lib.check(self._lib.bfhip_ctx_set_stream(self._ctx, C.c_void_p(stream.cuda_stream)))

    def synchronize(self):
        _lib.check(self._lib.bfhip_ctx_synchronize(self._ctx))

    def tensor(self, a, dtype=None):
        torch = _torch()
        if isinstance(a, torch.Tensor):
            t = a.to(self.device)
            if dtype is not None:
                t = t.to(dtype)
            return t.contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device).contiguous()

    def empty(self, shape, dtype=None):
        torch = _torch()
        return torch.empty(shape, dtype=dtype or torch.float64, device=self.device)

    def zeros(self, shape, dtype=None):
        torch = _torch()
        return torch.zeros(shape, dtype=dtype or torch.float64, device=self.device)


_contexts = {}


def get_context(device=None):
    """Process-wide default context of a device (default: torch's current device, i.e. the one a rank selected with
    ``torch.cuda.set_device(local_rank)``)."""
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device()
    if device not in _contexts:
        _contexts[device] = DeviceContext(device)
    ctx = _contexts[device]
    # the default context follows torch's current stream: the torch ops of the host side (cat, copies, conversions) and
    # the library's launches then share one stream whatever ``torch.cuda.stream(...)`` block the caller is in; work
    # already queued on the previous stream is ordered before anything that follows
    cur = torch.cuda.current_stream(ctx.device)
    if cur.cuda_stream != ctx.stream.cuda_stream:
        ev = torch.cuda.Event()
        ev.record(ctx.stream)
        cur.wait_event(ev)
        ctx.set_stream(cur)
    return ctx


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# tuning / test switch: fold Surrogate.input_scales into linear + quadratic surrogates at upload (density_desc_from_spec)
FOLD_INPUT_SCALES = __import__('os').environ.get('BFHIP_NO_SU_FOLD', '') in ('', '0')
FOLD_MAX_OFFSET = 30.
FOLD_MAX_OFFSET_CUBIC = 10.   # (cubic terms: (1 + |lo| / diff)^3)


def folds_input_scales(spec):
    """Whether density_desc_from_spec folds the spec's Surrogate.input_scales into the polynomial (see there): a linear +
    quadratic surrogate whose range lies within FOLD_MAX_OFFSET widths of the origin."""
    if spec.get('su_lo') is None or not FOLD_INPUT_SCALES:
        return False
    cubic = any(cf['order'] in ('cubic-2', 'cubic-3') for cf in spec['poly']['configs'])
    lo, diff = np.asarray(spec['su_lo'], dtype=np.float64), np.asarray(spec['su_diff'], dtype=np.float64)
    return bool(float(np.max(np.abs(lo) / np.abs(diff))) <= (FOLD_MAX_OFFSET_CUBIC if cubic else FOLD_MAX_OFFSET))


def density_desc_from_spec(spec):
    """Flatten a density spec (dict, see below) into a bfhip_density_desc + the arrays it points to.

    spec = {'d', 'ranges', 'hard_bounds', 'su_lo', 'su_diff',
            'poly': {'input_size', 'output_size' (=1), 'configs': [{'order', 'input_mask', 'output_mask', 'coef'}],
                     'use_bound', 'mu', 'hess', 'alpha', 'f_mu'},
            'use_decay', 'decay_mu', 'decay_hess', 'decay_alpha2', 'decay_gamma',
            'link': None | {'kind': 'gaussian', 'y', 'prec', 'logp0'}}

    The PolyConfig masks are scattered to the full input here, which is what PolyModel._fun_and_jac does
    on every call (modules/poly.py:474-477)."""
    d = int(spec['d'])
    poly = spec['poly']
    if int(poly['output_size']) != 1 or int(poly['input_size']) != d:
        raise ValueError('the device density needs a surrogate with output_size 1 and input_size d.')
    keep = []

    def f64(a, shape=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if shape is not None and a.shape != shape:
            raise ValueError('expected shape {}, got {}.'.format(shape, a.shape))
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_double))

    ds = _lib.DensityDesc()
    ds.d = d
    if spec.get('ranges') is not None:
        ds.ranges = f64(spec['ranges'], (d, 2))
        hb = spec.get('hard_bounds')
        if hb is not None:
            hb = np.ascontiguousarray(hb, dtype=np.uint8)
            if hb.shape != (d, 2):
                raise ValueError('hard_bounds should have shape (d, 2).')
            keep.append(hb)
            ds.hard_bounds = hb.ctypes.data_as(C.POINTER(C.c_uint8))
    su_lo = su_diff = None
    if spec.get('su_lo') is not None:
        su_lo = np.asarray(spec['su_lo'], dtype=np.float64).reshape(d)
        su_diff = np.asarray(spec['su_diff'], dtype=np.float64).reshape(d)
    c0 = 0.
    lin = np.zeros(d)
    quad = np.zeros((d, d))
    cubic2 = np.zeros((d, d))
    cubic3 = None
    has = dict.fromkeys(_ORDERS, False)
    for cf in poly['configs']:
        order = cf['order']
        if order not in _ORDERS:
            raise ValueError('unexpected PolyConfig order "{}".'.format(order))
        im = np.asarray(cf['input_mask'], dtype=np.int64)
        coef = np.asarray(cf['coef'], dtype=np.float64)
        if has[order]:
            raise ValueError('multiple {} PolyConfig(s) share the output variable.'.format(order))
        has[order] = True
        n = im.size
        if order == 'linear':
            c0 += coef[0, 0]
            lin[im] += coef[0, 1:]
        elif order == 'quadratic':
            iu = np.triu_indices(n)  # only j <= k is defined in the reference's blocks (modules/poly.py:146)
            quad[im[iu[0]], im[iu[1]]] += coef[0][iu]
        elif order == 'cubic-2':
            cubic2[np.ix_(im, im)] += coef[0]
        else:
            if cubic3 is None:
                cubic3 = np.zeros((d, d, d))
            j, k, l = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij')
            sel = (j < k) & (k < l)
            cubic3[im[j[sel]], im[k[sel]], im[l[sel]]] += coef[0][sel]
    mu_b = hess_b = None
    if poly.get('use_bound', False):
        mu_b, hess_b = np.asarray(poly['mu'], dtype=np.float64).reshape(d), np.asarray(poly['hess'], dtype=np.float64).reshape(d, d)
    if su_lo is not None and folds_input_scales(spec):
        # Surrogate.input_scales (module.py:190-226: x_s = (x - lo) / diff before the polynomial, the gradient divided by diff
        # after it) folded into a linear + quadratic polynomial's coefficients and its bound: with D = diag(1 / diff),
        #   c0 + l . x_s + x_s^T A x_s = c0' + l' . x + x^T A' x,   A' = D A D,  l' = D l - (A' + A'^T) lo,
        #   c0' = c0 - (D l) . lo + lo^T A' lo;   (x_s - mu)^T H (x_s - mu) = (x - mu')^T H' (x - mu'),  mu' = lo + diff mu,  H' = D H D
        # -- the same function of x (the bound's radius, the extrapolation outside it and the gradient with it: every term of
        # modules/poly.py:480-503 is D times its scaled-space form), equal to rounding.  The device then sees a surrogate
        # WITHOUT input scaling, which every fused sampler kernel takes (with it, only the generic instantiation of the sliced
        # kernel does).  A range that lies far from the origin in units of its own width (|lo| / diff > FOLD_MAX_OFFSET) keeps the
        # scaling as a device-side step: the folded polynomial is the scaled one expanded around x = 0, and its terms are
        # (1 + |lo| / diff)^2 (cubic configs: ^3, with a tighter limit) times the size of their sum -- 1e3 at the limit, i.e. 1e-13 relative.
        dinv = 1. / su_diff
        if has['cubic-2'] or has['cubic-3']:
            # with cubic configs: the third-order Taylor expansion of p(D (x - lo)) around x = 0 (exact: p is a cubic).  Value,
            # gradient and Hessian of p at a = -D lo in the scaled space; the third derivatives do not depend on the shift.
            a = -su_lo * dinv
            c3 = cubic3 if cubic3 is not None else np.zeros((d, d, d))
            s3 = sum(np.transpose(c3, pm) for pm in ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)))
            qs = quad + quad.T
            c2a = cubic2 @ a
            p_a = c0 + float(lin @ a) + float(a @ quad @ a) + float((a * a) @ c2a) + float(np.einsum('jkl,j,k,l->', c3, a, a, a))
            g_a = lin + qs @ a + 2. * a * c2a + cubic2.T @ (a * a) + 0.5 * np.einsum('ikl,k,l->i', s3, a, a)
            m1 = 2. * a[:, None] * cubic2
            h_a = qs + 2. * np.diag(c2a) + m1 + m1.T + np.einsum('iml,l->im', s3, a)
            c0 = p_a
            lin = g_a * dinv
            hs = h_a * np.outer(dinv, dinv)
            quad = np.triu(hs, 1) + 0.5 * np.diag(np.diag(hs))
            cubic2 = cubic2 * np.outer(dinv * dinv, dinv)
            if cubic3 is not None:
                cubic3 = cubic3 * dinv[:, None, None] * dinv[None, :, None] * dinv[None, None, :]
            has['quadratic'] = True   # (the shift makes quadratic terms of cubic ones)
        else:
            quad = quad * np.outer(dinv, dinv)
            dl = lin * dinv
            c0 = c0 - float(dl @ su_lo) + float(su_lo @ quad @ su_lo)
            lin = dl - (quad + quad.T) @ su_lo
        if mu_b is not None:
            mu_b = su_lo + su_diff * mu_b
            hess_b = hess_b * np.outer(dinv, dinv)
        su_lo = su_diff = None
    if su_lo is not None:
        ds.su_lo = f64(su_lo, (d,))
        ds.su_diff = f64(su_diff, (d,))
    ds.c0 = c0
    ds.lin = f64(lin)
    if has['quadratic']:
        ds.quad = f64(quad)
    if has['cubic-2']:
        ds.cubic2 = f64(cubic2)
    if has['cubic-3']:
        ds.cubic3 = f64(cubic3)
    all_linear = not (has['quadratic'] or has['cubic-2'] or has['cubic-3'])
    if poly.get('use_bound', False) and not all_linear:
        ds.use_bound = 1
        ds.mu = f64(mu_b, (d,))
        ds.hess = f64(hess_b, (d, d))
        ds.alpha = float(poly['alpha'])
        ds.f_mu = float(np.asarray(poly['f_mu']).reshape(-1)[0])
    if spec.get('use_decay', False):
        ds.use_decay = 1
        ds.decay_mu = f64(spec['decay_mu'], (d,))
        ds.decay_hess = f64(spec['decay_hess'], (d, d))
        ds.decay_alpha2 = float(spec['decay_alpha2'])
        ds.decay_gamma = float(spec['decay_gamma'])
    link = spec.get('link')
    if link is not None:  # the module downstream of the surrogate's output: {'kind': 'gaussian', 'y', 'prec', 'logp0'}
        if link.get('kind') != 'gaussian':
            raise ValueError('unknown link kind.')
        ds.link_kind, ds.link_y, ds.link_prec = 1, float(link['y']), float(link['prec'])
        ds.link_logp0 = float(link.get('logp0', 0.))
    return ds, keep


def pipeline_desc_from_spec(spec):
    """A density spec with a ``'chi2'`` stage -> ``bfhip_pipeline_desc`` + the arrays it points to.

    spec = the keys of ``density_desc_from_spec`` with a MULTI-output ``poly`` and no ``link``, plus
           'chi2': {'y' (m,), 'prec' (m, m) | 'prec_diag' (m,), 'logp0'} and optionally
           'prior': {'mu' (d,), 'prec_diag' (d,), 'c0'}  (original-space inputs; prec_diag 0 = no prior on that input)"""
    d = int(spec['d'])
    poly, chi2 = spec['poly'], spec['chi2']
    m = int(poly['output_size'])
    if int(poly['input_size']) != d:
        raise ValueError('the surrogate should have input_size d.')
    if spec.get('link') is not None:
        raise ValueError('a density has a link or a chi2 stage, not both.')
    keep = []

    def f64(a, shape=None):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if shape is not None and a.shape != shape:
            raise ValueError('expected shape {}, got {}.'.format(shape, a.shape))
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_double))

    ds = _lib.PipelineDesc()
    ds.d, ds.m = d, m
    if spec.get('ranges') is not None:
        ds.ranges = f64(spec['ranges'], (d, 2))
        hb = spec.get('hard_bounds')
        if hb is not None:
            hb = np.ascontiguousarray(hb, dtype=np.uint8)
            if hb.shape != (d, 2):
                raise ValueError('hard_bounds should have shape (d, 2).')
            keep.append(hb)
            ds.hard_bounds = hb.ctypes.data_as(C.POINTER(C.c_uint8))
    if spec.get('su_lo') is not None:
        ds.su_lo = f64(spec['su_lo'], (d,))
        ds.su_diff = f64(spec['su_diff'], (d,))
    pmd, pkeep = polymodel_desc_from_poly(poly)
    keep.append(pkeep)
    ds.model = pmd
    ds.y = f64(np.asarray(chi2['y'], dtype=np.float64).reshape(-1), (m,))
    if (chi2.get('prec') is None) == (chi2.get('prec_diag') is None):
        raise ValueError('give me exactly one of prec and prec_diag.')
    if chi2.get('prec') is not None:
        ds.prec = f64(chi2['prec'], (m, m))
    else:
        ds.prec_diag = f64(np.asarray(chi2['prec_diag'], dtype=np.float64).reshape(-1), (m,))
    ds.logp0 = float(chi2.get('logp0', 0.))
    prior = spec.get('prior')
    if prior is not None:
        ds.prior_mu = f64(np.asarray(prior['mu'], dtype=np.float64).reshape(-1), (d,))
        ds.prior_prec = f64(np.asarray(prior['prec_diag'], dtype=np.float64).reshape(-1), (d,))
        ds.prior_c0 = float(prior.get('c0', 0.))
    if spec.get('use_decay', False):
        ds.use_decay = 1
        ds.decay_mu = f64(spec['decay_mu'], (d,))
        ds.decay_hess = f64(spec['decay_hess'], (d, d))
        ds.decay_alpha2 = float(spec['decay_alpha2'])
        ds.decay_gamma = float(spec['decay_gamma'])
    return ds, keep


class DeviceDensity:
    """A surrogate log density resident on one GPU.

    Counterpart of ``Density.logp_and_grad(x, original_space, use_surrogate=True)`` (core/density.py:724-754)
    for a pipeline whose density variable is a single PolyModel output."""

    def __init__(self, spec, ctx=None):
        self.ctx = ctx if ctx is not None else get_context()
        self.d = int(spec['d'])
        self.spec = spec
        self.upload()

    def upload(self):
        """Make this density the context's current one (a context holds one density at a time).  An upload that is refused may
        have replaced the context's density already (the pipeline upload runs the density upload first): the context then has NO
        current density, and whichever density is used next uploads itself again."""
        self.ctx._current_density = None
        if self.spec.get('chi2') is not None:   # [multi-output surrogate, Gaussian likelihood, optional prior]: bfhip_pipeline_upload
            ds, keep = pipeline_desc_from_spec(self.spec)
            _lib.check(self.ctx._lib.bfhip_pipeline_upload(self.ctx.handle, C.byref(ds)))
        else:
            ds, keep = density_desc_from_spec(self.spec)
            _lib.check(self.ctx._lib.bfhip_density_upload(self.ctx.handle, C.byref(ds)))
        self.ctx._current_density = self

    def upload_if_needed(self):
        if getattr(self.ctx, '_current_density', None) is not self:
            self.upload()

    def logp_and_grad(self, x, original_space=False):
        """x: (n, d) or (d,) array/tensor -> (logp (n,), grad (n, d)) float64 device tensors."""
        torch = _torch()
        self.upload_if_needed()
        xt = self.ctx.tensor(x, torch.float64)
        single = xt.dim() == 1
        xt = xt.reshape(-1, self.d)
        n = xt.shape[0]
        logp = self.ctx.empty((n,))
        grad = self.ctx.empty((n, self.d))
        _lib.check(self.ctx._lib.bfhip_logp_grad(self.ctx.handle, n, _ptr(xt), int(bool(original_space)),
                                                 _ptr(logp), _ptr(grad)))
        return (logp[0], grad[0]) if single else (logp, grad)

    _WHICH = {'from_original': 0, 'from_original_grad': 1, 'from_original_grad2': 2, 'to_original': 3,
              'to_original_grad': 4, 'to_original_grad2': 5}

    def constraint(self, which, x):
        """Constraint transform of this density on device (core/density.py:142-163): (..., d) -> (..., d) tensor.
        Raises ValueError when a variable is out of bound in a ``from_original*`` call."""
        torch = _torch()
        self.upload_if_needed()
        xt = self.ctx.tensor(x, torch.float64)
        shape = xt.shape
        xt = xt.reshape(-1, self.d)
        out = self.ctx.empty(xt.shape)
        bad = torch.zeros((1,), dtype=torch.int32, device=self.ctx.device)
        _lib.check(self.ctx._lib.bfhip_constraint(self.ctx.handle, self._WHICH[which], xt.shape[0], _ptr(xt), _ptr(out),
                                                  _ptr(bad)))
        if which.startswith('from') and int(bad.item()):
            raise ValueError('variable #{} out of bound.'.format(int(bad.item()) - 1))
        return out.reshape(shape)

    def leapfrog(self, eps, var, q, p, grad, logp=None, energy=None, velocity=None):
        """In-place batched CpuLeapfrogIntegrator._step; all arguments float64 device tensors, (n,) or (n,d)."""
        self.upload_if_needed()
        n = q.shape[0]
        if logp is None:
            logp = self.ctx.empty((n,))
        if energy is None:
            energy = self.ctx.empty((n,))
        for t in (eps, var, q, p, grad, logp, energy):
            assert t.is_contiguous() and t.device == self.ctx.device
        _lib.check(self.ctx._lib.bfhip_leapfrog(self.ctx.handle, n, _ptr(eps), _ptr(var), _ptr(q), _ptr(p), _ptr(grad),
                                                _ptr(logp), _ptr(energy), _ptr(velocity)))
        return logp, energy


def polymodel_desc_from_poly(poly):
    """``PolyModel.poly_spec()`` -> a filled ``bfhip_polymodel_desc`` + the arrays it points to: dense per-output coefficients
    with the masks scattered, which is what ``_fun_and_jac`` does on every call (modules/poly.py:474-477)."""
    d, m = int(poly['input_size']), int(poly['output_size'])
    c0 = np.zeros(m)
    lin = np.zeros((m, d))
    quad = np.zeros((m, d, d))
    has_quad = False
    cubic = []
    for cf in poly['configs']:
        im = np.asarray(cf['input_mask'], dtype=np.int64)
        om = np.asarray(cf['output_mask'], dtype=np.int64)
        coef = np.asarray(cf['coef'], dtype=np.float64)
        if cf['order'] == 'linear':
            c0[om] += coef[:, 0]
            lin[np.ix_(om, im)] += coef[:, 1:]
        elif cf['order'] == 'quadratic':
            has_quad = True
            iu = np.triu_indices(im.size)
            for q, o in enumerate(om):
                quad[o, im[iu[0]], im[iu[1]]] += coef[q][iu]
        elif cf['order'] in ('cubic-2', 'cubic-3'):
            cubic.append((cf['order'], im, om, coef))
        else:
            raise ValueError('unexpected PolyConfig order "{}".'.format(cf['order']))
    # cubic configs, compact over the union of the dimensions they touch (per order)
    mask2 = np.unique(np.concatenate([c[1] for c in cubic if c[0] == 'cubic-2'] or [np.zeros(0, np.int64)])).astype(np.int32)
    mask3 = np.unique(np.concatenate([c[1] for c in cubic if c[0] == 'cubic-3'] or [np.zeros(0, np.int64)])).astype(np.int32)
    cub2 = np.zeros((m, mask2.size, mask2.size))
    cub3 = np.zeros((m, mask3.size, mask3.size, mask3.size))
    for order, im, om, coef in cubic:
        if order == 'cubic-2':
            pos = np.searchsorted(mask2, im)
            for q, o in enumerate(om):
                cub2[o][np.ix_(pos, pos)] += coef[q]
        else:
            pos = np.searchsorted(mask3, im)
            n_ = im.size
            jj, kk, ll = np.meshgrid(np.arange(n_), np.arange(n_), np.arange(n_), indexing='ij')
            sel = (jj < kk) & (kk < ll)  # only j < k < l is defined (modules/_poly.pyx:86-137)
            for q, o in enumerate(om):
                cub3[o, pos[jj[sel]], pos[kk[sel]], pos[ll[sel]]] += coef[q][sel]
    ds = _lib.PolymodelDesc()
    keep = []

    def f64(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_double))

    ds.d, ds.m = d, m
    ds.c0, ds.lin = f64(c0), f64(lin)
    if has_quad:
        ds.quad = f64(quad)
    if mask2.size:
        keep.append(mask2)
        ds.n2, ds.mask2, ds.cubic2 = int(mask2.size), mask2.ctypes.data_as(C.POINTER(C.c_int)), f64(cub2)
    if mask3.size:
        keep.append(mask3)
        ds.n3, ds.mask3, ds.cubic3 = int(mask3.size), mask3.ctypes.data_as(C.POINTER(C.c_int)), f64(cub3)
    if poly.get('use_bound', False) and (has_quad or cubic):  # (not for an all-linear model, modules/poly.py:467)
        ds.use_bound = 1
        ds.mu, ds.hess = f64(poly['mu']), f64(poly['hess'])
        ds.alpha = float(poly['alpha'])
        ds.f_mu = f64(np.asarray(poly['f_mu'], dtype=np.float64).reshape(m))
    return ds, keep


class DevicePolyModel:
    """A multi-output PolyModel resident on one GPU: ``PolyModel.fun / jac / fun_and_jac`` over batches of points
    (modules/poly.py:430-503) for linear, quadratic and cubic configs; masks are scattered to dense per-output
    coefficients here, which is what ``_fun_and_jac`` does on every call (modules/poly.py:474-477).

    poly : the dict ``PolyModel.poly_spec()`` returns."""

    def __init__(self, poly, ctx=None):
        self.ctx = ctx if ctx is not None else get_context()
        self.d, self.m = int(poly['input_size']), int(poly['output_size'])
        ds, keep = polymodel_desc_from_poly(poly)
        self._desc, self._keep = ds, keep
        self._uploaded = None

    def upload_if_needed(self):
        if self.ctx.__dict__.get('_pm_owner') is not self:
            _lib.check(self.ctx._lib.bfhip_polymodel_upload(self.ctx.handle, C.byref(self._desc)))
            self.ctx._pm_owner = self

    def fun_and_jac(self, x, jac=True):
        """x (n, d) or (d,) -> f (n, m) and, if ``jac``, the Jacobians (n, m, d); float64 device tensors."""
        torch = _torch()
        self.upload_if_needed()
        xt = self.ctx.tensor(x, torch.float64)
        single = xt.dim() == 1
        xt = xt.reshape(-1, self.d)
        n = xt.shape[0]
        f = self.ctx.empty((n, self.m))
        j = self.ctx.empty((n, self.m, self.d)) if jac else None
        _lib.check(self.ctx._lib.bfhip_polymodel_eval(self.ctx.handle, n, _ptr(xt), _ptr(f), _ptr(j)))
        if single:
            return f[0], (j[0] if jac else None)
        return f, j
