"""PolyConfig / PolyModel with the reference's interface (bayesfast/modules/poly.py), device-backed.

State is host NumPy (so ``copy.deepcopy`` and pickling keep working, core/recipe.py:822,1163); evaluation
and the least-squares fit run in libbfhip.so.  What differs from the reference, on purpose:

* ``fit`` solves the normal equations (FP64-MFMA Gram + equilibrated Cholesky) and refines the solution twice on the
  true residual (``bfhip_lstsq``: the accuracy of LAPACK gelsd up to a condition number of 1e7, tested against
  fixtures of the reference), and factorises once for all outputs that share a set of configs (the reference
  rebuilds and re-factorises A for every output, modules/poly.py:529).  A numerically rank-deficient design is
  reported (RuntimeWarning) and ridge-regularised; gelsd would return the minimum-norm solution.
* ``fun``/``jac``/``fun_and_jac`` run on the device: single-output surrogates (the log-density surrogate of the sampler
  path) through ``bfhip_logp_grad``, multi-output modules through one ``bfhip_polymodel_eval`` launch.
"""
import ctypes as C
import warnings
from collections import namedtuple

import numpy as np

from .. import _lib
from ..core.module import Surrogate

__all__ = ['PolyConfig', 'PolyModel']

BoundOptions = namedtuple('BoundOptions', ('use_bound', 'alpha', 'alpha_p', 'center_max'))
_ORDERS = ('linear', 'quadratic', 'cubic-2', 'cubic-3')


class PolyConfig:
    """Configuring the PolyModel (modules/poly.py:19-158).

    order : 'linear', 'quadratic', 'cubic-2' or 'cubic-3'
    input_mask, output_mask : None (all variables) or 1-d array_like of int (sorted, unique)"""

    def __init__(self, order, input_mask=None, output_mask=None):
        if order not in _ORDERS:
            raise ValueError('order should be one of ("linear", "quadratic", "cubic-2", "cubic-3"), instead of '
                             '"{}".'.format(order))
        self._order = order
        self._set_input_mask(input_mask)
        self._set_output_mask(output_mask)
        self._coef = None

    order = property(lambda self: self._order)
    input_mask = property(lambda self: self._input_mask)
    output_mask = property(lambda self: self._output_mask)

    @staticmethod
    def _mask(m):
        if m is None:
            return None
        m = np.sort(np.unique(np.asarray(m, dtype=int)))
        m.flags.writeable = False
        return m

    def _set_input_mask(self, im):
        self._input_mask = self._mask(im)

    def _set_output_mask(self, om):
        self._output_mask = self._mask(om)

    @property
    def input_size(self):
        return self._input_mask.size if self._input_mask is not None else None

    @property
    def output_size(self):
        return self._output_mask.size if self._output_mask is not None else None

    def _check_masks(self):
        if self._input_mask is None or self._output_mask is None:
            raise RuntimeError('you have not defined self.input_mask and/or self.output_mask yet.')

    @property
    def _A_shape(self):
        """Shape of the dense coefficient block (modules/poly.py:87-107)."""
        self._check_masks()
        n, m = self.input_size, self.output_size
        return {'linear': (m, n + 1), 'quadratic': (m, n, n), 'cubic-2': (m, n, n), 'cubic-3': (m, n, n, n)}[self._order]

    @property
    def _a_shape(self):
        """Number of independent coefficients per output (modules/poly.py:109-129)."""
        self._check_masks()
        n = self.input_size
        return ({'linear': n + 1, 'quadratic': n * (n + 1) // 2, 'cubic-2': n * n,
                 'cubic-3': n * (n - 1) * (n - 2) // 6}[self._order],)

    def _set(self, a, i):
        """Scatter the packed solution of output i into the dense block (modules/poly.py:131-158 ->
        modules/_poly.pyx:183-214).  Entries the kernels never read are zero here (the reference leaves them
        uninitialised)."""
        self._check_masks()
        a = np.ascontiguousarray(a, dtype=np.float64)
        i = int(i)
        if a.shape != self._a_shape:
            raise ValueError('shape of a {} does not match the expected shape {}.'.format(a.shape, self._a_shape))
        if not 0 <= i < self.output_size:
            raise ValueError('i = {} out of range for self.output_size = {}.'.format(i, self.output_size))
        n = self.input_size
        if self._order == 'linear':
            coefi = a
        elif self._order == 'quadratic':
            coefi = np.zeros((n, n))
            coefi[np.triu_indices(n)] = a
        elif self._order == 'cubic-2':
            coefi = a.reshape(n, n).copy()
        else:
            coefi = np.zeros((n, n, n))
            j, k, l = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing='ij')
            coefi[(j < k) & (k < l)] = a
        if self._coef is None:
            self._coef = np.zeros(self._A_shape)
        self._coef[i] = coefi


class PolyModel(Surrogate):
    """Polynomial surrogate model, up to cubic order (modules/poly.py:161-219).

    configs : str, PolyConfig, or 1-d array_like of them; 'quadratic' means ['linear', 'quadratic'] etc.
    bound_options : dict for ``set_bound_options``
    remaining arguments go to ``Surrogate`` (input_size, output_size, scope, input_vars, output_vars,
    input_scales, fit_options)."""

    def __init__(self, configs, bound_options=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if isinstance(configs, str):
            if configs not in _ORDERS:
                raise ValueError('if configs is a str, it should be "linear", "quadratic", "cubic-2" or "cubic-3".')
            configs = list(_ORDERS[:_ORDERS.index(configs) + 1])
        if isinstance(configs, PolyConfig):
            configs = [configs]
        if not hasattr(configs, '__iter__'):
            raise ValueError('invalid value for configs.')
        self._configs = []
        for i, conf in enumerate(configs):
            if isinstance(conf, str):
                conf = PolyConfig(conf)
            if not isinstance(conf, PolyConfig):
                raise ValueError('invalid value for the #{} element of configs.'.format(i))
            if conf._input_mask is None:
                conf._set_input_mask(np.arange(self._input_size))
            if conf._output_mask is None:
                conf._set_output_mask(np.arange(self._output_size))
            self._configs.append(conf)
        self._configs = tuple(self._configs)
        self._build_recipe()
        if bound_options is None:
            bound_options = {}
        if not isinstance(bound_options, dict):
            raise ValueError('bound_options should be a dict.')
        self.set_bound_options(**bound_options)
        self._mu = self._hess = self._f_mu = None

    configs = property(lambda self: self._configs)
    n_config = property(lambda self: len(self._configs))
    recipe = property(lambda self: self._recipe)

    @property
    def bound_options(self):
        return BoundOptions(self._use_bound, self._alpha, self._alpha_p, self._center_max)

    def set_bound_options(self, use_bound=True, alpha=None, alpha_p=100., center_max=True):
        """Linear extrapolation options far away from the fit points (modules/poly.py:232-260)."""
        self._use_bound = bool(use_bound)
        if alpha is None:
            self._alpha = None
        else:
            try:
                alpha = float(alpha)
                assert alpha > 0
            except Exception:
                raise ValueError('invalid value for alpha.')
            self._alpha = alpha
        if alpha_p is None:
            if alpha is None:
                raise ValueError('alpha and alpha_p cannot both be None.')
            self._alpha_p = None
        else:
            try:
                alpha_p = float(alpha_p)
                assert alpha_p > 0
            except Exception:
                raise ValueError('invalid value for alpha_p.')
            self._alpha_p = alpha_p
        self._center_max = bool(center_max)

    def _build_recipe(self):
        """(output, order) -> config index table with the reference's overlap checks (modules/poly.py:298-338)."""
        rr = np.full((self._output_size, 4), -1)
        for ii, conf in enumerate(self._configs):
            col = _ORDERS.index(conf.order)
            if np.any(rr[conf._output_mask, col] >= 0):
                raise ValueError('multiple {} PolyConfig(s) share at least one common output variable. Please check '
                                 'your PolyConfig #{}.'.format(conf.order.replace('-', '_'), ii))
            rr[conf._output_mask, col] = ii
        if np.any(np.all(rr < 0, axis=1)):
            raise ValueError('no PolyConfig has output for variable(s) {}.'.format(
                np.argwhere(np.all(rr < 0, axis=1)).flatten()))
        self._recipe = rr
        self._recipe.flags.writeable = False

    @property
    def n_param(self):
        return int(np.sum([conf._a_shape[0] for conf in self._configs]))

    @property
    def _all_linear(self):
        return all(conf.order == 'linear' for conf in self._configs)

    # ---- spec for the device / the test oracle ----
    def poly_spec(self, use_bound=None):
        for conf in self._configs:
            if conf._coef is None:
                raise RuntimeError('the PolyModel has not been fitted yet.')
        ub = (self._use_bound and not self._all_linear) if use_bound is None else bool(use_bound)
        poly = dict(input_size=self._input_size, output_size=self._output_size, use_bound=ub,
                    configs=[dict(order=c.order, input_mask=np.array(c._input_mask), output_mask=np.array(c._output_mask),
                                  coef=np.array(c._coef)) for c in self._configs])
        if ub:
            if self._mu is None:
                raise RuntimeError('the bound has not been set yet.')
            poly.update(mu=self._mu, hess=self._hess, alpha=self._alpha, f_mu=self._f_mu)
        return poly

    def _output_spec(self, ii, use_bound=None):
        """Density-style spec (d, poly with output_size 1) of output ii alone."""
        poly = self.poly_spec(use_bound)
        cfgs = []
        for c in poly['configs']:
            pos = np.argwhere(c['output_mask'] == ii)
            if pos.size:
                q = int(pos[0, 0])
                cfgs.append(dict(order=c['order'], input_mask=c['input_mask'], output_mask=np.array([0]),
                                 coef=c['coef'][q:q + 1]))
        one = dict(poly, output_size=1, configs=cfgs)
        if one['use_bound']:
            one['f_mu'] = np.asarray(poly['f_mu']).reshape(-1)[ii:ii + 1]
            if all(c['order'] == 'linear' for c in cfgs):
                one['use_bound'] = False
        return dict(d=self._input_size, ranges=None, hard_bounds=None, su_lo=None, su_diff=None, poly=one,
                    use_decay=False)

    def device_model(self, use_bound=None):
        """The multi-output module on the GPU (``DevicePolyModel``): batched ``fun_and_jac`` in one launch.  Rebuilt
        after every fit."""
        from ..device import DevicePolyModel
        key = (id(self._configs[0]._coef), use_bound)
        if getattr(self, '_dev_model_key', None) != key:
            self._dev_model = DevicePolyModel(self.poly_spec(use_bound))
            self._dev_model_key = key
        return self._dev_model

    def fun_and_jac_batch(self, x, jac=True):
        """f (n, m) and Jacobians (n, m, d) of a batch of points x (n, d) as device tensors (one launch)."""
        return self.device_model().fun_and_jac(x, jac=jac)

    def _device_eval(self, x, use_bound=None):
        """f (m,), j (m, d) of one point, on device."""
        from ..device import DeviceDensity
        x = np.asarray(x, dtype=np.float64).reshape(1, -1)
        f, j = self.device_model(use_bound).fun_and_jac(x)  # one launch for all outputs and config orders
        return f[0].cpu().numpy(), j[0].cpu().numpy()

    def _fun(self, x):
        return self._device_eval(x)[0]

    def _jac(self, x):
        return self._device_eval(x)[1]

    def _fun_and_jac(self, x):
        return self._device_eval(x)

    # ---- fit ----
    _N_REFINE = 2  # refinement steps of the least-squares solve

    def fit(self, x, y, logp=None, w=None):
        """Fit the polynomial model (modules/poly.py:505-589) on device.

        x (n, input_size), y (n, output_size); logp (n,) is only used for ``center_max``; w (n,) row weights."""
        import torch
        from ..device import get_context, _ptr
        x = np.asarray(x)
        y = np.asarray(y)
        if not (x.ndim == 2 and x.shape[-1] == self._input_size):
            raise ValueError('x should be a 2-d array, with shape (# of points, # of input_size), instead of '
                             '{}.'.format(x.shape))
        if not (y.ndim == 2 and y.shape[-1] == self._output_size):
            raise ValueError('y should be a 2-d array, with shape (# of points, # of output_size), instead of '
                             '{}.'.format(y.shape))
        if not x.shape[0] == y.shape[0]:
            raise ValueError('x and y have different # of points.')
        if x.shape[0] < self.n_param:
            raise ValueError('I need at least {} points, but you only gave me {}.'.format(self.n_param, x.shape[0]))
        if w is not None:
            w = np.atleast_1d(w)
            if not (w.ndim == 1 and w.shape[0] == x.shape[0]):
                raise ValueError('invalid shape for w.')
        ctx = get_context()
        lib, h = ctx._lib, ctx.handle
        n = x.shape[0]
        xt = ctx.tensor(x, torch.float64)
        wt = None if w is None else ctx.tensor(w, torch.float64)
        # outputs that use the same set of configs share one design matrix and ONE factorisation
        groups = {}
        for ii in range(self._output_size):
            groups.setdefault(tuple(int(v) for v in self._recipe[ii]), []).append(ii)
        pending = []
        for key, outs in groups.items():
            confs = [self._configs[k] for k in key if k >= 0]
            widths = [c._a_shape[0] for c in confs]
            P = int(sum(widths))
            A = ctx.empty((n, P))
            col = 0
            for c, wd in zip(confs, widths):
                xin = xt[:, torch.as_tensor(np.array(c._input_mask), device=ctx.device)].contiguous()
                _lib.check(lib.bfhip_design_block(h, _ORDERS.index(c.order), n, c.input_size, _ptr(xin), _ptr(wt), _ptr(A), P, col))
                col += wd
            B = ctx.tensor(y[:, outs], torch.float64)
            if wt is not None:
                B = (B * wt[:, None]).contiguous()  # b *= w, modules/poly.py:567
            G = ctx.empty((P, P))
            r = ctx.empty((P, len(outs)))
            info = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
            # normal equations + two refinement steps on the true residual: the accuracy of an orthogonal factorisation
            # for every design the pivot threshold lets through (include/bfhip.h: bfhip_lstsq)
            work = ctx.empty((n * len(outs) + P * len(outs),))
            _lib.check(lib.bfhip_lstsq(h, n, P, len(outs), _ptr(A), P, _ptr(B), _ptr(G), _ptr(r), self._N_REFINE, _ptr(work),
                                       _ptr(info)))
            pending.append((outs, confs, widths, P, A, B, G, r, info, work))
        # the bound's statistics are host work on x alone: they run while the device solves
        bound = self._bound_stats(x, logp) if (self._use_bound and not self._all_linear) else None
        for outs, confs, widths, P, A, B, G, r, info, work in pending:
            if int(info.item()) != 0:
                # numerically rank-deficient design matrix: LAPACK gelsd (modules/poly.py:570) returns the MINIMUM-NORM solution
                # x = A^+ b, and so does this path (round 6; a ridge-regularised solve before, another member of the solution set):
                # with G = A^T A = V diag(lam) V^T, x = V_k diag(1 / lam_k) V_k^T A^T b over the eigenvalues above the cut, then
                # two refinement steps on the true residual inside that subspace (the accuracy the Gram matrix alone loses).
                # The cut: gelsd drops singular values below eps sigma_max; the eigenvalues of G are known to ~P eps lam_max, so
                # directions with sigma < 3e-6 sigma_max count as null here -- the same coefficients whenever the deficiency is
                # exact (a duplicated or constant input: what occurs in practice, and what the reference's fixture has).
                # The eigen-decomposition is hipSOLVER's through torch.linalg.eigh on the device-resident G: a rare fallback.
                warnings.warn('the design matrix of the polynomial fit is numerically rank deficient (pivot {}); returning the '
                              'minimum-norm solution of the truncated normal equations.'.format(int(info.item())), RuntimeWarning)
                _lib.check(lib.bfhip_gram(h, n, P, len(outs), _ptr(A), P, _ptr(B), _ptr(G), _ptr(r)))
                torch.cuda.current_stream(ctx.device).wait_stream(ctx.stream)
                lam, V = torch.linalg.eigh(G, UPLO='U')
                if not bool(torch.isfinite(lam).all()) or float(lam[-1]) <= 0.:
                    raise np.linalg.LinAlgError('the normal equations of the polynomial fit are singular.')
                keep = lam > 1e-11 * lam[-1]
                Vk, ilam = V[:, keep], 1. / lam[keep]
                sol_t = Vk @ (ilam[:, None] * (Vk.T @ r))
                for _ in range(2):
                    res = B - A @ sol_t
                    sol_t = sol_t + Vk @ (ilam[:, None] * (Vk.T @ (A.T @ res)))
                r.copy_(sol_t)
                ctx.stream.wait_stream(torch.cuda.current_stream(ctx.device))
            sol = r.cpu().numpy()
            for jo, ii in enumerate(outs):
                k = 0
                for c, wd in zip(confs, widths):
                    qq = int(np.argwhere(c._output_mask == ii)[0, 0])
                    c._set(sol[k:k + wd, jo], qq)
                    k += wd
        del pending
        self._dev_model_key = None  # the device copy of the module is stale now
        if bound is not None:
            self._apply_bound(*bound)
        self._dev_model_key = None

    def _bound_stats(self, x, logp=None):
        """The part of ``_set_bound`` that only needs the fit points: mu, H = inv(cov), alpha and the point f_mu is taken at
        (modules/poly.py:262-276)."""
        try:
            x = np.ascontiguousarray(x, dtype=np.float64)
            assert x.shape[-1] == self._input_size and x.ndim == 2
        except Exception:
            raise ValueError('invalid value for x.')
        from ..utils.threads import blas_single_thread
        with blas_single_thread():  # (spinning BLAS workers starve the GPU runtime's threads: utils/threads.py)
            mu = np.mean(x, axis=0)
            hess = np.linalg.inv(np.cov(x, rowvar=False))
            if self._alpha_p is not None:
                dx = x - mu  # (the reference's three-operand einsum, modules/poly.py:277, as one matrix product: 11 ms -> 1 ms)
                _beta = np.sum((dx @ hess) * dx, axis=1)**0.5
        alpha = None
        if self._alpha_p is not None:
            if self._alpha_p < 100.:
                alpha = float(np.percentile(_beta, self._alpha_p))
            else:
                alpha = float(np.max(_beta) * self._alpha_p / 100.)
        mu_f = mu
        if self._center_max:
            try:
                logp = np.asarray(logp)
                assert x.shape[0] == logp.shape[0] and logp.ndim == 1
                mu_f = x[np.argmax(logp)]
            except Exception:
                warnings.warn('invalid value for logp. Disabling center_max for now.', RuntimeWarning)
        return mu, hess, alpha, mu_f

    def _apply_bound(self, mu, hess, alpha, mu_f):
        self._mu, self._hess = mu, hess
        if alpha is not None:
            self._alpha = alpha
        self._f_mu = self._device_eval(mu_f, use_bound=False)[0]  # (modules/poly.py:277-292: the fitted model at mu_f)

    def _set_bound(self, x, logp=None):
        """mu, H = inv(cov), alpha and f_mu of the extrapolation bound (modules/poly.py:262-292)."""
        self._apply_bound(*self._bound_stats(x, logp))
