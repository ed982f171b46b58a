"""Surrogate density: the part of ``bayesfast.core.density.Density`` that the sampler path touches, for a
pipeline whose log density is one PolyModel output, or a Gaussian likelihood of one PolyModel output.

Host state is NumPy; every ``logp``/``grad``/``logp_and_grad`` call is one batched device launch
(``bfhip_logp_grad``), replacing the per-row Python recursion of core/density.py:523-525."""
import numpy as np

from ..modules.poly import PolyModel

__all__ = ['SurrogateDensity', 'Chi2PipelineDensity', 'GaussianLink']


class GaussianLink:
    """The analytic module downstream of a single-output surrogate: logp = logp0 - prec (m - y)^2 / 2 of the surrogate's
    output m.  ``SurrogateDensity(surrogate, link=GaussianLink(y, prec))`` is the reference's
    ``Density(module_list=[model, like], surrogate_list=[PolyModel(scope=(0, 1))])`` with such a ``like`` module
    (core/density.py:527-560; examples/2d-donut.ipynb: m = |x|, logp = -(m - 5)^2 / 0.5 is GaussianLink(5., 4.)).
    The fused sampler evaluates it in the kernel (``bfhip_density_desc.link_*``)."""

    def __init__(self, y, prec, logp0=0.):
        self.y, self.prec, self.logp0 = float(y), float(prec), float(logp0)
        if not self.prec > 0:
            raise ValueError('prec should be positive.')

    def spec(self):
        return dict(kind='gaussian', y=self.y, prec=self.prec, logp0=self.logp0)

    def fun(self, m):
        r = np.asarray(m, dtype=np.float64) - self.y
        return self.logp0 - 0.5 * (r * (self.prec * r))

    def jac(self, m):
        return -(self.prec * (np.asarray(m, dtype=np.float64) - self.y))


class SurrogateDensity:
    """log density = PolyModel surrogate (+ decay penalty) in a constrained parameter space.

    surrogate : PolyModel with output_size 1
    input_scales : None or (d, 2) array, ``Density(input_scales=...)`` (core/density.py:33-58)
    hard_bounds : bool or (d,) / (d, 2) array_like (core/density.py:60-76)
    decay_options : dict for ``set_decay_options`` (core/density.py:761-794)
    link : None (the surrogate's output is the log density) or a ``GaussianLink`` (the surrogate replaces the first module
        of a two-module pipeline; the link is the second)
    """

    _multi_output = False   # (Chi2PipelineDensity: the surrogate's m outputs feed a likelihood module)

    def __init__(self, surrogate, input_scales=None, hard_bounds=False, decay_options=None, link=None):
        if not isinstance(surrogate, PolyModel) or (surrogate.output_size != 1 and not self._multi_output):
            raise ValueError('surrogate should be a PolyModel with output_size 1.')
        if link is not None and not isinstance(link, GaussianLink):
            raise ValueError('link should be a GaussianLink or None.')
        self.link = link
        self.surrogate = surrogate
        self._d = surrogate.input_size
        if input_scales is None:
            self._input_scales = None
        else:
            sc = np.ascontiguousarray(input_scales, dtype=np.float64)
            if sc.ndim == 1:
                sc = np.array((np.zeros_like(sc), sc)).T.copy()
            if sc.shape != (self._d, 2) or not np.all(sc[:, 1] > sc[:, 0]):
                raise ValueError('invalid value for input_scales.')
            self._input_scales = sc
        try:
            hb = np.atleast_1d(hard_bounds).astype(bool).astype(np.uint8)
            if hb.ndim == 1:
                hb = np.array((hb, hb)).T.copy()
            if hb.shape[0] == 1:
                hb = np.repeat(hb, self._d, 0)
            assert hb.shape == (self._d, 2)
        except Exception:
            raise ValueError('Invalid value for hard_bounds')
        self._hard_bounds = np.ascontiguousarray(hb)
        self.set_decay_options(**(decay_options or {}))
        self._mu = self._hess = None
        self._device = None

    input_size = property(lambda self: self._d)

    def set_decay_options(self, use_decay=False, alpha=None, alpha_p=150., gamma=0.1):
        """core/density.py:761-794."""
        self._use_decay = bool(use_decay)
        self._alpha = None if alpha is None else float(alpha)
        self._alpha_2 = None if alpha is None else float(alpha)**2
        if alpha_p is None and alpha is None:
            raise ValueError('alpha and alpha_p cannot both be None.')
        self._alpha_p = None if alpha_p is None else float(alpha_p)
        gamma = float(gamma)
        if not gamma > 0:
            raise ValueError('invalid value for gamma.')
        self._gamma = gamma
        self._device = None

    # ---- constraint transforms (core/density.py:142-163 -> transforms/_constraint.pyx), on device ----
    def _transform_device(self):
        """A device density that carries only the transform (usable before the surrogate is fitted)."""
        from ..device import DeviceDensity
        if getattr(self, '_tdev', None) is None:
            d = self._d
            spec = dict(d=d, ranges=self._input_scales,
                        hard_bounds=self._hard_bounds if self._input_scales is not None else None, su_lo=None,
                        su_diff=None, use_decay=False,
                        poly=dict(input_size=d, output_size=1, use_bound=False,
                                  configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.array([0]),
                                                coef=np.zeros((1, d + 1)))]))
            self._tdev = DeviceDensity(spec)
        return self._tdev

    def _constraint(self, which, x):
        if self._input_scales is None:  # core/density.py:100-107: no scales -> identity, unit Jacobian, zero second derivative
            x = np.array(x, dtype=np.float64)
            return x if which in ('from_original', 'to_original') else (np.ones_like(x) if which.endswith('grad') else np.zeros_like(x))
        return self._transform_device().constraint(which, np.asarray(x, dtype=np.float64)).cpu().numpy()

    def from_original(self, x):
        return self._constraint('from_original', x)

    def from_original_grad(self, x):
        return self._constraint('from_original_grad', x)

    def from_original_grad2(self, x):
        return self._constraint('from_original_grad2', x)

    def to_original(self, x):
        return self._constraint('to_original', x)

    def to_original_grad(self, x):
        return self._constraint('to_original_grad', x)

    def to_original_grad2(self, x):
        return self._constraint('to_original_grad2', x)

    # device-tensor variants used by ``sample`` (no host round trip; identity without input_scales, as the
    # reference's ``_constraint`` returns x / ones / zeros then, core/density.py:100-107)
    def to_original_device(self, x_dev):
        if self._input_scales is None:
            return x_dev
        return self._transform_device().constraint('to_original', x_dev)

    def to_original_density_device(self, density_dev, x_dev):
        """core/density.py:188-195 on device tensors: density - sum(log|d x_orig / d x_trans|)."""
        if self._input_scales is None:
            return density_dev
        import torch
        g = self._transform_device().constraint('to_original_grad', x_dev)
        return density_dev - torch.sum(torch.log(torch.abs(g)), dim=-1)

    def to_original_density(self, density, x_trans):
        """core/density.py:188-195: density in the original space from the transformed-space value."""
        diff = np.sum(np.log(np.abs(self.to_original_grad(x_trans))), axis=-1)
        return np.asarray(density) - diff

    # ---- fit ----
    input_vars = ('__var__',)   # Density(input_vars=...), core/density.py:256
    density_name = '__var__'    # Density(density_name=...), core/density.py:640

    output_var = None           # with a link: the name of the surrogate's output variable in the var_dicts

    def fit(self, x, logp=None, y=None):
        """``Density.fit`` (core/density.py:813-830): either ``fit(var_dicts)`` as in the reference -- a sequence of
        VariableDict-like objects whose ``_fun`` (or ``fun``) maps variable names to arrays; the points are the
        concatenated ``input_vars`` and the true log-densities ``_fun[density_name][0]`` (:833-838) -- or
        ``fit(x, logp)`` with x (n, d) original-space points and logp (n,).  With a ``link`` the surrogate is fitted to the
        true values y (n,) of the module it replaces (:826-830, ``su.fit(x, y, logp)``), not to logp."""
        if logp is None:
            vds = list(x)
            if not vds or not all(hasattr(v, '_fun') or hasattr(v, 'fun') for v in vds):
                raise ValueError('var_dicts should consist of VariableDict(s).')
            get = lambda v: v._fun if hasattr(v, '_fun') else v.fun
            x = np.array([np.concatenate([np.atleast_1d(get(v)[n]) for n in self.input_vars]) for v in vds])
            logp = np.array([np.atleast_1d(get(v)[self.density_name])[0] for v in vds])
            if self.link is not None:
                if self.output_var is None:
                    raise ValueError('set output_var (the name of the surrogate\'s output variable) to fit from var_dicts.')
                y = np.array([np.atleast_1d(get(v)[self.output_var])[0] for v in vds])
        x = np.ascontiguousarray(x, dtype=np.float64)
        logp = np.asarray(logp, dtype=np.float64).reshape(-1)
        if x.ndim != 2 or x.shape != (logp.size, self._d):
            raise ValueError('x should have shape (n, d) and logp shape (n,).')
        if self._use_decay:
            self._set_decay(x)
        su = self.surrogate
        xs = x if su._input_scales is None else (x - su._input_scales[:, 0]) / su._input_scales_diff
        if self.link is None:
            su.fit(xs, logp[:, None], logp)
        else:
            if y is None:
                raise ValueError('a density with a link is fitted to the outputs y of the module the surrogate replaces.')
            y = np.asarray(y, dtype=np.float64).reshape(-1)
            if y.size != logp.size:
                raise ValueError('y should have shape (n,).')
            su.fit(xs, y[:, None], logp)
        self._device = None

    def _set_decay(self, x):
        """core/density.py:796-811."""
        from ..utils.threads import blas_single_thread
        with blas_single_thread():  # (utils/threads.py)
            self._mu = np.mean(x, axis=0)
            self._hess = np.linalg.inv(np.cov(x, rowvar=False))
            if self._alpha_p is not None:
                dx = x - self._mu
                beta = np.sum((dx @ self._hess) * dx, axis=1)**0.5  # (the three-operand einsum as one matrix product)
        if self._alpha_p is not None:
            self._alpha = float(np.percentile(beta, self._alpha_p) if self._alpha_p < 100 else
                                np.max(beta) * self._alpha_p / 100)
            self._alpha_2 = self._alpha**2

    # ---- device ----
    def spec(self):
        su = self.surrogate
        spec = dict(d=self._d, ranges=self._input_scales,
                    hard_bounds=self._hard_bounds if self._input_scales is not None else None,
                    su_lo=None if su._input_scales is None else su._input_scales[:, 0],
                    su_diff=None if su._input_scales is None else su._input_scales_diff,
                    poly=su.poly_spec(), use_decay=self._use_decay, link=None if self.link is None else self.link.spec())
        if self._use_decay:
            if self._mu is None:
                raise RuntimeError('the decay statistics have not been set; call fit first.')
            spec.update(decay_mu=self._mu, decay_hess=self._hess, decay_alpha2=self._alpha_2, decay_gamma=self._gamma)
        return spec

    def device(self, ctx=None):
        from ..device import DeviceDensity
        if self._device is None or (ctx is not None and self._device.ctx is not ctx):
            self._device = DeviceDensity(self.spec(), ctx)
        return self._device

    def logp_and_grad(self, x, original_space=True):
        """``Density.logp_and_grad(x, original_space, use_surrogate=True)`` (core/density.py:724-754)."""
        x = np.asarray(x, dtype=np.float64)
        lp, g = self.device().logp_and_grad(x, original_space)
        return lp.cpu().numpy(), g.cpu().numpy()

    def logp(self, x, original_space=True):
        return self.logp_and_grad(x, original_space)[0]

    __call__ = logp

    def grad(self, x, original_space=True):
        return self.logp_and_grad(x, original_space)[1]


class Chi2PipelineDensity(SurrogateDensity):
    """A pipeline evaluated AND sampled on the device: a multi-output ``PolyModel`` surrogate (x -> m outputs), a Gaussian
    likelihood of those outputs and, optionally, a diagonal Gaussian prior of the inputs, i.e. the reference's
    ``Density(module_list=[model, chi2(, post)], surrogate_list=[PolyModel])`` with ``use_surrogate=True``
    (core/density.py:487-566: the surrogate replaces the module in its scope, :527-551; the Jacobians are chained,
    ``jac = np.dot(J_out, J_in)``, :552-560; ``logp`` and ``grad`` are read from the density variable, :737-739;
    examples/des-y1-w-cosmosis.ipynb cells 12-18 is this with d = 27, m = 457).

        like = logp0 - (f(x) - y)^T prec (f(x) - y) / 2
        logp = like + prior_c0 - sum_i prior_prec[i] (x_i - prior_mu[i])^2 / 2

    surrogate : PolyModel with output_size m (any config orders, masks, bound, input_scales)
    y : (m,) data vector;  prec : (m, m) precision matrix, or  prec_diag : (m,) inverse variances
    logp0 : additive constant of the log-likelihood
    prior_mu, prior_prec : (d,) each or None -- prior_prec[i] = 0 leaves input i without a prior; prior_c0 its constant
    input_scales, hard_bounds, decay_options : as ``SurrogateDensity``

    ``sample(density, ...)`` runs NUTS / HMC on it inside the fused kernel (``bfhip_pipeline_upload``: the (m, d) Jacobian is
    never formed; two FP64-MFMA contractions per gradient, bayesfast_amd/csrc/bfhip_pld.h).  ``logp_and_grad_device`` keeps
    the first implementation -- ``bfhip_polymodel_eval`` (f and the (n, m, d) Jacobians) + ``bfhip_chi2_stage`` -- as an
    independent second route the tests compare with."""

    _multi_output = True

    def __init__(self, surrogate, y, prec=None, prec_diag=None, logp0=0., prior_mu=None, prior_prec=None, prior_c0=0.,
                 input_scales=None, hard_bounds=False, decay_options=None):
        super().__init__(surrogate, input_scales=input_scales, hard_bounds=hard_bounds, decay_options=decay_options)
        m = surrogate.output_size
        self._y = np.ascontiguousarray(y, dtype=np.float64).reshape(m)
        if (prec is None) == (prec_diag is None):
            raise ValueError('give me exactly one of prec and prec_diag.')
        self._prec = None if prec is None else np.ascontiguousarray(prec, dtype=np.float64).reshape(m, m)
        self._pdiag = None if prec_diag is None else np.ascontiguousarray(prec_diag, dtype=np.float64).reshape(m)
        self._logp0 = float(logp0)
        if (prior_mu is None) != (prior_prec is None):
            raise ValueError('prior_mu and prior_prec go together.')
        self._prior_mu = None if prior_mu is None else np.ascontiguousarray(prior_mu, dtype=np.float64).reshape(self._d)
        self._prior_prec = None if prior_prec is None else np.ascontiguousarray(prior_prec, dtype=np.float64).reshape(self._d)
        if self._prior_prec is not None and not np.all(self._prior_prec >= 0):
            raise ValueError('prior_prec should be non-negative.')
        self._prior_c0 = float(prior_c0)

    def fit(self, x, logp, y=None):
        """``Density.fit`` (core/density.py:813-830) for this pipeline: x (n, d) original-space points, y (n, m) the true
        outputs of the module the surrogate replaces, logp (n,) the true log-densities (the bound's ``center_max`` and the
        decay statistics use them)."""
        if y is None:
            raise ValueError('the surrogate of a pipeline density is fitted to the outputs y (n, m) of the module it replaces.')
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        logp = np.asarray(logp, dtype=np.float64).reshape(-1)
        if x.ndim != 2 or x.shape != (logp.size, self._d) or y.shape != (logp.size, self.surrogate.output_size):
            raise ValueError('x should have shape (n, d), y shape (n, m) and logp shape (n,).')
        if self._use_decay:
            self._set_decay(x)
        su = self.surrogate
        xs = x if su._input_scales is None else (x - su._input_scales[:, 0]) / su._input_scales_diff
        su.fit(xs, y, logp)
        self._device = None

    def spec(self):
        sp = super().spec()
        sp['link'] = None
        sp['chi2'] = dict(y=self._y, prec=self._prec, prec_diag=self._pdiag, logp0=self._logp0)
        sp['prior'] = None if self._prior_mu is None else dict(mu=self._prior_mu, prec_diag=self._prior_prec, c0=self._prior_c0)
        return sp

    def logp_and_grad_device(self, x, grad=True):
        """Second route (original-space x, no transforms / prior / decay): x (n, d) array or device tensor -> logp (n,),
        grad (n, d) device tensors through ``bfhip_polymodel_eval`` + ``bfhip_chi2_stage``."""
        import torch
        from .. import _lib
        from ..device import _ptr
        if self._input_scales is not None or self.surrogate._input_scales is not None or self._prior_mu is not None or self._use_decay:
            raise NotImplementedError('the two-kernel route covers the plain [surrogate, chi-square] pipeline only.')
        dm = self.surrogate.device_model()
        ctx = dm.ctx
        xt = ctx.tensor(x, torch.float64).reshape(-1, self.input_size)
        f, j = dm.fun_and_jac(xt, jac=grad)
        n, m, d = xt.shape[0], self.surrogate.output_size, self.input_size
        lp = ctx.empty((n,))
        g = ctx.empty((n, d)) if grad else None
        y = ctx.tensor(self._y)
        pr = None if self._prec is None else ctx.tensor(self._prec)
        pd = None if self._pdiag is None else ctx.tensor(self._pdiag)
        _lib.check(ctx._lib.bfhip_chi2_stage(ctx.handle, n, m, d, _ptr(f), _ptr(j) if grad else None, _ptr(y), _ptr(pr), _ptr(pd),
                                             self._logp0, _ptr(lp), _ptr(g)))
        return lp, g

    def logp_and_grad(self, x, original_space=True):
        x = np.asarray(x, dtype=np.float64)
        lp, g = self.device().logp_and_grad(x.reshape(-1, self.input_size), original_space)
        lp, g = lp.cpu().numpy(), g.cpu().numpy()
        return (lp[0], g[0]) if x.ndim == 1 else (lp, g)

    def logp(self, x, original_space=True):
        return self.logp_and_grad(x, original_space)[0]

    __call__ = logp

    def grad(self, x, original_space=True):
        return self.logp_and_grad(x, original_space)[1]
