"""The drop-in seam under the reference: objects that pass the reference's own ``isinstance`` gates.

The reference has no FFI boundary on the sampler path -- its seam is Python, and it is gated on ITS classes:
``Recipe`` / ``Density`` accept a surrogate only if ``isinstance(s, bayesfast.core.module.Surrogate)``
(core/recipe.py:52-61, core/density.py:306-310), ``Recipe`` hands its own ``NTrace`` / ``HTrace`` to ``sample``
(core/recipe.py:979-981,1160-1173) and feeds the result to its own ``_get_step_size`` / ``_get_metric``, which test
``isinstance(t, (_HTrace, TraceTuple))`` (samplers/sample_trace.py:804-847).  A mirrored class with the same name does
not pass any of these.  This module therefore builds, when the reference package is importable, true SUBCLASSES of the
reference's classes (the bases are taken from the imported package at call time; nothing of the reference is copied):

* ``PolyModel``        subclass of ``bayesfast.modules.PolyModel``: identical configuration, evaluation and bound handling
                       (inherited), but ``fit`` solves the least-squares problems on the GPU (``bfhip_lstsq``).
* ``GaussianLikelihood`` subclass of ``bayesfast.core.module.Module``: the analytic module ``logp = logp0 - prec (m - y)^2 / 2``
                       of a surrogate's single output (examples/2d-donut.ipynb's second module), which the device density
                       evaluates inside the kernel (``bfhip_density_desc.link_*``).
* ``TraceTuple``       subclass of ``bayesfast.samplers.TraceTuple`` over the device-resident result: ``get``, ``samples``,
                       ``logp`` ... read the device arrays; iterating it yields real ``NTrace`` / ``HTrace`` objects (one per
                       chain, built lazily) whose ``step_size`` is a ``DualAverageAdaptation`` and whose ``metric`` is a
                       ``QuadMetric*`` carrying the adapted state, so ``_get_step_size`` / ``_get_metric`` accept them.
* ``sample``           the signature of ``bayesfast.core.sample.sample``: takes the reference's ``Density`` (read by duck
                       typing, ``adapters.py``) and the reference's ``NTrace`` / ``HTrace`` / dict / a ``TraceTuple`` returned
                       earlier (continue the chains), runs all chains in fused device launches.
* ``patch(bayesfast)`` rebinds ``sample`` where ``Recipe`` looks it up (``bayesfast.core.recipe.sample``) and at the
                       package's public names, and ``PolyModel`` at ``bayesfast.modules`` / ``bayesfast``; returns a
                       callable that undoes it.

Densities the device path does not cover (no surrogate in use, several surrogates, arbitrary Python modules downstream of
the surrogate) are refused loudly, or -- only with ``patch(..., fallback=True)`` -- left to the reference's own sampler.
There is no CPU path of this package behind any of these objects.
"""
import copy

import numpy as np

__all__ = ['reference_classes', 'PolyModel', 'GaussianLikelihood', 'GaussianPrior', 'GaussianBaseDensity', 'sample', 'patch', 'as_surrogate_density']

_CLASSES = {}


def _import_reference(bayesfast=None):
    if bayesfast is None:
        import bayesfast  # the reference package (h3jia/bayesfast), wherever the caller installed it
    return bayesfast


def _device_fit(mirror, x, y, logp, w):
    """The coefficient fit on the device (``bayesfast_amd.PolyModel.fit``: design blocks, MFMA Gram, refined solve)."""
    mirror.fit(x, y, logp, w)


def reference_classes(bayesfast=None):
    """The namespace of subclasses described in the module docstring, built once per imported reference package."""
    bf = _import_reference(bayesfast)
    if id(bf) in _CLASSES:
        return _CLASSES[id(bf)]
    from types import SimpleNamespace
    from .modules.poly import PolyConfig as _OurConfig, PolyModel as _OurModel
    from .samplers import sample_trace as _our_st
    st = bf.samplers.sample_trace
    stats_mod = bf.samplers.hmc_utils.stats
    DualAverageAdaptation = bf.samplers.hmc_utils.step_size.DualAverageAdaptation

    class PolyModel(bf.modules.PolyModel):
        __doc__ = ("``bayesfast.modules.PolyModel`` (modules/poly.py:161-589) whose ``fit`` runs on the GPU.  Everything else "
                   "-- configs, recipe, bound options, ``_set_bound``, evaluation -- is the reference's own code.")

        def _device_mirror(self):
            cfgs = [_OurConfig(c.order, np.array(c.input_mask), np.array(c.output_mask)) for c in self.configs]
            return _OurModel(cfgs, input_size=int(self._input_size), output_size=int(self._output_size),
                             bound_options=dict(use_bound=False))

        def fit(self, x, y, logp=None, w=None):
            """modules/poly.py:505-589 with the per-output ``lstsq`` (:570) replaced by the device solve; the argument
            checks are those of ``bayesfast_amd.PolyModel.fit`` (same conditions and messages as :509-526)."""
            mirror = self._device_mirror()
            _device_fit(mirror, x, y, logp, w)
            for mine, fitted in zip(self.configs, mirror.configs):
                coef = np.ascontiguousarray(fitted._coef, dtype=np.float64)
                if coef.shape != tuple(mine._A_shape):
                    raise RuntimeError('unexpected coefficient block {} for a {} config.'.format(coef.shape, mine.order))
                mine._coef = coef
            if self._use_bound and not self._all_linear:
                self._set_bound(np.asarray(x), logp)  # the reference's own host code (modules/poly.py:262-292)

    class GaussianLikelihood(bf.core.module.Module):
        __doc__ = ("like = logp0 - (m - y)^T prec (m - y) / 2 of a surrogate's output variable m, as a ``bayesfast.Module`` with "
                   "analytic fun / jac: the module downstream of a surrogate that the device density chains in the kernel "
                   "(core/density.py:552-560).  Scalar y and prec: the likelihood of a single-output surrogate "
                   "(examples/2d-donut.ipynb's second module).  y (m,) with prec (m, m), prec_diag (m,) or neither (identity): "
                   "the chi-square of a multi-output surrogate (examples/des-y1-w-cosmosis.ipynb cell 12: chi2_f / chi2_fj).")

        def __init__(self, y, prec=None, logp0=0., input_vars='__var__', output_vars='__var__', prec_diag=None, **kwargs):
            yv = np.atleast_1d(np.asarray(y, dtype=np.float64))
            if yv.ndim != 1:
                raise ValueError('y should be a scalar or a 1-d array.')
            m = yv.size
            if m == 1 and prec_diag is None and np.ndim(prec) == 0 and prec is not None:
                self._bfhip_link = dict(kind='gaussian', y=float(yv[0]), prec=float(prec), logp0=float(logp0))
                if not self._bfhip_link['prec'] > 0:
                    raise ValueError('prec should be positive.')
                lk = self._bfhip_link

                def fun(mm):
                    r = np.asarray(mm, dtype=np.float64) - lk['y']
                    return lk['logp0'] - 0.5 * (r * (lk['prec'] * r))

                def jac(mm):
                    r = np.atleast_1d(np.asarray(mm, dtype=np.float64) - lk['y'])
                    return np.diag(-(lk['prec'] * r))
            else:
                if prec is not None and prec_diag is not None:
                    raise ValueError('give me at most one of prec and prec_diag.')
                P = None if prec is None else np.ascontiguousarray(prec, dtype=np.float64).reshape(m, m)
                pd = (np.ones(m) if P is None else None) if prec_diag is None else np.ascontiguousarray(prec_diag, dtype=np.float64).reshape(m)
                self._bfhip_chi2 = dict(y=yv.copy(), prec=P, prec_diag=pd, logp0=float(logp0))
                c2 = self._bfhip_chi2

                def _r(mm):
                    dlt = np.asarray(mm, dtype=np.float64).reshape(m) - c2['y']
                    return dlt, (c2['prec'] @ dlt if c2['prec'] is not None else c2['prec_diag'] * dlt)

                def fun(mm):
                    dlt, r = _r(mm)
                    return np.atleast_1d(c2['logp0'] - 0.5 * (dlt @ r))

                def jac(mm):
                    return -_r(mm)[1][np.newaxis]

            super().__init__(fun=fun, jac=jac, input_vars=input_vars, output_vars=output_vars, **kwargs)

    class GaussianPrior(bf.core.module.Module):
        __doc__ = ("logp = like + c0 - sum_i prec[i] (x_i - mu[i])^2 / 2: the LAST module of a pipeline, reading the likelihood "
                   "variable and the density's input variable (examples/des-y1-w-cosmosis.ipynb cell 12: des_post_f / "
                   "des_post_fj), evaluated inside the kernel as the prior stage of the pipeline density.  mu, prec: (d,), "
                   "prec[i] = 0 for inputs without a prior; or indices + mu + sigma for the inputs that have one.")

        def __init__(self, input_size, mu=None, prec=None, c0=0., indices=None, sigma=None, input_vars=('like', 'x'), output_vars='logp',
                     **kwargs):
            d = int(input_size)
            if indices is not None:
                idx = np.asarray(indices, dtype=int).reshape(-1)
                mu_full, prec_full = np.zeros(d), np.zeros(d)
                mu_full[idx] = np.asarray(mu, dtype=np.float64).reshape(-1)
                prec_full[idx] = 1. / np.asarray(sigma, dtype=np.float64).reshape(-1)**2
            else:
                mu_full = np.ascontiguousarray(mu, dtype=np.float64).reshape(d)
                prec_full = np.ascontiguousarray(prec, dtype=np.float64).reshape(d)
            if not np.all(prec_full >= 0):
                raise ValueError('the prior precisions should be non-negative.')
            self._bfhip_prior = dict(mu=mu_full, prec_diag=prec_full, c0=float(c0))
            pr = self._bfhip_prior

            def fun(like, x):
                dx = np.asarray(x, dtype=np.float64) - pr['mu']
                return np.atleast_1d(like) + pr['c0'] - 0.5 * np.sum(pr['prec_diag'] * dx * dx)

            def jac(like, x):
                dx = np.asarray(x, dtype=np.float64) - pr['mu']
                return np.concatenate((np.ones((1, 1)), -(pr['prec_diag'] * dx)[np.newaxis]), axis=-1)

            super().__init__(fun=fun, jac=jac, input_vars=list(input_vars), output_vars=output_vars, **kwargs)

    class GaussianBaseDensity(bf.core.density.DensityLite):
        __doc__ = ("The base density of the tempered samplers, N(mean, cov) in the sampler's space, as a ``bayesfast.DensityLite`` "
                   "with analytic logp / grad: ``TNTrace(density_base=GaussianBaseDensity(mean, cov), logxi=...)`` passes the "
                   "reference's type check (samplers/sample_trace.py:547-555) and the device's tempered kernel reads mean and "
                   "cov from it (``bfhip_tnuts_run`` takes a quadratic base log-density).")

        def __init__(self, mean, cov):
            mean = np.atleast_1d(np.asarray(mean, dtype=np.float64))
            cov = np.atleast_2d(np.asarray(cov, dtype=np.float64))
            if cov.shape != (mean.size, mean.size):
                raise ValueError('cov should have shape (d, d).')
            prec = np.linalg.inv(cov)
            c0 = -0.5 * (mean.size * np.log(2 * np.pi) + np.linalg.slogdet(cov)[1])
            self._bfhip_gaussian = dict(mean=mean, cov=cov)

            def logp(x):
                r = np.asarray(x, dtype=np.float64) - mean
                return c0 - 0.5 * np.einsum('...i,ij,...j->...', r, prec, r)

            def grad(x):
                return -(np.asarray(x, dtype=np.float64) - mean) @ prec

            super().__init__(logp=logp, grad=grad, input_size=mean.size, vectorized=True)

    class _ChainStats:
        """Mixin: the per-chain statistics as arrays (``NStats`` / ``HStats`` keep Python lists, stats.py:39-52)."""

        def _fill(self, table, n_warmup):
            for i, k in enumerate(self.stats_items):
                col = table[:, i]
                if k in ('tree_depth', 'tree_size', 'n_int_step'):
                    col = col.astype(int)
                elif k in ('warmup', 'diverging', 'accepted'):
                    col = col.astype(bool)
                setattr(self, '_' + k, col)
            self._n_warmup_fixed = int(n_warmup)
            return self

        n_warmup = property(lambda self: self._n_warmup_fixed)

    class _NStats(_ChainStats, stats_mod.NStats):
        pass

    class _HStats(_ChainStats, stats_mod.HStats):
        pass

    class _TNStats(_ChainStats, stats_mod.TNStats):
        pass

    class TraceTuple(st.TraceTuple):
        __doc__ = ("``bayesfast.samplers.TraceTuple`` (samplers/sample_trace.py:631-801) over a device-resident result "
                   "(``bayesfast_amd.TraceTuple``); see the module docstring.")

        def __init__(self, inner, template):
            # (the base constructor wants finished per-chain traces; they are built on demand instead)
            self._inner = inner
            self._template = template
            self._sampler = inner.sampler
            self._chain_traces = None

        # ---- the reference's readers, answered from the device arrays ----
        n_chain = property(lambda self: self._inner.n_chain)
        i_iter = property(lambda self: self._inner.i_iter)
        input_size = property(lambda self: self._inner.input_size)
        finished = property(lambda self: self._inner.finished)
        samples = property(lambda self: self._inner.samples)
        samples_original = property(lambda self: self._inner.samples_original)
        logp = property(lambda self: self._inner.logp)
        logp_original = property(lambda self: self._inner.logp_original)
        n_call = property(lambda self: self._inner.n_call)

        @property
        def n_iter(self):
            return self._inner.n_iter

        @n_iter.setter
        def n_iter(self, n):
            n = int(n)
            if n < self.i_iter or n < self.n_warmup:
                raise ValueError('invalid value for n_iter.')
            self._inner._trace.n_iter = n
            self._template._n_iter = n

        @property
        def n_warmup(self):
            return self._inner.n_warmup

        def get(self, since_iter=None, include_warmup=False, original_space=True, return_type='samples', flatten=True):
            return self._inner.get(since_iter, include_warmup, original_space, return_type, flatten)

        __call__ = get

        # ---- per-chain traces of the reference's own classes, for _get_step_size / _get_metric / users ----
        @property
        def _sample_traces(self):
            if self._chain_traces is None:
                self._chain_traces = tuple(self._chain_trace(i) for i in range(self.n_chain))
            return self._chain_traces

        @property
        def stats(self):
            return [t.stats for t in self._sample_traces]

        def _chain_trace(self, i):
            inner = self._inner
            inner.gather()  # (host arrays of all chains; a plain copy without a process group)
            adapted = inner._adapted_state()
            t = copy.copy(self._template)
            # (a template that already carries adaptation / metric INSTANCES: _set_*_2 leave them alone, so each chain
            # gets its own copy -- otherwise every chain's adapted state would land in the caller's one object)
            if isinstance(getattr(t, '_step_size', None), DualAverageAdaptation):
                t._step_size = copy.copy(t._step_size)
            if hasattr(getattr(t, '_metric', None), 'velocity'):
                t._metric = copy.copy(t._metric)
            t._chain_id = i
            t._x_0 = np.array(adapted['x_0'][i])
            t._x_0_transformed = True
            t._set_step_size_2()   # DualAverageAdaptation from the trace's options (samplers/sample_trace.py:365-373) ...
            t._set_metric_2()      # ... and the QuadMetric* (:418-455); their adapted state is filled in below
            t._chain_initialized = True
            ss = t._step_size
            if isinstance(ss, DualAverageAdaptation):
                ss._log_step = float(adapted['log_step'][i])
                ss._log_bar = float(adapted['log_bar'][i])
                ss._hbar = float(adapted['hbar'][i])
                ss._count = int(adapted['count'][i])
            m = t._metric
            if hasattr(m, '_cov'):   # QuadMetricFull(Adapt), samplers/hmc_utils/metrics.py:103-111
                import scipy.linalg
                m._cov = np.array(adapted['cov'][i])
                m._chol = scipy.linalg.cholesky(m._cov, lower=True)
            elif hasattr(m, '_var'):   # QuadMetricDiag(Adapt), :60-71: random() reads _inv_std, velocity() reads _var
                m._var = np.array(adapted['var'][i])
                m._std = m._var**0.5
                m._inv_std = 1. / m._std
            t._samples = inner._samples[i]
            t._samples_original = inner._samples_original[i]
            t._logp_original = inner._logp_original[i]
            if self._sampler == 'TNUTS':   # TNStats: (u, weight) in front of the NUTS columns (samplers/hmc_utils/stats.py:22-24)
                t._stats = _TNStats()._fill(np.concatenate([inner._array('stats_t')[i], inner._stats[i]], axis=1), inner.n_warmup)
            else:
                cls = _HStats if self._sampler == 'HMC' else _NStats
                t._stats = cls()._fill(inner._stats[i], inner.n_warmup)
            return t

    ns = SimpleNamespace(bayesfast=bf, PolyModel=PolyModel, GaussianLikelihood=GaussianLikelihood, GaussianPrior=GaussianPrior,
                         GaussianBaseDensity=GaussianBaseDensity, TraceTuple=TraceTuple,
                         _our_st=_our_st)
    _CLASSES[id(bf)] = ns
    return ns


def PolyModel(*args, **kwargs):
    """``reference_classes().PolyModel(...)``: a reference ``PolyModel`` (it IS one) that fits on the GPU."""
    return reference_classes().PolyModel(*args, **kwargs)


def GaussianLikelihood(*args, **kwargs):
    """``reference_classes().GaussianLikelihood(...)``."""
    return reference_classes().GaussianLikelihood(*args, **kwargs)


def GaussianBaseDensity(*args, **kwargs):
    """``reference_classes().GaussianBaseDensity(...)``."""
    return reference_classes().GaussianBaseDensity(*args, **kwargs)


def GaussianPrior(*args, **kwargs):
    """``reference_classes().GaussianPrior(...)``."""
    return reference_classes().GaussianPrior(*args, **kwargs)


def as_surrogate_density(density):
    """A reference ``Density`` with its surrogate in use -> the ``bayesfast_amd.SurrogateDensity`` the device runs.

    Covered pipelines (anything else raises ``NotImplementedError``):
      * one ``PolyModel`` surrogate with ``output_size == 1`` whose scope spans every module, i.e. the surrogate's output
        is the density variable;
      * the same surrogate spanning all modules but the last, the last being a ``GaussianLikelihood`` of its output.
    """
    from .core.density import SurrogateDensity, GaussianLink
    from .adapters import surrogate_density_from_reference
    if isinstance(density, SurrogateDensity):
        return density
    if not hasattr(density, '_surrogate_list') or not hasattr(density, '_module_list'):
        raise NotImplementedError('the device sampler runs Density objects with a surrogate (DensityLite has none).')
    sl, ml = list(density._surrogate_list), list(density._module_list)
    if getattr(density, '_use_surrogate', True) is False or len(sl) == 0:
        raise NotImplementedError('the device sampler runs the SURROGATE density; this Density is not using one.')
    if len(sl) != 1:
        raise NotImplementedError('the device sampler takes exactly one PolyModel surrogate.')
    su = sl[0]
    i_step, n_step = int(su._scope[0]) % max(len(ml), 1), int(su._scope[1])
    rest = ml[n_step:] if i_step == 0 else None
    if rest is not None and len(rest) >= 1 and hasattr(rest[0], '_bfhip_chi2') and len(rest) <= 2:
        # [surrogate of the first modules (any output_size), Gaussian likelihood of its outputs(, Gaussian prior)]: the pipeline
        # density (bfhip_pipeline_upload; examples/des-y1-w-cosmosis.ipynb cells 12-18)
        return _pipeline_density_from_reference(density, su, rest)
    if int(su._output_size) != 1:
        raise NotImplementedError('a surrogate with output_size > 1 needs a bayesfast_amd.integrate.GaussianLikelihood of its '
                                  'outputs (and optionally a GaussianPrior) as the modules behind it.')
    link = None
    if i_step == 0 and n_step == len(ml):
        pass
    elif i_step == 0 and n_step == len(ml) - 1 and hasattr(ml[-1], '_bfhip_link'):
        last = ml[-1]
        if list(last.input_vars) != list(su.output_vars):
            raise NotImplementedError('the Gaussian likelihood should read the surrogate\'s output variable.')
        link = GaussianLink(last._bfhip_link['y'], last._bfhip_link['prec'], last._bfhip_link['logp0'])
    else:
        raise NotImplementedError('the modules downstream of the surrogate are arbitrary Python; only a '
                                  'bayesfast_amd.integrate.GaussianLikelihood of its output runs inside the kernel.')
    out = surrogate_density_from_reference(density)
    out.link = link
    out._device = None
    return out


def _pipeline_density_from_reference(density, su, rest):
    """The reference ``Density`` [.., surrogate ``su``, GaussianLikelihood(, GaussianPrior)] -> ``Chi2PipelineDensity`` with the
    reference object's surrogate coefficients, bound, transforms and decay state (attributes read by duck typing)."""
    from .core.density import Chi2PipelineDensity
    from .adapters import polymodel_from_reference
    like = rest[0]
    if list(like.input_vars) != list(su.output_vars):
        raise NotImplementedError('the Gaussian likelihood should read the surrogate\'s output variable.')
    c2 = like._bfhip_chi2
    if c2['y'].size != int(su._output_size):
        raise ValueError('the likelihood\'s data vector and the surrogate\'s output_size disagree.')
    prior = None
    if len(rest) == 2:
        post = rest[1]
        if not hasattr(post, '_bfhip_prior'):
            raise NotImplementedError('the module behind the likelihood is arbitrary Python; only a '
                                      'bayesfast_amd.integrate.GaussianPrior runs inside the kernel.')
        if list(post.input_vars) != list(like.output_vars) + list(density.input_vars):
            raise NotImplementedError('the prior module should read [the likelihood variable, the density\'s input variable].')
        prior = post._bfhip_prior
    d = int(density.input_size)
    if list(su.input_vars) != list(density.input_vars) or int(su._input_size) != d:
        raise NotImplementedError('the surrogate should read the density\'s whole input.')
    hb = density._hard_bounds
    hb = bool(hb) if isinstance(hb, (bool, np.bool_)) else np.array(hb)
    out = Chi2PipelineDensity(polymodel_from_reference(su), c2['y'], prec=c2['prec'], prec_diag=c2['prec_diag'], logp0=c2['logp0'],
                              prior_mu=None if prior is None else prior['mu'], prior_prec=None if prior is None else prior['prec_diag'],
                              prior_c0=0. if prior is None else prior['c0'],
                              input_scales=None if density._input_scales is None else np.array(density._input_scales, dtype=np.float64),
                              hard_bounds=hb if density._input_scales is not None else False,
                              decay_options=dict(use_decay=bool(density._use_decay)))
    if density._use_decay:
        out._mu = np.array(density._mu, dtype=np.float64)
        out._hess = np.array(density._hess, dtype=np.float64)
        out._alpha_2 = float(density._alpha_2)
        out._alpha = float(density._alpha_2)**0.5
        out._gamma = float(density._gamma)
    out._device = None
    return out


def _our_trace(ns, ref_trace, sampler):
    """The reference's trace object -> this package's option object of the same meaning (fields read by duck typing:
    samplers/sample_trace.py:159-172,460-512)."""
    st = ns._our_st
    bf = ns.bayesfast
    from bayesfast.samplers.hmc_utils.step_size import DualAverageAdaptation
    mt = bf.samplers.hmc_utils.metrics
    kw = dict(n_chain=ref_trace._n_chain, n_iter=ref_trace._n_iter, n_warmup=ref_trace._n_warmup, x_0=ref_trace._x_0,
              max_change=ref_trace._max_change)
    # step size: a number with the adaptation options, or a DualAverageAdaptation INSTANCE (then the option fields do not
    # exist on the trace, samplers/sample_trace.py:318-322); a fresh instance is read back into the same options
    ss = ref_trace._step_size
    if isinstance(ss, DualAverageAdaptation):
        if ss._count != 1 or ss._hbar != 0. or ss._log_bar != ss._log_step:
            raise NotImplementedError('a DualAverageAdaptation that has already adapted cannot seed the device chains; '
                                      'continue from the TraceTuple of the device sampler instead.')
        d = np.asarray(ref_trace._x_0).shape[-1] if ref_trace._x_0 is not None else ref_trace.input_size
        kw.update(step_size=float(np.exp(ss._log_step)) * d**0.25, adapt_step_size=bool(ss._adapt), target_accept=float(ss._target),
                  gamma=float(ss._gamma), k=float(ss._k), t_0=float(ss._t_0))
    else:
        kw.update(step_size=ss, adapt_step_size=ref_trace._adapt_step_size, target_accept=ref_trace._target_accept,
                  gamma=ref_trace._gamma, k=ref_trace._k, t_0=ref_trace._t_0)
    # metric: 'diag' / 'full' / an array with the adaptation options, or a QuadMetric INSTANCE (samplers/sample_trace.py:377-379)
    m = ref_trace._metric
    if isinstance(m, mt.QuadMetric):
        adapting = isinstance(m, (mt.QuadMetricDiagAdapt, mt.QuadMetricFullAdapt))
        if adapting and (m._n_samples != 0 or m._previous_update != 0):
            raise NotImplementedError('a metric that has already adapted cannot seed the device chains; continue from the '
                                      'TraceTuple of the device sampler instead.')
        kw.update(metric=np.array(m._cov if hasattr(m, '_cov') else m._var), adapt_metric=adapting)
        if adapting:
            fg = m._foreground_cov if hasattr(m, '_foreground_cov') else m._foreground_var
            kw.update(initial_mean=np.array(fg.mean), initial_weight=float(fg.n_samples), adapt_window=int(m._adapt_window),
                      update_window=int(m._update_window), doubling=bool(m._doubling))
    else:
        kw.update(metric=m, adapt_metric=ref_trace._adapt_metric, initial_mean=ref_trace._initial_mean,
                  initial_weight=ref_trace._initial_weight, adapt_window=ref_trace._adapt_window,
                  update_window=ref_trace._update_window, doubling=ref_trace._doubling)
    # one integer from the trace's generator seeds the per-chain xoshiro streams (the reference spawns one PCG64 per chain
    # from the same generator, samplers/sample_trace.py:192-193)
    kw['random_generator'] = int(ref_trace.random_generator.integers(0, 2**63 - 1))
    if sampler == 'TNUTS':
        gb = ref_trace.density_base._bfhip_gaussian
        t = st.TNTrace(st.GaussianBase(gb['mean'], gb['cov']), logxi=ref_trace.logxi, max_treedepth=ref_trace._max_treedepth, **kw)
    elif sampler == 'NUTS':
        t = st.NTrace(max_treedepth=ref_trace._max_treedepth, **kw)
    else:
        t = st.HTrace(n_int_step=ref_trace._n_int_step, **kw)
    t._x_0_transformed = bool(ref_trace._x_0_transformed)
    return t


def sample(density, sample_trace=None, sampler='NUTS', n_run=None, parallel_backend=None, verbose=True):
    """``bayesfast.core.sample.sample`` (core/sample.py:26-220) on the GPU: same arguments, same meaning;
    ``parallel_backend`` is accepted and unused (chains shard over ``torch.distributed`` ranks instead).  Returns a
    ``bayesfast.samplers.TraceTuple`` (a subclass instance, see the module docstring)."""
    from .core.sample import sample as _sample
    ns = reference_classes()
    bf = ns.bayesfast
    st = bf.samplers.sample_trace
    den = as_surrogate_density(density)
    if isinstance(sample_trace, ns.TraceTuple):       # continue the chains (core/sample.py:94-96)
        inner = _sample(den, sample_trace._inner, n_run=n_run, verbose=verbose)
        return ns.TraceTuple(inner, sample_trace._template)
    if isinstance(sample_trace, st.TraceTuple):
        raise ValueError('this TraceTuple was not produced by the device sampler and cannot be continued by it.')
    if sample_trace is None or isinstance(sample_trace, dict):  # core/sample.py:80-92
        kw = {} if sample_trace is None else sample_trace
        if sampler == 'NUTS':
            sample_trace = st.NTrace(**kw)
        elif sampler == 'HMC':
            sample_trace = st.HTrace(**kw)
        elif sampler == 'TNUTS':   # core/sample.py:83-84
            sample_trace = st.TNTrace(**kw)
        elif sampler == 'THMC':
            raise NotImplementedError('THMC: the reference\'s own THTrace constructor raises (samplers/sample_trace.py:600).')
        elif sampler == 'Ensemble':
            raise NotImplementedError
        else:
            raise ValueError('unexpected value for sampler.')
    if isinstance(sample_trace, st.THTrace):
        raise NotImplementedError('the drop-in seam covers NUTS, HMC and TNUTS.')
    elif isinstance(sample_trace, st.TNTrace):
        sampler = 'TNUTS'
        if not hasattr(sample_trace.density_base, '_bfhip_gaussian'):
            raise NotImplementedError('the device\'s tempered sampler takes a Gaussian base density: '
                                      'bayesfast_amd.integrate.GaussianBaseDensity(mean, cov).')
    elif isinstance(sample_trace, st.NTrace):
        sampler = 'NUTS'
    elif isinstance(sample_trace, st.HTrace):
        sampler = 'HMC'
    else:
        raise ValueError('unexpected value for sample_trace.')
    if sample_trace.x_0 is None:  # core/sample.py:106-113: Sobol-normal starts, in the sampler's space
        dim = den.input_size
        sample_trace._x_0 = bf.utils.sobol.multivariate_normal(np.zeros(dim), np.eye(dim), sample_trace.n_chain)
        sample_trace._x_0_transformed = True
    elif not sample_trace.x_0_transformed:  # :114-116
        sample_trace._x_0 = density.from_original(sample_trace._x_0)
        sample_trace._x_0_transformed = True
    inner = _sample(den, _our_trace(ns, sample_trace, sampler), n_run=n_run, verbose=verbose)
    return ns.TraceTuple(inner, sample_trace)


def patch(bayesfast=None, fallback=False):
    """Rebind the reference's names to the device path: ``bayesfast.core.recipe.sample`` (what ``Recipe`` calls),
    ``bayesfast.core.sample.sample``, ``bayesfast.core.sample`` / ``bayesfast.sample`` where they name the function, and
    ``PolyModel`` at ``bayesfast.modules.poly`` / ``bayesfast.modules`` / ``bayesfast.core.recipe`` / ``bayesfast``.
    ``fallback=True``: densities the device path does not cover go to the reference's own sampler instead of raising.
    Returns ``unpatch()``."""
    ns = reference_classes(bayesfast)
    bf = ns.bayesfast
    import sys
    sample_module = sys.modules[bf.__name__ + '.core.sample']
    recipe_module = sys.modules[bf.__name__ + '.core.recipe']
    original_sample = sample_module.sample

    def _sample(density, sample_trace=None, sampler='NUTS', n_run=None, parallel_backend=None, verbose=True):
        try:
            return sample(density, sample_trace, sampler, n_run, parallel_backend, verbose)
        except NotImplementedError:
            if not fallback:
                raise
            return original_sample(density, sample_trace, sampler, n_run, parallel_backend, verbose)

    _sample.__doc__ = sample.__doc__
    saved = []

    def rebind(obj, name, value):
        if hasattr(obj, name):
            saved.append((obj, name, getattr(obj, name)))
            setattr(obj, name, value)

    rebind(sample_module, 'sample', _sample)
    rebind(recipe_module, 'sample', _sample)
    for pkg in (bf.core, bf):
        if getattr(pkg, 'sample', None) is original_sample:
            rebind(pkg, 'sample', _sample)
    ref_poly = sys.modules[bf.__name__ + '.modules.poly'].PolyModel
    for obj in (sys.modules[bf.__name__ + '.modules.poly'], bf.modules, recipe_module, bf):
        if getattr(obj, 'PolyModel', None) is ref_poly:
            rebind(obj, 'PolyModel', ns.PolyModel)

    def unpatch():
        while saved:
            obj, name, value = saved.pop()
            setattr(obj, name, value)

    return unpatch
