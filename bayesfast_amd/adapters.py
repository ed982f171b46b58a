"""Adapters from the reference's objects to this package's device descriptions.

``density_spec_from_reference(density)`` reads a fitted ``bayesfast.Density`` whose surrogate list holds ONE
``bayesfast.modules.PolyModel`` with ``output_size == 1`` (the log-density surrogate of the sampler path) and returns the
plain spec dict that ``bayesfast_amd.device.DeviceDensity`` / ``density_desc_from_spec`` turn into a
``bfhip_density_desc`` (include/bfhip.h).  Only attributes are read (duck typing): the reference package is not imported.

Fields read, with the reference lines that define them:
  Density   ._input_scales (core/density.py:33-58), ._hard_bounds (:60-76), ._use_decay / ._mu / ._hess / ._alpha_2 /
            ._gamma (:761-811), ._surrogate_list (:306-310), .input_size
  PolyModel .configs[i].order / .input_mask / .output_mask / ._coef (modules/poly.py:24-158), ._input_size /
            ._output_size, ._use_bound / ._all_linear / ._mu / ._hess / ._alpha / ._f_mu (:232-292),
            ._input_scales / ._input_scales_diff (core/module.py:76-83)
"""
import numpy as np

__all__ = ['poly_spec_from_reference', 'density_spec_from_reference', 'device_density_from_reference',
           'polymodel_from_reference', 'surrogate_density_from_reference']


def poly_spec_from_reference(pm):
    """``bayesfast.modules.PolyModel`` -> the ``poly`` part of a density spec (or the argument of ``DevicePolyModel``)."""
    cfgs = [dict(order=c.order, input_mask=np.array(c.input_mask), output_mask=np.array(c.output_mask), coef=np.array(c._coef))
            for c in pm.configs]
    poly = dict(input_size=int(pm._input_size), output_size=int(pm._output_size), configs=cfgs,
                use_bound=bool(pm._use_bound and not pm._all_linear))
    if poly['use_bound']:
        poly.update(mu=np.array(pm._mu), hess=np.array(pm._hess), alpha=float(pm._alpha), f_mu=np.array(pm._f_mu))
    return poly


def density_spec_from_reference(den):
    """``bayesfast.Density`` (fitted, one PolyModel surrogate with output_size 1) -> density spec dict."""
    surrogates = list(den._surrogate_list)
    if len(surrogates) != 1:
        raise ValueError('the device density takes exactly one surrogate (the log-density PolyModel).')
    su = surrogates[0]
    d = int(den.input_size)
    spec = dict(d=d, ranges=None, hard_bounds=None, su_lo=None, su_diff=None, poly=poly_spec_from_reference(su),
                use_decay=bool(den._use_decay), link=None)  # (a density with a likelihood module: integrate.as_surrogate_density)
    if den._input_scales is not None:
        spec['ranges'] = np.array(den._input_scales, dtype=np.float64)
        hb = den._hard_bounds
        if isinstance(hb, (bool, np.bool_)):
            hb = hb * np.ones((d, 2), np.uint8)
        spec['hard_bounds'] = np.array(hb, dtype=np.uint8)
    if su._input_scales is not None:
        spec['su_lo'] = np.array(su._input_scales[:, 0], dtype=np.float64)
        spec['su_diff'] = np.array(su._input_scales_diff, dtype=np.float64)
    if den._use_decay:
        spec.update(decay_mu=np.array(den._mu), decay_hess=np.array(den._hess), decay_alpha2=float(den._alpha_2),
                    decay_gamma=float(den._gamma))
    return spec


def device_density_from_reference(den, ctx=None):
    """The reference's fitted Density, resident on the GPU: ``DeviceDensity.logp_and_grad`` then answers
    ``den.logp_and_grad(x, original_space=..., use_surrogate=True)`` for batches of points, and
    ``bayesfast_amd.chains.DeviceChains`` samples it."""
    from .device import DeviceDensity
    return DeviceDensity(density_spec_from_reference(den), ctx)


def polymodel_from_reference(pm):
    """``bayesfast.modules.PolyModel`` -> ``bayesfast_amd.PolyModel`` with the same configs, coefficients, bound and input
    scales (attributes copied; nothing is re-fitted)."""
    from .modules.poly import PolyConfig, PolyModel
    configs = [PolyConfig(c.order, np.array(c.input_mask), np.array(c.output_mask)) for c in pm.configs]
    use_bound = bool(pm._use_bound)
    out = PolyModel(configs, input_size=int(pm._input_size), output_size=int(pm._output_size),
                    bound_options=dict(use_bound=use_bound),
                    input_scales=None if pm._input_scales is None else np.array(pm._input_scales, dtype=np.float64))
    for c, r in zip(out.configs, pm.configs):
        c._coef = np.array(r._coef, dtype=np.float64)
    if use_bound and not pm._all_linear:
        out._mu = np.array(pm._mu, dtype=np.float64)
        out._hess = np.array(pm._hess, dtype=np.float64)
        out._alpha = float(pm._alpha)
        out._f_mu = np.array(pm._f_mu, dtype=np.float64)
    for k in ('_alpha_p', '_center_max'):
        if hasattr(pm, k) and hasattr(out, k):
            setattr(out, k, getattr(pm, k))
    out._dev_model_key = None
    return out


def surrogate_density_from_reference(den):
    """``bayesfast.Density`` (one PolyModel surrogate with output_size 1) -> ``bayesfast_amd.SurrogateDensity``: what
    ``bayesfast_amd.sample`` runs on, with the reference object's transforms, bound and decay state.  ``sample()`` calls this
    for any density object that is not a ``SurrogateDensity`` but has a ``_surrogate_list``."""
    from .core.density import SurrogateDensity
    surrogates = list(den._surrogate_list)
    if len(surrogates) != 1:
        raise ValueError('the device density takes exactly one surrogate (the log-density PolyModel).')
    d = int(den.input_size)
    hb = den._hard_bounds
    if isinstance(hb, (bool, np.bool_)):
        hb = bool(hb)
    else:
        hb = np.array(hb)
    out = SurrogateDensity(polymodel_from_reference(surrogates[0]),
                           input_scales=None if den._input_scales is None else np.array(den._input_scales, dtype=np.float64),
                           hard_bounds=hb if den._input_scales is not None else False,
                           decay_options=dict(use_decay=bool(den._use_decay)))
    if den._use_decay:
        out._mu = np.array(den._mu, dtype=np.float64)
        out._hess = np.array(den._hess, dtype=np.float64)
        out._alpha_2 = float(den._alpha_2)
        out._alpha = float(den._alpha_2)**0.5
        out._gamma = float(den._gamma)
    assert out._d == d
    out._device = None
    return out
