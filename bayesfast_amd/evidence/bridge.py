"""Bridge sampling estimate of the ratio of two normalising constants (reference: bayesfast/evidence/bridge.py:10-76).

With samples x_p ~ p (n_p of them, possibly as (chain, iteration)) and x_q ~ q (n_q), and the four log-density arrays
log p(x_p), log p(x_q), log q(x_p), log q(x_q), the optimal-bridge estimate log r of log(Z_p / Z_q) is the root of a score
function that is a difference of two log-sum-exp reductions over all samples (:44-49) -- ``bfhip_bridge_sums`` evaluates
it on the device for every trial value of the secant search -- and its relative error combines the variances of two
per-sample terms (:52-57, ``bfhip_bridge_terms``), the p-side one inflated by the autocorrelation time of the chains."""
import warnings

import numpy as np
from scipy.optimize import root_scalar

from ..utils.acor import integrated_time

__all__ = ['bridge']

_NAMES = ('logp_p', 'logp_q', 'logq_p', 'logq_q')


def _checked(arrays):
    """The four inputs as float64 arrays; p-side and q-side pairs must agree in shape, samples may be 1-d or (chain, step)."""
    try:
        out = [np.asarray(a, dtype=np.float64) for a in arrays]
    except Exception:
        raise ValueError('invalid value for the inputs.')
    for name, arr in ((_NAMES[3], out[3]), (_NAMES[0], out[0])):
        if arr.ndim not in (1, 2):
            raise ValueError('dim of {} should be 1 or 2, instead of {}.'.format(name, arr.ndim))
    for i, k in ((0, 2), (1, 3)):
        if out[i].shape != out[k].shape:
            raise ValueError('shape of {}, {}, is different from shape of {}, {}.'.format(_NAMES[i], out[i].shape, _NAMES[k],
                                                                                          out[k].shape))
    return out


def _relative_variance(terms, tau, n):
    return tau * np.var(terms) / np.mean(terms)**2 / n


def bridge(logp_p, logp_q, logq_p, logq_q):
    """Returns ``(logr, logr_err)``."""
    import torch
    from .. import _lib
    from ..device import get_context, _ptr
    lpp, lpq, lqp, lqq = _checked((logp_p, logp_q, logq_p, logq_q))
    n_p, n_q = lpp.size, lqq.size
    ctx = get_context()
    dev = [ctx.tensor(a.reshape(-1)) for a in (lpp, lpq, lqp, lqq)]
    shift = np.log(n_p / n_q)
    a_p = (dev[2] - dev[0] - shift).contiguous()   # log q - log p on the p samples
    b_q = (dev[1] - dev[3] + shift).contiguous()   # log p - log q on the q samples
    sums = torch.empty(2, dtype=torch.float64, device=ctx.device)

    def score(logr):
        _lib.check(ctx._lib.bfhip_bridge_sums(ctx.handle, n_p, _ptr(a_p), n_q, _ptr(b_q), float(logr), _ptr(sums)))
        on_p, on_q = sums.cpu().numpy()
        return float(on_p - on_q)

    logr = root_scalar(score, x0=0., x1=5.).root  # secant search from the reference's two starting values
    t_q = torch.empty(n_q, dtype=torch.float64, device=ctx.device)
    t_p = torch.empty(n_p, dtype=torch.float64, device=ctx.device)
    _lib.check(ctx._lib.bfhip_bridge_terms(ctx.handle, n_p, _ptr(dev[0]), _ptr(dev[2]), n_q, _ptr(dev[1]), _ptr(dev[3]), float(logr),
                                           _ptr(t_q), _ptr(t_p)))
    t_q, t_p = t_q.cpu().numpy(), t_p.cpu().numpy()
    var_q = _relative_variance(t_q, 1., n_q)
    # the p samples are correlated along their chains: the autocorrelation time is estimated twice, on the chains as
    # given and on the flattened series; the larger error is reported, and a large gap between the two is a warning sign
    errors = []
    for series in (t_p.reshape(lpp.shape), t_p):
        tau = integrated_time(series[..., np.newaxis])[0]
        errors.append(np.sqrt(_relative_variance(t_p, tau, n_p) + var_q))
    lo, hi = min(errors), max(errors)
    if (hi - lo) / lo > 0.25:
        warnings.warn('the estimated error for logr may be unreliable, since flattening before estimating tau makes the '
                      'result differ by more than 25%.', RuntimeWarning)
    if hi > 0.25:
        warnings.warn('the estimated error for logr may be unreliable, since the result is larger than 0.25.', RuntimeWarning)
    return logr, hi
