"""Bridge sampling estimate of a normalising constant (bayesfast/evidence/bridge.py:10-76).

logr solves score(logr) = 0 (secant iteration from 0 and 5, as ``scipy.optimize.root_scalar(x0=0., x1=5.)``); every
evaluation of the score is two log-sum-exp reductions over all p and q samples, ``bfhip_bridge_sums``; the per-sample
terms of the error estimate come from ``bfhip_bridge_terms``.  The autocorrelation time of those terms is host NumPy
(an FFT per chain, bayesfast_amd/utils/acor.py)."""
import warnings

import numpy as np
from scipy.optimize import root_scalar

from ..utils.acor import integrated_time

__all__ = ['bridge']


def bridge(logp_p, logp_q, logq_p, logq_q):
    import torch
    from .. import _lib
    from ..device import get_context, _ptr
    try:
        lpp, lpq = np.asarray(logp_p, dtype=np.float64), np.asarray(logp_q, dtype=np.float64)
        lqp, lqq = np.asarray(logq_p, dtype=np.float64), np.asarray(logq_q, dtype=np.float64)
    except Exception:
        raise ValueError('invalid value for the inputs.')
    if lqq.ndim not in (1, 2):
        raise ValueError('dim of logq_q should be 1 or 2, instead of {}.'.format(lqq.ndim))
    if lpp.ndim not in (1, 2):
        raise ValueError('dim of logp_p should be 1 or 2, instead of {}.'.format(lpp.ndim))
    if lpp.shape != lqp.shape:
        raise ValueError('shape of logp_p, {}, is different from shape of logq_p, {}.'.format(lpp.shape, lqp.shape))
    if lpq.shape != lqq.shape:
        raise ValueError('shape of logp_q, {}, is different from shape of logq_q, {}.'.format(lpq.shape, lqq.shape))
    n_p, n_q = lpp.size, lqq.size
    ctx = get_context()
    d_lpp, d_lpq = ctx.tensor(lpp.reshape(-1)), ctx.tensor(lpq.reshape(-1))
    d_lqp, d_lqq = ctx.tensor(lqp.reshape(-1)), ctx.tensor(lqq.reshape(-1))
    a = (d_lqp - d_lpp - np.log(n_p / n_q)).contiguous()
    b = (d_lpq - d_lqq + np.log(n_p / n_q)).contiguous()
    out2 = torch.empty(2, dtype=torch.float64, device=ctx.device)

    def score(logr):
        _lib.check(ctx._lib.bfhip_bridge_sums(ctx.handle, n_p, _ptr(a), n_q, _ptr(b), float(logr), _ptr(out2)))
        c, dd = out2.cpu().numpy()
        return float(c - dd)

    logr = root_scalar(score, x0=0., x1=5.).root
    f1 = torch.empty(n_q, dtype=torch.float64, device=ctx.device)
    f2 = torch.empty(n_p, dtype=torch.float64, device=ctx.device)
    _lib.check(ctx._lib.bfhip_bridge_terms(ctx.handle, n_p, _ptr(d_lpp), _ptr(d_lqp), n_q, _ptr(d_lpq), _ptr(d_lqq), float(logr),
                                           _ptr(f1), _ptr(f2)))
    f1, f2 = f1.cpu().numpy(), f2.cpu().numpy()
    re2_q = np.var(f1) / np.mean(f1)**2 / n_q
    tau_uf = integrated_time(f2.reshape(lpp.shape)[..., np.newaxis])[0]
    re2_p_uf = tau_uf * np.var(f2) / np.mean(f2)**2 / n_p
    err_uf = (re2_p_uf + re2_q)**0.5
    tau_f = integrated_time(f2[..., np.newaxis])[0]
    re2_p_f = tau_f * np.var(f2) / np.mean(f2)**2 / n_p
    err_f = (re2_p_f + re2_q)**0.5
    diff_err = abs(err_f - err_uf) / min(err_f, err_uf)
    logr_err = max(err_f, err_uf)
    if diff_err > 0.25:
        warnings.warn('the estimated error for logr may be unreliable, since flattening before estimating tau makes the '
                      'result differ by more than 25%.', RuntimeWarning)
    if logr_err > 0.25:
        warnings.warn('the estimated error for logr may be unreliable, since the result is larger than 0.25.', RuntimeWarning)
    return logr, logr_err
