"""Gaussianized bridge sampling (reference: bayesfast/evidence/gaussianized.py:179-216): the evidence of a posterior from
its samples.  The first half of the samples trains a ``SIT`` density q, the second half and fresh draws from q feed the
bridge estimator, and since q is normalised the bridge's log r is the log-evidence."""
import warnings

import numpy as np

from ..transforms.sit import SIT
from .bridge import bridge

__all__ = ['GBS']


def _optional_positive(value, kind, name):
    if value is None:
        return None
    try:
        value = kind(value)
    except Exception:
        value = None
    if value is None or not value > 0:
        raise ValueError('invalid value for {}.'.format(name))
    return value


def _evaluate(logp, points):
    """logp on an array of points (..., d): in one call when the callable takes a batch, point by point otherwise."""
    lead, d = points.shape[:-1], points.shape[-1]
    rows = points.reshape(-1, d)
    values = None
    try:
        values = np.asarray(logp(rows), dtype=np.float64)
        if values.shape != (rows.shape[0],):
            values = None
    except Exception:
        values = None
    if values is None:
        values = np.array([logp(row) for row in rows], dtype=np.float64)
    return values.reshape(lead)


class GBS:
    """``GBS(sit=None, parallel_backend=None, n_q=None, f_call=0.05)`` with the reference's meaning: ``sit`` a ``SIT``, the
    keyword arguments of one, or None; ``n_q`` draws from the fitted SIT, or ``f_call`` times the number of density calls
    the sampler spent when the samples come as a ``TraceTuple``.  ``parallel_backend`` is accepted and ignored."""

    def __init__(self, sit=None, parallel_backend=None, n_q=None, f_call=0.05):
        if isinstance(sit, SIT):
            self.sit = sit
        elif sit is None or isinstance(sit, dict):
            self.sit = SIT(**(sit or {}))
        else:
            raise ValueError('invalid value for sit.')
        self.n_q = _optional_positive(n_q, int, 'n_q')
        self.f_call = _optional_positive(f_call, float, 'f_call')

    def _draws_from_q(self, n_samples, n_call):
        if self.n_q is not None:
            return self.n_q
        if self.f_call is not None:
            if n_call is not None:
                return int(n_call * self.f_call)
            warnings.warn('f_call should be used only when x_p is a TraceTuple. Using equal-sample allocation for now.',
                          RuntimeWarning)
        return n_samples

    def run(self, x_p, logp, logp_p=None):
        """x_p: posterior samples (n, d), (chain, iteration, d) or a ``TraceTuple``; logp: the unnormalised log-posterior;
        logp_p: its values on x_p if already known.  Returns ``(logz, logz_err)``."""
        from ..utils.threads import blas_single_thread
        if not callable(logp):
            raise ValueError('logp should be callable.')
        # (host BLAS on one thread, as the reference runs it, core/sample.py:167: the d x d factorisations of a SIT fit gain nothing
        # from threads, and OpenBLAS workers spinning after each call exhaust a container's CPU quota -- every thread of the process,
        # the ROCm runtime's included, then stands still until the next 100 ms scheduling period: nine such stalls were 0.5 s of a
        # config-5 GBS run)
        with blas_single_thread():
            return self._run(x_p, logp, logp_p)

    def _run(self, x_p, logp, logp_p):
        from ..samplers.sample_trace import TraceTuple
        n_call = None
        if isinstance(x_p, TraceTuple):
            dev = self._run_on_device(x_p, logp, logp_p)
            if dev is not None:
                return dev
            n_call, x_p = x_p.n_call, x_p.get(flatten=False)
        else:
            try:
                x_p = np.asarray(x_p, dtype=np.float64)
            except Exception:
                x_p = None
            if x_p is None or x_p.ndim not in (2, 3):
                raise ValueError('invalid value for x_p.')
        n_samples = int(np.prod(x_p.shape[:-1]))
        if x_p.shape[-1] < 2 or n_samples < 2:
            raise ValueError('invalid shape for x_p.')
        n_q = self._draws_from_q(n_samples, n_call)
        if x_p.shape[0] == 1:   # a single chain: drop the chain axis, the halves are then halves of the iterations
            x_p = x_p[0]
        cut = x_p.shape[0] // 2
        train, test = x_p[:cut], x_p[cut:]
        self.sit.fit(data=train)
        x_q = self.sit.sample(n_q)[0]
        known = None
        if logp_p is not None:
            known = np.asarray(logp_p)
            if known.shape == x_p.shape[:-1]:
                known = known[cut:]
            else:
                warnings.warn('the logp_p you gave me seems not correct. Will recompute it from logp and x_p.', RuntimeWarning)
                known = None
        if known is None:
            known = _evaluate(logp, test)
        return bridge(known, _evaluate(logp, x_q), self.sit.logq(test), self.sit.logq(x_q))

    def _run_on_device(self, trace, logp, logp_p):
        """The same estimate with the samples where ``sample()`` left them: a TraceTuple of one rank whose arrays are device tensors,
        a SIT with the default generator and ``logp`` the ``logp`` method of a ``SurrogateDensity`` (whose kernel takes device
        tensors).  The halves, the SIT's draws and the four log-density vectors stay on the GPU; only the vectors (n,) visit the
        host, for ``bridge``.  None when any of that does not hold (the host path runs)."""
        import torch
        from ..core.density import SurrogateDensity
        from ..utils import sobol
        from .. import parallel
        if type(logp) is SurrogateDensity:
            den = logp
        else:
            den = getattr(logp, '__self__', None)
            if not (isinstance(den, SurrogateDensity) and getattr(logp, '__func__', None) in (SurrogateDensity.logp, SurrogateDensity.__call__)):
                return None
        if self.sit.mvn_generator is not sobol.multivariate_normal or parallel.world()[1] > 1:
            return None
        t = trace.device('samples_original')
        if not isinstance(t, torch.Tensor) or t.dim() != 3:
            return None
        x_p = t[:, trace.n_warmup:]                       # TraceTuple.get(flatten=False): (chain, iteration, d) after the warm-up
        if trace.n_warmup >= trace.i_iter - 1:
            raise ValueError('since_iter is too large. Nothing to return.')
        n_samples = int(x_p.shape[0] * x_p.shape[1])
        if x_p.shape[-1] < 2 or n_samples < 2:
            raise ValueError('invalid shape for x_p.')
        n_q = self._draws_from_q(n_samples, trace.n_call)
        if x_p.shape[0] == 1:
            x_p = x_p[0]
        cut = x_p.shape[0] // 2
        d = x_p.shape[-1]
        train, test = x_p[:cut].reshape(-1, d), x_p[cut:].reshape(-1, d).contiguous()
        self.sit.fit(data=train)
        x_q = self.sit._sample_device(n_q)
        dd = den.device()
        known = None
        if logp_p is not None:
            known = np.asarray(logp_p)
            if known.shape == tuple(x_p.shape[:-1]):
                known = known[cut:]
            else:
                warnings.warn('the logp_p you gave me seems not correct. Will recompute it from logp and x_p.', RuntimeWarning)
                known = None
        if known is None:
            known = dd.logp_and_grad(test, True)[0].cpu().numpy().reshape(tuple(x_p.shape[:-1])[0] - cut, *x_p.shape[1:-1])
        logp_q = dd.logp_and_grad(x_q, True)[0].cpu().numpy()
        logq_p = self.sit._logq_device(test).cpu().numpy().reshape(np.shape(known))
        return bridge(known, logp_q, logq_p, self.sit._logq_device(x_q).cpu().numpy())

    __call__ = run
