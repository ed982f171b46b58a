"""Gaussianized bridge sampling (bayesfast/evidence/gaussianized.py:179-216): fit a SIT to the first half of the
posterior samples, draw from it, and bridge between the two sample sets."""
import warnings

import numpy as np

from ..transforms.sit import SIT
from .bridge import bridge

__all__ = ['GBS']


class GBS:
    """``GBS(sit=None, parallel_backend=None, n_q=None, f_call=0.05)`` with the reference's meaning; ``parallel_backend``
    is accepted and ignored (``logp`` is called on whole arrays when it accepts them, row by row otherwise)."""

    def __init__(self, sit=None, parallel_backend=None, n_q=None, f_call=0.05):
        if sit is None:
            sit = {}
        if isinstance(sit, dict):
            sit = SIT(**sit)
        elif not isinstance(sit, SIT):
            raise ValueError('invalid value for sit.')
        self.sit = sit
        if n_q is not None:
            try:
                n_q = int(n_q)
                assert n_q > 0
            except Exception:
                raise ValueError('invalid value for n_q.')
        self.n_q = n_q
        if f_call is not None:
            try:
                f_call = float(f_call)
                assert f_call > 0
            except Exception:
                raise ValueError('invalid value for f_call.')
        self.f_call = f_call

    @staticmethod
    def _map(logp, x):
        shape = x.shape
        flat = x.reshape((-1, shape[-1]))
        try:
            out = np.asarray(logp(flat), dtype=np.float64)
            assert out.shape == (flat.shape[0],)
        except Exception:
            out = np.asarray([logp(r) for r in flat], dtype=np.float64)
        return out.reshape(shape[:-1])

    def run(self, x_p, logp, logp_p=None):
        from ..samplers.sample_trace import TraceTuple
        if not callable(logp):
            raise ValueError('logp should be callable.')
        n_call = None
        if isinstance(x_p, TraceTuple):
            n_call = x_p.n_call
            x_p = x_p.get(flatten=False)
        else:
            try:
                x_p = np.asarray(x_p)
                assert 2 <= x_p.ndim <= 3
            except Exception:
                raise ValueError('invalid value for x_p.')
        if self.n_q is not None:
            n_q = self.n_q
        elif self.f_call is not None and n_call is not None:
            n_q = int(n_call * self.f_call)
        else:
            if self.f_call is not None:
                warnings.warn('f_call should be used only when x_p is a TraceTuple. Using equal-sample allocation for now.',
                              RuntimeWarning)
            n_q = int(np.prod(x_p.shape[:-1]))
        if not (x_p.shape[-1] > 1 and np.prod(x_p.shape[:-1]) > 1):
            raise ValueError('invalid shape for x_p.')
        if x_p.shape[0] == 1:
            x_p = x_p[0]
        return self._compute_evidence(logp, x_p, logp_p, n_q)

    __call__ = run

    def _compute_evidence(self, logp, x_p, logp_p, n_q):
        n_half = x_p.shape[0] // 2
        self.sit.fit(data=x_p[:n_half])
        x_q = self.sit.sample(n_q)[0]
        if logp_p is not None:
            try:
                logp_p = np.asarray(logp_p)
                assert logp_p.shape == x_p.shape[:-1]
                logp_p = logp_p[n_half:]
            except Exception:
                warnings.warn('the logp_p you gave me seems not correct. Will recompute it from logp and x_p.', RuntimeWarning)
                logp_p = None
        if logp_p is None:
            logp_p = self._map(logp, x_p[n_half:])
        logp_q = self._map(logp, x_q)
        logq_p = self.sit.logq(x_p[n_half:])
        logq_q = self.sit.logq(x_q)
        return bridge(logp_p, logp_q, logq_p, logq_q)
