"""Integrated autocorrelation time with Sokal's automatic window (the estimator the reference vendors from emcee in
bayesfast/utils/acor.py:79-145); used by the bridge estimator's error (evidence/bridge.py:59-64).  Host NumPy: an FFT
of the (chains, iterations) series of bridge terms."""
import logging

import numpy as np

__all__ = ['integrated_time', 'AutocorrError']


class AutocorrError(Exception):
    """The chain is too short for a reliable estimate."""

    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super().__init__(*args, **kwargs)


def _acf(x):
    n = 1
    while n < len(x):
        n <<= 1
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    a = np.fft.ifft(f * np.conjugate(f))[:len(x)].real
    return a / a[0]


def integrated_time(x, c=5, tol=50, quiet=False):
    """x: (n_t,), (n_t, n_d) or (n_walker, n_t, n_d).  tau per dimension: 2 sum_{t<=W} rho(t) - 1 with the smallest
    window W >= c tau(W); the autocorrelation function is averaged over the walkers."""
    x = np.atleast_1d(x)
    if x.ndim == 1:
        x = x[np.newaxis, :, np.newaxis]
    elif x.ndim == 2:
        x = x[np.newaxis]
    if x.ndim != 3:
        raise ValueError('invalid dimensions.')
    n_w, n_t, n_d = x.shape
    tau = np.empty(n_d)
    for k in range(n_d):
        rho = np.zeros(n_t)
        for w in range(n_w):  # accumulated walker by walker, then divided (utils/acor.py:118-121)
            rho += _acf(x[w, :, k])
        rho /= n_w
        taus = 2.0 * np.cumsum(rho) - 1.0
        inside = np.arange(n_t) < c * taus
        win = np.argmin(inside) if np.any(inside) else n_t - 1
        tau[k] = taus[win]
    short = tol * tau > n_t
    if np.any(short):
        msg = ('The chain is shorter than {0} times the integrated autocorrelation time for {1} parameter(s). Use this '
               'estimate with caution and run a longer chain!\nN/{0} = {2:.0f};\ntau: {3}').format(tol, np.sum(short), n_t / tol, tau)
        if not quiet:
            raise AutocorrError(tau, msg)
        logging.warning(msg)
    return tau
