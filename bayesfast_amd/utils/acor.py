"""Integrated autocorrelation time with Sokal's automatic window -- the estimator behind the error bar of the bridge
estimate (reference: bayesfast/utils/acor.py:79-145, vendored from emcee; used at evidence/bridge.py:59-64).

All series are transformed at once: one real FFT over the time axis of the (walker, time, dimension) array gives every
autocorrelation function, their walker average rho_k(t), the running sums tau_k(W) = 2 sum_{t <= W} rho_k(t) - 1, and
for each dimension the first window W with W >= c tau_k(W).  Host NumPy."""
import logging

import numpy as np

__all__ = ['integrated_time', 'AutocorrError']


class AutocorrError(Exception):
    """The series is too short for the estimate to be trusted; ``tau`` carries the estimate anyway."""

    def __init__(self, tau, *args, **kwargs):
        self.tau = tau
        super().__init__(*args, **kwargs)


def _as_walkers(x):
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 0:
        x = x.reshape(1)
    shape = {1: (1, x.shape[0], 1), 2: (1,) + x.shape, 3: x.shape}.get(x.ndim)
    if shape is None:
        raise ValueError('invalid dimensions.')
    return x.reshape(shape)


def integrated_time(x, c=5, tol=50, quiet=False):
    """x: (n_t,), (n_t, n_d) or (n_walker, n_t, n_d) -> tau (n_d,).  Raises ``AutocorrError`` (or, with ``quiet``, logs a
    warning) when the series is shorter than ``tol`` autocorrelation times."""
    x = _as_walkers(x)
    n_t = x.shape[1]
    n_fft = 2 << max(n_t - 1, 0).bit_length()           # twice the next power of two: no wrap-around in the products
    spec = np.fft.fft(x - x.mean(axis=1, keepdims=True), n=n_fft, axis=1)
    acov = np.fft.ifft(spec * np.conjugate(spec), axis=1)[:, :n_t].real
    rho = (acov / acov[:, :1]).mean(axis=0)               # (n_t, n_d): every walker normalised by its own variance
    running = 2.0 * np.cumsum(rho, axis=0) - 1.0
    lags = np.arange(n_t)[:, None]
    beyond = lags >= c * running                           # first lag that is at least c running autocorrelation times
    window = beyond.argmax(axis=0)                         # (0 when the window never closes, as the reference's auto_window)
    tau = running[window, np.arange(running.shape[1])]
    too_short = tol * tau > n_t
    if too_short.any():
        text = ('The chain is shorter than {} times the integrated autocorrelation time for {} parameter(s) '
                '(N / {} = {:.0f}, tau = {}): use this estimate with caution and run a longer chain.'
                .format(tol, int(too_short.sum()), tol, n_t / tol, tau))
        if not quiet:
            raise AutocorrError(tau, text)
        logging.warning(text)
    return tau
