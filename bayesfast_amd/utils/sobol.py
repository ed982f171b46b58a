"""Sobol-normal points: the reference's default starting points of ``sample()`` (core/sample.py:106-113 ->
utils/sobol.py:12-61, utils/_sobol.pyx).

The reference generates the Sobol sequence itself from the Joe-Kuo direction numbers ``new-joe-kuo-6.21201``, in Gray-code
order, and skips the first point (the origin).  scipy's unscrambled ``scipy.stats.qmc.Sobol`` is the same sequence from the
same direction numbers (equal to the reference's points bit for bit: tests/golden/sobol.npz), so nothing of it is
re-implemented here."""
import warnings

import numpy as np

__all__ = ['uniform', 'multivariate_normal']


def uniform(low, high, size, skip=1):
    """utils/sobol.py:12-46: ``size`` points of the d-dimensional Sobol sequence after ``skip`` points, scaled to [low, high)."""
    from scipy.stats import qmc
    low, high = np.atleast_1d(low), np.atleast_1d(high)
    if not (low.ndim == 1 and low.shape == high.shape):
        raise ValueError('low and high should be 1-d arraies with the same shape, but you give me low.shape = {}, '
                         'high.shape = {}.'.format(low.shape, high.shape))
    try:
        size = int(size)
        assert size > 0
    except Exception:
        raise ValueError('size should be a positive int, instead of {}.'.format(size))
    try:
        skip = int(skip)
        assert skip >= 0
    except Exception:
        raise ValueError('skip should be a non-negative int, instead of {}.'.format(skip))
    d = low.shape[0]
    if d > 21201:
        raise NotImplementedError('d = {} is not supported, as the direction numbers end at d_max = 21201.'.format(d))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', UserWarning)  # (scipy: "balance properties require n to be a power of 2")
        points = qmc.Sobol(d, scramble=False).random(size + skip)[skip:]
    return low + (high - low) * points


def multivariate_normal(mean, cov, size, skip=1):
    """utils/sobol.py:49-61: Sobol points through the normal quantile function and the eigen-decomposition of ``cov``."""
    from scipy.special import ndtri
    from .threads import blas_single_thread
    mean, cov = np.atleast_1d(mean), np.atleast_2d(cov)
    d = mean.shape[0]
    if not (mean.shape == (d,) and cov.shape == (d, d)):
        raise ValueError('the shape of mean is not consistent with the shape of cov.')
    # (scipy.stats.norm.ppf is ndtri behind 50 ms of argument handling at 4096 x 64; the points are inside (0, 1))
    points = ndtri(uniform(np.zeros(d), np.ones(d), size, skip))
    if np.array_equal(cov, np.eye(d)):
        # eigh(I) = (ones, I) and a product with the exact identity leaves every entry as it is: the same bits without the
        # product (sample()'s default starts are this case)
        return mean + points
    with blas_single_thread():  # (utils/threads.py)
        a, w = np.linalg.eigh(cov)
        return mean + (points * a**0.5) @ w.T
