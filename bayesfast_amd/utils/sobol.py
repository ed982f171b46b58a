"""Sobol-normal points: the reference's default starting points of ``sample()`` (core/sample.py:106-113 ->
utils/sobol.py:12-61, utils/_sobol.pyx).

The reference generates the Sobol sequence itself from the Joe-Kuo direction numbers ``new-joe-kuo-6.21201``, in Gray-code
order, and skips the first point (the origin).  scipy's unscrambled ``scipy.stats.qmc.Sobol`` is the same sequence from the
same direction numbers (equal to the reference's points bit for bit: tests/golden/sobol.npz), so nothing of it is
re-implemented here."""
import warnings

import numpy as np

__all__ = ['uniform', 'multivariate_normal']

_D_MAX = 21201  # dimensions the direction numbers cover


def _counts(size, skip):
    ok = isinstance(size, (int, np.integer)) or (isinstance(size, float) and size.is_integer())
    if not ok or int(size) < 1:
        raise ValueError('size: a positive number of points is needed, got {!r}.'.format(size))
    ok = isinstance(skip, (int, np.integer)) or (isinstance(skip, float) and skip.is_integer())
    if not ok or int(skip) < 0:
        raise ValueError('skip: a count of leading points to drop (>= 0) is needed, got {!r}.'.format(skip))
    return int(size), int(skip)


def _unit_points(d, size, skip):
    """Points skip .. skip + size - 1 of the d-dimensional sequence, in the unit cube."""
    from scipy.stats import qmc
    if d > _D_MAX:
        raise NotImplementedError('the Sobol direction numbers end at dimension {}; d = {}.'.format(_D_MAX, d))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', UserWarning)  # (scipy: "balance properties require n to be a power of 2")
        return qmc.Sobol(d, scramble=False).random(size + skip)[skip:]


def uniform(low, high, size, skip=1):
    """``size`` Sobol points after the first ``skip``, stretched to the box [low, high) (utils/sobol.py:12-46)."""
    lo, hi = np.atleast_1d(low).astype(float), np.atleast_1d(high).astype(float)
    if lo.ndim != 1 or lo.shape != hi.shape:
        raise ValueError('low and high: two 1-d arrays of one length are needed, got shapes {} and {}.'.format(lo.shape, hi.shape))
    size, skip = _counts(size, skip)
    return lo + (hi - lo) * _unit_points(lo.shape[0], size, skip)


def multivariate_normal(mean, cov, size, skip=1):
    """Sobol points through the normal quantile function and the eigen-decomposition of ``cov`` (utils/sobol.py:49-61)."""
    from scipy.special import ndtri
    from .threads import blas_single_thread
    mean, cov = np.atleast_1d(mean), np.atleast_2d(cov)
    d = mean.shape[0]
    if mean.ndim != 1 or cov.shape != (d, d):
        raise ValueError('mean {} and cov {} do not describe one normal distribution.'.format(mean.shape, cov.shape))
    size, skip = _counts(size, skip)
    # (scipy.stats.norm.ppf is ndtri behind 50 ms of argument handling at 4096 x 64; the points are inside (0, 1))
    points = ndtri(_unit_points(d, size, skip))
    if np.array_equal(cov, np.eye(d)):
        # eigh(I) = (ones, I) and a product with the exact identity leaves every entry as it is: the same bits without the
        # product (sample()'s default starts are this case)
        return mean + points
    with blas_single_thread():  # (utils/threads.py)
        a, w = np.linalg.eigh(cov)
        return mean + (points * a**0.5) @ w.T


def standard_normal_device(d, size, ctx, skip=1):
    """``multivariate_normal(zeros(d), eye(d), size)`` as a device tensor: the Sobol points from the host generator, the normal
    quantile function on the device (``bfhip_ndtri``, Cephes' algorithm as SciPy's: 112 k x 128 points were 0.14 s of a config-5
    GBS run in SciPy's, and 0.28 s of run-time compilation at the first call of ``torch.special.ndtri``)."""
    import torch
    from .. import _lib
    from ..device import _ptr
    size, skip = _counts(size, skip)
    u = ctx.tensor(_unit_points(d, size, skip), torch.float64)
    _lib.check(ctx._lib.bfhip_ndtri(ctx.handle, u.numel(), _ptr(u), _ptr(u)))
    return u
