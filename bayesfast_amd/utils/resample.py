"""Systematic resampling of the previous round's samples by their surrogate log-density.

Role in the reference: ``bayesfast.utils.misc.SystematicResampler`` (utils/misc.py:21-108), called by
``Recipe._sam_step`` (core/recipe.py:1074-1075) to choose the points at which the true model is evaluated before
the next surrogate fit.  Written here from its specification, as two separable pieces:

* ``systematic_ranks`` -- WHICH order statistics are taken.  The percentile nodes p_0 < ... < p_K cut the sorted
  array into K intervals; interval j receives ``floor(n w_j)`` of the n draws (the last one also the remainder)
  and spends them on equally spaced ranks between ``p_j (m - 1) / 100`` and ``p_{j+1} (m - 1) / 100`` (the right
  edge included only for the last interval), truncated to integers.  Pure index arithmetic, O(n), on the host.
* the selection of those order statistics -- a stable sort of the values.  For device tensors it runs on the GPU
  (``bfhip_sort_keys``, radix sort of order-preserving keys); sharded over ranks it becomes the exact distributed
  selection of ``bayesfast_amd.core.refit.select_rows_sharded``.  Host arrays take NumPy's stable argsort.
"""
import warnings

import numpy as np

__all__ = ['SystematicResampler', 'systematic_ranks', 'draws_per_interval']


def _as_percent_nodes(nodes):
    p = np.array(nodes, dtype=np.float64, ndmin=1)
    ok = p.ndim == 1 and p.size >= 2 and bool(np.all(p[1:] > p[:-1])) and p[0] >= 0. and p[-1] <= 100.
    if not ok:
        raise ValueError('invalid value for nodes.')
    return p


def _as_interval_weights(weights, n_interval):
    if weights is None:
        return np.full(n_interval, 1. / n_interval)
    w = np.array(weights, dtype=np.float64, ndmin=1)
    if w.ndim != 1 or w.size != n_interval or not bool(np.all(w > 0.)):
        raise ValueError('invalid value for weights.')
    return w / w.sum()


def draws_per_interval(n, weights):
    """How the n draws are shared out: ``floor(n w_j)`` each, the rounding loss goes to the last interval."""
    share = np.floor(n * np.asarray(weights)).astype(np.int64)
    share[-1] = n - int(share[:-1].sum())
    return share


def systematic_ranks(m, n, nodes=(1., 100.), weights=None):
    """Ranks (0-based positions in the ascending order of an m-element array) of the n systematic draws."""
    p = _as_percent_nodes(nodes)
    w = _as_interval_weights(weights, p.size - 1)
    share = draws_per_interval(int(n), w)
    edge = p * (m - 1) / 100                       # rank coordinate of every node
    first = np.concatenate(([0], np.cumsum(share)[:-1]))
    which = np.repeat(np.arange(share.size), share)  # interval of every draw
    t = np.arange(int(n)) - first[which]             # its position inside the interval
    closed = which == share.size - 1                 # only the last interval reaches its right edge
    parts = np.where(closed, share[which] - 1, share[which])
    width = edge[which + 1] - edge[which]
    with np.errstate(divide='ignore', invalid='ignore'):
        pitch = width / parts
    # (a single draw in the closed interval has no pitch: it sits on the left edge)
    coord = np.where(parts > 0, t * pitch, 0.) + edge[which]
    last = first + share - 1
    end_of_closed = closed & (np.arange(int(n)) == last[which]) & (share[which] > 1)
    coord = np.where(end_of_closed, edge[which + 1], coord)
    return coord.astype(np.int64)


class SystematicResampler:
    """Systematically resamples the input array (same constructor and call signature as the reference's).

    Parameters
    ----------
    nodes : 1-d array_like of float, percentiles dividing the intervals, default ``(1., 100.)``
    weights : 1-d array_like of float or None, relative weights of the intervals (None: equal)
    require_unique : bool, raise (True) or warn (False) when the same element is drawn twice
    """

    def __init__(self, nodes=(1., 100.), weights=None, require_unique=True):
        self._nodes = _as_percent_nodes(nodes)
        self._weights = _as_interval_weights(weights, self._nodes.size - 1)
        self._require_unique = bool(require_unique)

    nodes = property(lambda self: self._nodes.copy())
    weights = property(lambda self: self._weights.copy())

    def ranks(self, m, n):
        """The ranks drawn out of m elements, after the uniqueness check."""
        try:
            n = int(n)
            if n <= 0:
                raise ValueError
        except Exception:
            raise ValueError('invalid value for n.')
        r = systematic_ranks(int(m), n, self._nodes, self._weights)
        n_distinct = int(np.count_nonzero(np.diff(r)) + 1)  # (ranks ascend within and across intervals)
        if n_distinct < r.size:
            text = ('{:.1f}% of the resampled points are not unique. Please consider giving me more '
                    'points.'.format(100. * (1. - n_distinct / r.size)))
            if self._require_unique:
                raise RuntimeError(text)
            warnings.warn(text, RuntimeWarning)
        return r

    def run(self, a, n):
        """Indices of the n resampled elements of the 1-d array (or device tensor) ``a``."""
        if hasattr(a, 'is_cuda') and a.is_cuda:
            from ..core.refit import device_argsort
            if a.dim() != 1:
                raise ValueError('invalid value for a.')
            r = self.ranks(a.shape[0], n)
            order = device_argsort(a)
            return order[order.new_tensor(r)]
        try:
            a = np.asarray(a, dtype=np.float64)
            if a.ndim != 1:
                raise ValueError
        except Exception:
            raise ValueError('invalid value for a.')
        return np.argsort(a, kind='stable')[self.ranks(a.size, n)]

    __call__ = run
