"""``SystematicResampler``: picks the refit points out of the previous round's samples
(bayesfast/utils/misc.py:21-108; called from ``Recipe._sam_step``, core/recipe.py:1074-1075).

Host-side glue either side of the sampler: one argsort of the all-chain logq array and an index pattern."""
import warnings

import numpy as np

__all__ = ['SystematicResampler']


class SystematicResampler:
    """Systematically resamples the input array.

    nodes : percentiles dividing the intervals, default ``(1., 100.)``
    weights : relative weights of the intervals (None: equal)
    require_unique : raise instead of warn when indices repeat
    """

    def __init__(self, nodes=(1., 100.), weights=None, require_unique=True):
        try:
            self._nodes = np.asarray(nodes, dtype=np.float64)
            assert self._nodes.ndim == 1 and self._nodes.size > 1
            assert np.all(np.diff(self._nodes) > 0)
            assert self._nodes[0] >= 0 and self._nodes[-1] <= 100
            self._n_node = self._nodes.size
        except Exception:
            raise ValueError('invalid value for nodes.')
        if weights is None:
            self._weights = np.ones(self._n_node - 1) / (self._n_node - 1)
        else:
            try:
                self._weights = np.asarray(weights, dtype=np.float64)
                assert np.all(self._weights > 0)
                assert self._weights.ndim == 1
                assert self._weights.size == self._n_node - 1
                self._weights = self._weights / np.sum(self._weights)
            except Exception:
                raise ValueError('invalid value for weights.')
        self._require_unique = bool(require_unique)

    def run(self, a, n):
        """Indices of ``n`` elements of the 1-d array ``a``, evenly spaced in rank between the node percentiles."""
        try:
            a = np.asarray(a, dtype=np.float64)
            assert a.ndim == 1
        except Exception:
            raise ValueError('invalid value for a.')
        try:
            n = int(n)
            assert n > 0
        except Exception:
            raise ValueError('invalid value for n.')
        n_w = (n * self._weights).astype(np.int64)
        n_w[-1] += n - np.sum(n_w)
        n_c = np.cumsum(np.insert(n_w, 0, 0))
        i_all = np.empty(n, dtype=np.int64)
        m = len(a)
        for j in range(self._n_node - 1):
            last = (j == self._n_node - 2)
            i_j = np.linspace(self._nodes[j] * (m - 1) / 100, self._nodes[j + 1] * (m - 1) / 100, n_w[j], last)
            i_all[n_c[j]:n_c[j + 1]] = i_j.astype(np.int64)
        n_unique = np.unique(i_all).size
        if n_unique < i_all.size:
            message = ('{:.1f}% of the resampled points are not unique. Please consider giving me more '
                       'points.'.format(100 - n_unique / i_all.size * 100))
            if self._require_unique:
                raise RuntimeError(message)
            warnings.warn(message, RuntimeWarning)
        return np.argsort(a)[i_all]

    __call__ = run
