"""Sliced Iterative Transform: the generative model of Gaussianized bridge sampling (bayesfast/transforms/sit.py).

Each iteration rotates the data with FastICA and then Gaussianizes every rotated coordinate by a monotone map
``norm.ppf(KDE cdf)`` represented as a piecewise cubic (transforms/sit.py:223-255,305-321).  What runs where:

* the construction of the ~100-knot splines: host;  FastICA: scikit-learn's algorithm on the device (``transforms/ica.py``:
  the (n x d)(d x d) products and the tanh on the GPU, the d x d eigen-decompositions on the host);
* the KDE cdf at the knots -- an (n_data x n_knot) reduction per coordinate, the dominant cost of ``fit`` --
  ``bfhip_kde_cdf``; the rotations of the whole data set: device matmuls; applying the d splines to all points
  (``fit``'s update of the data, ``forward_transform`` / ``logq``, ``backward_transform`` / ``sample``):
  ``bfhip_spline_apply``.  The data stay on the GPU between iterations.

``mvn_generator`` defaults, as in the reference (transforms/sit.py:214-221), to its Sobol-normal points (``utils/sobol.py``).
There is no plotting."""
import warnings

import numpy as np

from ..utils.spline import GaussianizingSpline, SplineTable

__all__ = ['SIT']


class SIT:
    """Same constructor arguments as the reference where they apply (transforms/sit.py:60-75)."""

    def __init__(self, n_iter=10, parallel_backend=None, bw_factor=1., m_ica=20000, random_generator=None, m_plot=8,
                 cubic_options=None, ica_options=None, mvn_generator=None):
        try:
            self.n_iter = int(n_iter)
            assert self.n_iter > 0
        except Exception:
            raise ValueError('n_iter should be a positive int.')
        try:
            self.bw_factor = float(bw_factor)
            assert self.bw_factor > 0
        except Exception:
            raise ValueError('bw_factor should be a positive float.')
        try:
            self.m_ica = int(m_ica)
            assert self.m_ica > 0
        except Exception:
            raise ValueError('m_ica should be a positive int.')
        self.random_generator = np.random.default_rng(random_generator)
        self.cubic_options = dict(cubic_options or {})
        self.ica_options = dict(ica_options if ica_options is not None else {'max_iter': 100})
        if mvn_generator is not None and not callable(mvn_generator):
            raise ValueError('invalid value for mvn_generator.')
        if mvn_generator is None:  # transforms/sit.py:214-221
            from ..utils.sobol import multivariate_normal as mvn_generator
        self.mvn_generator = mvn_generator
        self._data = None
        self._tables = []
        self._A = self._B = self._m = self._logdetA = None

    @classmethod
    def _from_parts(cls, A, B, m, logdetA, splines):
        """A fitted model from its arrays: rotations (n_iter, d, d), means (n_iter, d), log|det A| (n_iter,) and, per
        iteration and coordinate, the (knots, values, coefficient rows) of the Gaussianizing spline."""
        import torch

        class _S:
            def __init__(self, x, y, c):
                self.x, self.y, self.c = np.asarray(x), np.asarray(y), np.asarray(c)

        self = cls(n_iter=len(splines))
        ctx = self._ctx()
        self._A, self._B, self._m = np.asarray(A), np.asarray(B), np.asarray(m)
        self._logdetA = np.asarray(logdetA)
        self._tables = [SplineTable([_S(*t) for t in it], ctx) for it in splines]
        self._data = torch.zeros((1, self._A.shape[-1]), dtype=torch.float64, device=ctx.device)
        self._weights = np.ones(1)
        return self

    i_iter = property(lambda self: len(self._tables))
    dim = property(lambda self: int(self._data.shape[-1]))
    weights = property(lambda self: self._weights)

    @property
    def data(self):
        return None if self._data is None else self._data.cpu().numpy()

    def add_iter(self, n):
        self.n_iter += int(n)

    def _ctx(self):
        from ..device import get_context
        return get_context()

    # ---- one iteration: transforms/sit.py:229-255 ----
    def _ica(self, x_dev):
        """``FastICA(**ica_options).fit`` on at most ``m_ica`` of the points (transforms/sit.py:235-244), on the device
        (``transforms/ica.py``: scikit-learn's algorithm with its defaults; options other than ``max_iter``, ``tol``,
        ``random_state``, ``w_init`` go to scikit-learn itself, on the host, as the reference does)."""
        import torch
        io = dict(self.ica_options)
        if 'random_state' not in io:
            io['random_state'] = int(self.random_generator.integers(0, 2**32))
        n = int(x_dev.shape[0])
        n_ica = min(n, self.m_ica)
        rows = self.random_generator.choice(n, n_ica, False)
        if set(io) <= {'max_iter', 'tol', 'random_state', 'w_init'}:
            from .ica import fastica_device
            sub = x_dev[torch.as_tensor(rows, device=x_dev.device)]
            comp, mean, _ = fastica_device(sub, random_state=io['random_state'], max_iter=io.get('max_iter', 200),
                                           tol=io.get('tol', 1e-4), w_init=io.get('w_init'))
            return comp, mean
        from sklearn.decomposition import FastICA
        ica = FastICA(**io)
        ica.fit(x_dev.cpu().numpy()[rows])
        return ica.components_, ica.mean_

    def _gaussianize(self, y):
        """Splines of all coordinates of y (n, d) device tensor: knots and edge points on the host, KDE cdfs on the device."""
        import torch
        from scipy.special import ndtri
        from .. import _lib
        from ..device import _ptr
        ctx = self._ctx()
        n, d = y.shape
        yT = y.T.contiguous()                       # (d, n): one row per coordinate
        # kde bandwidth (utils/kde.py:85-151): sqrt of the weighted variance times Scott's factor times bw_factor
        wn = self._weights / np.sum(self._weights)
        w = ctx.tensor(wn)                          # (kde normalises its weights, utils/kde.py:76-77)
        neff = 1. / np.sum(wn**2)
        mean = (yT * w).sum(1) / w.sum()
        var = (((yT - mean[:, None])**2) * w).sum(1) / w.sum() / (1. - float(np.sum(wn**2)))  # np.cov(aweights=w, bias=False)
        h = torch.sqrt(var) * (neff**(-1. / 5)) * self.bw_factor
        y_sorted_dev = torch.sort(yT, dim=1).values    # (percentiles of the knots and edge points)
        hd = h.contiguous()
        splines = self._build_on_device(ctx, y_sorted_dev, yT, w, hd)
        if splines is not None:
            return SplineTable(splines, ctx)
        # host construction (options outside the device builder's range): one copy of the sorted coordinates to the host, the
        # function values of every round of all d splines from ONE kde-cdf call
        y_sorted = y_sorted_dev.cpu().numpy()
        splines = GaussianizingSpline.build_many(y_sorted, self._batch_fun(ctx, yT, w, hd, y_sorted), presorted=True, **self.cubic_options)
        return SplineTable(splines, ctx)

    @staticmethod
    def _batch_fun(ctx, yT, w, hd, y_sorted, rows=None):
        """``batch_fun`` of ``GaussianizingSpline.build_many`` for the coordinates ``rows`` (default: all): norm.ppf(kde.cdf(.)) of
        one round's requests in one device call (rows padded with their last point)."""
        import torch
        from scipy.special import ndtri
        from .. import _lib
        from ..device import _ptr
        d, n = yT.shape
        rows = list(range(d)) if rows is None else list(rows)

        def batch_fun(requests):
            m = max(r.size for r in requests if r is not None)
            pts = np.empty((d, m))
            pts[:] = y_sorted[:, :1]
            for j, r in zip(rows, requests):
                if r is not None:
                    pts[j, :r.size] = r
                    pts[j, r.size:] = r[-1]
            p = ctx.tensor(pts)
            out = torch.empty_like(p)
            _lib.check(ctx._lib.bfhip_kde_cdf(ctx.handle, d, n, _ptr(yT), _ptr(w), _ptr(hd), m, _ptr(p), _ptr(out)))
            vals = ndtri(out.cpu().numpy())          # (scipy's norm.ppf is this function)
            return [None if r is None else vals[j, :r.size] for j, r in zip(rows, requests)]
        return batch_fun

    _BUILD_STRIDE = 512

    def _build_on_device(self, ctx, y_sorted_dev, yT, w, hd):
        """All d splines of the iteration in ONE launch (``bfhip_spline_build``: a workgroup per coordinate runs the whole construction,
        cdf sums included) and one copy of the finished knots, values and coefficient rows to the host.  None when the options are
        outside the kernel's range (then the host construction runs).  A coordinate the kernel gives up on (more than 512 knots, a
        singular slope system) is built by the host construction; the reference's own failures are raised as it raises them."""
        import torch
        from .. import _lib
        from ..device import _ptr
        o = dict(bins=100, edge_bins=1, edge_points=10, max_width=5, split=4, max_add=5)
        if not set(self.cubic_options) <= set(o) or not hasattr(ctx, 'handle'):
            return None
        o.update(self.cubic_options)
        try:
            bins, edge_points, split, max_add = int(o['bins']), int(o['edge_points']), int(o['split']), int(o['max_add'])
            edge_bins = int(min(o['edge_bins'], bins // 4))
            max_width = float(o['max_width'])
        except Exception:
            return None
        if edge_bins < 1:
            return None
        grid = np.linspace(0, 100, bins + 1)[edge_bins:-edge_bins]
        inner = np.linspace(0, 100, edge_points + 2)[1:-1]
        stride = self._BUILD_STRIDE
        if not (3 <= grid.size <= stride and 1 <= inner.size <= 128 and split >= 2 and max_add >= 0 and max_width > 0):
            return None
        d, n = yT.shape
        ox, oy = ctx.empty((d, stride)), ctx.empty((d, stride))
        oc = ctx.empty((d, 4 * (stride + 1)))
        on = torch.zeros((d, 2), dtype=torch.int32, device=yT.device)
        grid_d, inner_d = ctx.tensor(grid), ctx.tensor(inner)     # (named: they must outlive the launch's argument list)
        _lib.check(ctx._lib.bfhip_spline_build(ctx.handle, d, n, _ptr(y_sorted_dev), _ptr(yT), _ptr(w), _ptr(hd), grid.size,
                                               _ptr(grid_d), edge_bins, inner.size, _ptr(inner_d), max_width, split,
                                               max_add, stride, _ptr(ox), _ptr(oy), _ptr(oc), _ptr(on)))
        cnt = on.cpu().numpy()
        ox, oy, oc = ox.cpu().numpy(), oy.cpu().numpy(), oc.cpu().numpy()
        if np.any(cnt[:, 1] & 4):
            raise ValueError('the knots are too unevenly spaced.')
        splines, redo = [], []
        for j in range(d):
            m, flag = int(cnt[j, 0]), int(cnt[j, 1])
            if flag & (1 | 2 | 8) or m < 2:
                redo.append(j)
                splines.append(None)
                continue
            if flag & 16:
                warnings.warn(RuntimeWarning('Not all the intervals are monotone.'))
            splines.append(GaussianizingSpline.from_arrays(ox[j, :m].copy(), oy[j, :m].copy(), oc[j, :4 * (m + 1)].reshape(m + 1, 4).copy()))
        if redo:
            y_sorted = y_sorted_dev.cpu().numpy()
            fun = self._batch_fun(ctx, yT, w, hd, y_sorted, rows=redo)
            for j, sp in zip(redo, GaussianizingSpline.build_many(y_sorted[redo], fun, presorted=True, **self.cubic_options)):
                splines[j] = sp
        return splines

    def fit(self, data=None, weights=None, n_run=None):
        """transforms/sit.py:257-341 (without the plots)."""
        from ..utils.threads import blas_single_thread
        with blas_single_thread():     # (see evidence/gbs.py: spinning BLAS workers stall the whole process in a CPU-quota container)
            return self._fit(data, weights, n_run)

    def _fit(self, data, weights, n_run):
        import torch
        ctx = self._ctx()
        if data is not None and isinstance(data, torch.Tensor):
            # (device-resident callers, GBS on a TraceTuple: the samples never visit the host)
            if data.numel() == 0 or data.dim() < 2:
                raise ValueError('invalid value for data.')
            data = data.to(torch.float64).reshape(-1, data.shape[-1])
            if data.shape[-1] == 1:
                raise ValueError('I cannot do rotations for only one variable.')
            n = data.shape[0]
            if weights is not None:
                weights = np.asarray(weights, dtype=np.float64)
                if weights.shape != (n,):
                    raise ValueError('invalid value for weights.')
                self._weights = weights
            else:
                self._weights = np.ones(n) / n
            self._data = ctx.tensor(data).clone()
            self._data_init = None
            d = data.shape[-1]
            self._tables = []
            self._A, self._B = np.zeros((0, d, d)), np.zeros((0, d, d))
            self._m, self._logdetA = np.zeros((0, d)), np.zeros(0)
        elif data is not None:
            try:
                data = np.array(data, dtype=np.float64)
                assert data.size > 0
            except Exception:
                raise ValueError('invalid value for data.')
            if data.ndim == 2:
                pass
            elif data.ndim >= 3:
                data = data.reshape((-1, data.shape[-1]))
            else:
                raise ValueError('invalid shape for data.ndim.')
            if data.shape[-1] == 1:
                raise ValueError('I cannot do rotations for only one variable.')
            n = data.shape[0]
            if weights is not None:
                weights = np.asarray(weights, dtype=np.float64)
                if weights.shape != (n,):
                    raise ValueError('invalid value for weights.')
                self._weights = weights
            else:
                self._weights = np.ones(n) / n
            self._data = ctx.tensor(data)
            self._data_init = data.copy()
            d = data.shape[-1]
            self._tables = []
            self._A, self._B = np.zeros((0, d, d)), np.zeros((0, d, d))
            self._m, self._logdetA = np.zeros((0, d)), np.zeros(0)
        elif self._data is None:
            raise ValueError('you have not given me the data to fit.')
        if n_run is None:
            n_run = self.n_iter - self.i_iter
        else:
            n_run = int(n_run)
            if n_run <= 0:
                raise ValueError('invalid value for n_run.')
            if n_run > self.n_iter - self.i_iter:
                self.n_iter = self.i_iter + n_run
        for _ in range(n_run):
            comp, ica_mean = self._ica(self._data)
            # y = ica.transform(x) scaled to unit variance; A = components / std, B = inv(A), m = mean(x) (:237-243)
            m = self._data.mean(0).cpu().numpy()
            y = (self._data - ctx.tensor(ica_mean)) @ ctx.tensor(comp.T.copy())
            s = y.std(0, unbiased=False)
            y = y / s
            A = comp / s.cpu().numpy()[:, None]
            table = self._gaussianize(y)
            self._tables.append(table)
            self._data = table.apply('evaluate', y)
            self._A = np.concatenate((self._A, A[None]), 0)
            self._B = np.concatenate((self._B, np.linalg.inv(A)[None]), 0)
            self._m = np.concatenate((self._m, m[None]), 0)
            self._logdetA = np.append(self._logdetA, np.log(np.abs(np.linalg.det(A))))
            finite = torch.isfinite(self._data).all(1)
            if not bool(finite.all()):
                warnings.warn('inf encountered for some data points. We will remove these inf points for now.', RuntimeWarning)
                self._data = self._data[finite]
                self._weights = self._weights[finite.cpu().numpy()]

    # ---- transforms: transforms/sit.py:372-459 ----
    def _points(self, x):
        try:
            y = np.array(x, dtype=np.float64)
        except Exception:
            raise ValueError('invalid value for x.')
        if y.ndim == 1:
            y = y[None]
        if y.shape[-1] != self.dim:
            raise ValueError('invalid shape for x.')
        return y.reshape((-1, y.shape[-1])), y.shape

    def _rotations_on_device(self):
        """The iterations' means and rotation matrices as device tensors (cached per fitted iteration count)."""
        ctx = self._ctx()
        c = getattr(self, '_dev_rot', None)
        if c is None or c[0] != self.i_iter or c[1] is not ctx:
            c = (self.i_iter, ctx, [ctx.tensor(self._m[i]) for i in range(self.i_iter)],
                 [ctx.tensor(self._A[i].T.copy()) for i in range(self.i_iter)], [ctx.tensor(self._B[i].T.copy()) for i in range(self.i_iter)])
            self._dev_rot = c
        return c[2], c[3], c[4]

    def _forward_device(self, y):
        """``forward_transform`` of y (n, d) device tensor -> (y', log|J|) device tensors."""
        import torch
        m, At, _ = self._rotations_on_device()
        log_j = torch.zeros(y.shape[0], dtype=torch.float64, device=y.device)
        for i in range(self.i_iter):
            y = (y - m[i]) @ At[i]
            log_j += torch.log(self._tables[i].apply('derivative', y)).sum(1)
            y = self._tables[i].apply('evaluate', y)
        log_j += float(np.sum(self._logdetA))
        return y, log_j

    def _backward_device(self, x):
        """``backward_transform`` of x (n, d) device tensor -> (x', log|J|) device tensors."""
        import torch
        m, _, Bt = self._rotations_on_device()
        log_j = torch.zeros(x.shape[0], dtype=torch.float64, device=x.device)
        for i in reversed(range(self.i_iter)):
            x = self._tables[i].apply('solve', x)
            log_j += torch.log(self._tables[i].apply('derivative', x)).sum(1)
            x = x @ Bt[i] + m[i]
        log_j += float(np.sum(self._logdetA))
        return x, log_j

    def _logq_device(self, x):
        """``logq`` of x (n, d) device tensor -> (n,) device tensor (the standard normal's log-density summed on the device)."""
        y, log_j = self._forward_device(x)
        return (-0.5 * y * y - 0.9189385332046727).sum(1) + log_j

    def _sample_device(self, n):
        """``sample(n)[0]`` as a device tensor (the default Sobol-normal generator; None for a user's generator)."""
        from ..utils import sobol
        if self.mvn_generator is not sobol.multivariate_normal:
            return None
        y = sobol.standard_normal_device(self.dim, int(n), self._ctx())
        return self._backward_device(y)[0]

    def forward_transform(self, x, use_parallel=False):
        flat, shape = self._points(x)
        y, log_j = self._forward_device(self._ctx().tensor(flat))
        return y.cpu().numpy().reshape(shape), log_j.cpu().numpy().reshape(shape[:-1])

    def backward_transform(self, y, use_parallel=False):
        flat, shape = self._points(y)
        x, log_j = self._backward_device(self._ctx().tensor(flat))
        return x.cpu().numpy().reshape(shape), log_j.cpu().numpy().reshape(shape[:-1])

    def logq(self, x, use_parallel=False):
        y, log_j = self.forward_transform(x)
        return np.sum(-0.5 * y * y - 0.9189385332046727, axis=-1) + log_j  # norm.logpdf

    def sample(self, n, use_parallel=False):
        try:
            n = int(n)
            assert n > 0
        except Exception:
            raise ValueError('n should be a positive int.')
        from ..utils import sobol
        if self.mvn_generator is sobol.multivariate_normal:   # the default: Sobol points, the quantile function on the device
            y = sobol.standard_normal_device(self.dim, n, self._ctx()).cpu().numpy()
        else:
            y = self.mvn_generator(np.zeros(self.dim), np.eye(self.dim), n)
        x, log_j = self.backward_transform(y)
        return x, log_j, y
