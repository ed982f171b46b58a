"""Monotone piecewise-cubic "Gaussianizing" map of one coordinate (the reference's ``cubic_spline``,
bayesfast/utils/cubic.py:19-260 with the kernels of utils/_cubic.pyx).

Construction is host-side control logic on ~100 knots (written from the algorithm's description below; the function
values at the knots come from the device, ``bfhip_kde_cdf``); applying the finished maps to batches of points is the
device kernel ``bfhip_spline_apply`` (``SplineTable``).

Algorithm (utils/cubic.py:61-140):
  1. knots = distinct percentiles of the samples on a uniform percentile grid with ``edge_bins`` cells cut off either end;
  2. beyond the outermost knots the map continues linearly, with the least-squares slope through the origin of
     ``fun`` on ``edge_points`` percentile points of the samples outside (left: below knot ``edge_bins``; right: above
     knot ``-edge_bins - 1``), measured from the outermost knot;
  3. interior gaps wider than ``max_width`` mean gaps get ``ceil(width / split) - 1`` extra, equally spaced knots;
  4. a C2 cubic spline with those two end slopes clamped is fitted (tridiagonal system for the knot slopes);
  5. while some interval is not monotone and fewer than ``max_add`` rounds were spent, the offending intervals get
     ``split - 1`` extra knots each and the spline is refitted (before the last round, flat stretches of the values are
     straightened); intervals still not monotone become straight segments.
"""
import warnings

import numpy as np
from scipy.linalg import solve_banded

__all__ = ['GaussianizingSpline', 'SplineTable']


def _monotone_flags(coef, knots):
    """One flag per interior interval: is the cubic monotone increasing on it (utils/_cubic.pyx:166-186,336-343)?"""
    c = coef[1:-1]
    w = np.diff(knots)
    c0, c1, c2 = c[:, 0], c[:, 1], c[:, 2]
    slope_l = c2                                   # derivative at the left end of the interval
    slope_r = 3 * c0 * w * w + 2 * c1 * w + c2
    bend_l = c1                                    # (half the) second derivative at the two ends
    bend_r = 3 * c0 * w + c1
    disc = c1 * c1 - 3 * c0 * c2
    ok = (slope_l > 0) & (slope_r > 0) & (bend_l * bend_r >= 0)
    ok |= (c0 > 0) & (disc < 0)
    return ok


def percentile_sorted(xs, q):
    """``np.percentile(x, q)`` (the default linear method) from the SORTED copy xs of x: the same two order statistics and
    the same interpolation arithmetic as numpy's (virtual index (n - 1) q, lerp a + (b - a) t, taken from the
    other end for t >= 1/2), so the result is bit-identical -- without a selection pass over the data per call
    (tests/test_evidence.py checks it against numpy)."""
    xs = np.asarray(xs, dtype=np.float64)
    n = xs.shape[0]
    quant = np.true_divide(np.asarray(q, dtype=np.float64), np.float64(100))
    virt = (n - 1) * quant
    lo = np.floor(virt).astype(np.intp)
    hi = lo + 1
    over = virt >= n - 1
    lo[over] = -1
    hi[over] = -1
    under = virt < 0
    lo[under] = 0
    hi[under] = 0
    t = virt - lo
    a, b = xs[lo], xs[hi]
    step = b - a
    out = a + step * t
    far = t >= 0.5
    out[far] = (b - step * (1 - t))[far]
    return out


class GaussianizingSpline:
    """``cubic_spline(x_all, fun, **options)``: knots ``x``, values ``y`` and coefficient rows ``c`` (n + 1, 4)."""

    def __init__(self, x_all, fun, bins=100, edge_bins=1, edge_points=10, max_width=5, split=4, max_add=5, presorted=False):
        steps = self._build(x_all, presorted, bins, edge_bins, edge_points, max_width, split, max_add)
        try:
            pts = next(steps)
            while True:
                pts = steps.send(np.asarray(fun(pts), dtype=np.float64))
        except StopIteration:
            pass

    @classmethod
    def from_arrays(cls, x, y, c):
        """A finished spline from its knots, values and coefficient rows (the device builder's output)."""
        out = cls.__new__(cls)
        out.x, out.y, out.c = x, y, c
        out._k_left, out._k_right = float(c[0, 2]), float(c[-1, 2])
        return out

    @classmethod
    def build_many(cls, rows, batch_fun, presorted=False, **options):
        """One spline per row of ``rows``, built side by side: every spline asks for function values at a few points
        several times on its way (knots, edge points, extra knots); ``batch_fun(requests)`` gets the requests of one round
        -- a list with a point array or None per row -- and returns the list of value arrays, so that the d coordinates of a
        SIT iteration cost a handful of device calls instead of eight per coordinate."""
        out = [cls.__new__(cls) for _ in rows]
        steps = [o._build(r, presorted, **options) for o, r in zip(out, rows)]
        req = []
        for g in steps:
            try:
                req.append(next(g))
            except StopIteration:
                req.append(None)
        while any(r is not None for r in req):
            vals = batch_fun(req)
            nxt = []
            for g, r, v in zip(steps, req, vals):
                if r is None:
                    nxt.append(None)
                    continue
                try:
                    nxt.append(g.send(np.asarray(v, dtype=np.float64)))
                except StopIteration:
                    nxt.append(None)
            req = nxt
        return out

    def _build(self, x_all, presorted=False, bins=100, edge_bins=1, edge_points=10, max_width=5, split=4, max_add=5):
        """The construction as a generator: yields the points it needs ``fun`` at, is sent the values."""
        xs = np.asarray(x_all, dtype=np.float64).reshape(-1)
        if not presorted:
            xs = np.sort(xs)  # one sort serves the three percentile sets below
        edge_bins = int(min(edge_bins, bins // 4))
        grid = np.linspace(0, 100, bins + 1)[edge_bins:-edge_bins]
        self.x = np.unique(percentile_sorted(xs, grid))
        self.y = np.asarray((yield self.x), dtype=np.float64)
        inner = np.linspace(0, 100, edge_points + 2)[1:-1]
        below = xs[:np.searchsorted(xs, self.x[edge_bins], 'left')]            # x_all[x_all < knot], sorted
        above = xs[np.searchsorted(xs, self.x[-edge_bins - 1], 'right'):]      # x_all[x_all > knot], sorted
        self._k_left = yield from self._edge_slope(below, self.x[0], self.y[0], inner)
        self._k_right = yield from self._edge_slope(above, self.x[-1], self.y[-1], inner)
        yield from self._fill_wide_gaps(max_width, split)
        self._fit()
        good = _monotone_flags(self.c, self.x)
        rounds = 0
        while not good.all() and rounds < max_add:
            # np.linspace(x[j], x[j + 1], split + 1)[1:-1] of every offending interval at once (linspace's own arithmetic:
            # k * ((b - a) / split) + a; a call per interval was 0.15 s of a ten-iteration SIT fit)
            bad = np.flatnonzero(~good)
            lo, hi = self.x[bad], self.x[bad + 1]
            new = (np.arange(1, split, dtype=np.float64)[None, :] * ((hi - lo) / split)[:, None] + lo[:, None]).reshape(-1)
            yield from self._insert(new)
            if rounds == max_add - 1:
                self._straighten_flat_values()
            self._fit()
            good = _monotone_flags(self.c, self.x)
            rounds += 1
        if not good.all():  # utils/cubic.py:128-136: straight segments where the cubic still turns
            for i in np.flatnonzero(~good) + 1:
                self.c[i] = (0., 0., (self.y[i] - self.y[i - 1]) / (self.x[i] - self.x[i - 1]), self.y[i - 1])
            if not _monotone_flags(self.c, self.x).all():
                warnings.warn(RuntimeWarning('Not all the intervals are monotone.'))

    @staticmethod
    def _edge_slope(outside, knot, value, inner):
        t = percentile_sorted(outside - knot, inner)  # (outside is sorted, and stays so under the shift)
        f = np.asarray((yield t + knot)) - value
        return np.sum(t * f) / np.sum(t * t)

    def _insert(self, new):
        at = np.searchsorted(self.x, new)
        vals = np.asarray((yield new), dtype=np.float64)
        self.x = np.insert(self.x, at, new)
        self.y = np.insert(self.y, at, vals)

    def _fill_wide_gaps(self, max_width, split):
        rel = np.diff(self.x)
        rel = rel / np.mean(rel)
        n = self.x.size
        first = 0  # the first / last gap that is not wider than max_width bound the region that gets refined
        while rel[first] > max_width:
            first += 1
            if first >= n - 2:
                break
        last = n - 2
        while rel[last] > max_width:
            last -= 1
            if last <= 0:
                break
        if first > last:
            raise ValueError('the knots are too unevenly spaced.')
        wide = np.flatnonzero(rel[first:last + 1] > max_width) + first
        if wide.size:
            new = np.concatenate([np.linspace(self.x[j], self.x[j + 1], int(np.ceil(rel[j] / split)) + 1)[1:-1] for j in wide])
            yield from self._insert(new)

    def _fit(self):
        """Clamped C2 cubic spline: slopes s at the knots from the tridiagonal continuity system, then the local
        coefficients in powers of (x - x_left) (utils/cubic.py:142-184)."""
        n = self.x.size
        w = np.diff(self.x)
        chord = np.diff(self.y) / w
        band = np.zeros((3, n))
        rhs = np.empty(n)
        band[1, 1:-1] = 2 * (w[:-1] + w[1:])
        band[0, 2:] = w[:-1]
        band[2, :-2] = w[1:]
        rhs[1:-1] = 3 * (w[1:] * chord[:-1] + w[:-1] * chord[1:])
        band[1, 0] = band[1, -1] = 1.
        rhs[0], rhs[-1] = self._k_left, self._k_right
        s = solve_banded((1, 1), band, rhs, check_finite=False)
        t = (s[:-1] + s[1:] - 2 * chord) / w
        self.c = np.zeros((n + 1, 4))
        self.c[0, 2:] = (self._k_left, self.y[0])
        self.c[-1, 2:] = (self._k_right, self.y[-1])
        self.c[1:-1, 0] = t / w
        self.c[1:-1, 1] = (chord - s[:-1]) / w - t
        self.c[1:-1, 2] = s[:-1]
        self.c[1:-1, 3] = self.y[:-1]

    def _straighten_flat_values(self):
        """Runs of (almost) non-increasing values are replaced by the straight line across them (utils/cubic.py:190-216)."""
        w = np.diff(self.x)
        k = np.diff(self.y) / w
        bad = np.flatnonzero(k < 1e-10)
        while bad.size:
            while bad.size:
                i = 0
                start = np.max(bad[0] - 1, 0)  # (as in the reference: numpy max over a scalar, i.e. bad[0] - 1)
                while i < bad.size - 1 and bad[i + 1] - bad[i] <= 2:
                    i += 1
                end = min(bad[i] + 1, k.size - 1)
                line = (self.y[end + 1] - self.y[start]) / (self.x[end + 1] - self.x[start])
                for j in range(start + 1, end + 1):
                    self.y[j] = self.y[start] + line * (self.x[j] - self.x[start])
                bad = bad[i + 1:]
            k = np.diff(self.y) / w
            bad = np.flatnonzero(k < 1e-8)


class SplineTable:
    """The d splines of one SIT iteration packed for ``bfhip_spline_apply`` (knot offsets, knots, values, coefficient
    rows) and resident on the device."""

    def __init__(self, splines, ctx):
        import torch
        self.d = len(splines)
        off = np.concatenate(([0], np.cumsum([s.x.size for s in splines]))).astype(np.int32)
        self.ctx = ctx
        self.off = torch.as_tensor(off, device=ctx.device)
        self.knots = ctx.tensor(np.concatenate([s.x for s in splines]))
        self.values = ctx.tensor(np.concatenate([s.y for s in splines]))
        self.coef = ctx.tensor(np.concatenate([s.c.reshape(-1) for s in splines]))
        self.splines = splines

    def apply(self, mode, x):
        """mode 'evaluate' | 'derivative' | 'solve' on x (n, d) device tensor -> (n, d)."""
        import torch
        from .. import _lib
        from ..device import _ptr
        x = x.contiguous()
        out = torch.empty_like(x)
        _lib.check(self.ctx._lib.bfhip_spline_apply(self.ctx.handle, {'evaluate': 0, 'derivative': 1, 'solve': 2}[mode], x.shape[0],
                                                    self.d, _ptr(x), _ptr(self.off), _ptr(self.knots), _ptr(self.values),
                                                    _ptr(self.coef), _ptr(out)))
        return out
