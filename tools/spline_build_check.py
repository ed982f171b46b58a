"""Tuning: the device spline builder against the host construction on one data set; prints the differences per array.
usage: python3 tools/spline_build_check.py"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.transforms import SIT
from bayesfast_amd.utils.spline import GaussianizingSpline
from bayesfast_amd.device import get_context
ctx = get_context(0)
rng = np.random.default_rng(21)
n, d = 30000, 12
y = rng.laplace(size=(n, d)) * np.linspace(0.5, 3., d)
sit = SIT(n_iter=1, random_generator=1)
sit._weights = np.ones(n) / n
yd = ctx.tensor(y)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    dev = sit._gaussianize(yd).splines
    yT = yd.T.contiguous()
    wn = sit._weights / np.sum(sit._weights)
    w = ctx.tensor(wn)
    neff = 1. / np.sum(wn**2)
    mean = (yT * w).sum(1) / w.sum()
    var = (((yT - mean[:, None])**2) * w).sum(1) / w.sum() / (1. - float(np.sum(wn**2)))
    h = (torch.sqrt(var) * (neff**(-1. / 5)) * sit.bw_factor).contiguous()
    ys = torch.sort(yT, dim=1).values.cpu().numpy()
    host = GaussianizingSpline.build_many(ys, SIT._batch_fun(ctx, yT, w, h, ys), presorted=True, **sit.cubic_options)
for j, (a, b) in enumerate(zip(dev, host)):
    print(j, a.x.size, b.x.size, 'x equal', np.array_equal(a.x, b.x), 'y', np.abs(a.y - b.y).max(), 'k', a._k_left - b._k_left, a._k_right - b._k_right)
    if a.x.size == b.x.size:
        dc = np.abs(a.c - b.c)
        i = np.unravel_index(np.argmax(dc), dc.shape)
        print('   c max diff', dc.max(), 'at', i, 'of', a.c.shape, a.c[i[0]], b.c[i[0]], 'w', np.diff(a.x)[max(i[0] - 1, 0)] if i[0] > 0 else None)
