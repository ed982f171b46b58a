"""Tuning: the in-step layouts ('group', 'split') and 'wave' on the headline surrogate at a given target_accept (tree size) and
chain count: post-adaptation launches, leapfrog steps/s.  usage: python tools/layout_ab.py [target_accept] [chains] [dim]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ta = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(D, fit_scale=float(os.environ.get('FIT_SCALE', 1.5)))   # (FIT_SCALE < 1.3: chains leak through the bound)
if os.environ.get('DECAY'):   # the decay term of core/density.py:740-746 around the bound's ellipsoid, never active here
    po = spec['poly']
    spec = dict(spec, use_decay=True, decay_mu=po['mu'], decay_hess=po['hess'], decay_alpha2=(float(os.environ['DECAY']) * po['alpha'])**2, decay_gamma=0.1)
if os.environ.get('BOUNDED'):  # behind the constraint transform: all four kinds of bounds (density.py:92-140)
    lo = np.full(D, -9.) + np.arange(D) * 0.01
    spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * (D // 4), dtype=np.uint8))
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(1).normal(size=(C, D)) * (0.3 if os.environ.get('BOUNDED') else 1.)
KN = _lib.last_kernel
for layout in ('group', 'split', 'wave', 'auto'):
    ch = DeviceChains(dens, x0, seed=3)
    kw = dict(n_warmup=750, target_accept=ta, check=False, layout=layout)
    ch.run(750, 'NUTS', **kw)
    s, st = ch.run(250, 'NUTS', **kw)
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    for _ in range(4):
        s, st = ch.run(250, 'NUTS', **kw)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    ts = st[:, :, _lib.NSTATS.index('tree_size')].mean().item()
    gc = torch.zeros(2, dtype=torch.int64, device='cuda')
    print('d %d target_accept %.2f chains %d layout %-5s (%s): %.4g leapfrog steps/s, mean tree size %.1f' % (
        D, ta, C, layout, KN(), (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), ts))
