"""Tuning: one long launch (warm-up inside the launch, so the chains of a group drift apart in their trees) -- kernel time only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
d = int(os.environ.get('DIM', 64)); Cn = int(os.environ.get('CHAINS', 4096)); N = int(os.environ.get('ITERS', 1500)); NW = int(os.environ.get('NWARM', 500))
spec, _ = correlated_gaussian_spec(d, fit_scale=float(os.environ.get('FIT_SCALE', 1.5)))
dens = DeviceDensity(spec, get_context(0))
for rep in range(3):
    dc = DeviceChains(dens, np.random.default_rng(1).normal(size=(Cn, d)), seed=7 + rep)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    s, st = dc.run(N, 'NUTS', n_warmup=NW, target_accept=float(os.environ.get('TARGET', 0.8)))
    e1.record(); torch.cuda.synchronize(); dt = e0.elapsed_time(e1) * 1e-3
    ts = st[:, :, _lib.NSTATS.index('tree_size')]
    print('%s launch %d: %d iterations %.1f ms, %.3e leapfrog/s, tree size mean %.2f (post-warm-up %.2f, max %d)' % (
        os.environ.get('BFHIP_NUTS_KERNEL', 'pipe'), rep, N, dt * 1e3, float(ts.sum()) / dt, float(ts.mean()), float(ts[:, NW:].mean()), int(ts[:, NW:].max())))
    del s, st
# the same run cut into launches of CHUNK iterations (every launch re-aligns the chains of a group)
for CH in [int(c) for c in os.environ.get('CHUNKS', '').split(',') if c]:
    dc = DeviceChains(dens, np.random.default_rng(1).normal(size=(Cn, d)), seed=7)
    samples = torch.empty((Cn, N, d), dtype=torch.float64, device='cuda'); stats = torch.empty((Cn, N, _lib.STAT_STRIDE if hasattr(_lib, 'STAT_STRIDE') else 11), dtype=torch.float64, device='cuda')
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    tot = 0.
    for i0 in range(0, N, CH):
        s, st = dc.run(min(CH, N - i0), 'NUTS', n_warmup=NW, target_accept=float(os.environ.get('TARGET', 0.8)), check=False)
        tot += float(0)
    e1.record(); torch.cuda.synchronize(); dt = e0.elapsed_time(e1) * 1e-3
    print('%s in launches of %d iterations: %.1f ms' % (os.environ.get('BFHIP_NUTS_KERNEL', 'pipe'), CH, dt * 1e3))
