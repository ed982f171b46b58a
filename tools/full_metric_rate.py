"""Tuning: leapfrog steps/s of NUTS with the full-rank metric on the headline surrogate, after adaptation (a fixed per-chain
covariance) and during it.  usage: python tools/full_metric_rate.py [chains]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(1).normal(size=(C, 64))
for label, kw, nwarm in (('adapting (warm-up)', dict(metric='full'), 10**6), ('fixed (after warm-up)', dict(metric='full'), 300)):
    ch = DeviceChains(dens, x0, seed=3, **kw)
    ch.run(300, 'NUTS', n_warmup=nwarm, check=False)
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    s, st = ch.run(100, 'NUTS', n_warmup=nwarm, check=False)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    ts = st[:, :, _lib.NSTATS.index('tree_size')].mean().item()
    print('full-rank metric, %s, %d chains: %.4g leapfrog steps/s, mean tree size %.1f, %.1f ms per 100 iterations (%s)' % (
        label, C, (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), ts, e0.elapsed_time(e1),
        ''))
