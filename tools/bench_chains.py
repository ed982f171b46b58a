"""Leapfrog rate of the fused sampler against the number of chains per GPU (64-d headline surrogate)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
dens = DeviceDensity(spec, ctx)
for C in (256, 1024, 4096, 8192, 16384, 32768):
    x0 = np.random.default_rng(1).normal(size=(C, 64))
    dc = DeviceChains(dens, x0, seed=3)
    smp, st = ctx.empty((C, 100, 64)), ctx.empty((C, 100, _lib.STAT_STRIDE))
    for _ in range(3):
        dc.run(100, 'NUTS', n_warmup=300, check=False, samples=smp, stats=st)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); l0 = dc.total_leapfrog
        dc.run(100, 'NUTS', n_warmup=300, check=False, samples=smp, stats=st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ts.append(((dc.total_leapfrog - l0) / dt, dt * 1e3))
    dc.raise_on_error()
    r = np.array(ts)
    print('%6d chains: %.3e leapfrog/s, %.1f ms per 100 iterations' % (C, r[:, 0].mean(), r[:, 1].mean()), flush=True)
    del dc, smp, st
