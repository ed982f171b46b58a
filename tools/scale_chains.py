"""Tuning: launch time of 100 NUTS iterations against the number of chains (d = 128 by default)."""
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
for d in (128,):
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    for C in (8, 128, 1024, 2048, 4096):
        dc = DeviceChains(dens, np.random.default_rng(1).normal(size=(C, d)), seed=3)
        for _ in range(3): dc.run(100, 'NUTS', n_warmup=300, check=False)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        s, st = dc.run(100, 'NUTS', n_warmup=300, check=False)
        e1.record(); torch.cuda.synchronize()
        ts = st[:, :, _lib.NSTATS.index('tree_size')]
        print('d %d chains %5d: %.2f ms per 100 iterations, tree size mean %.2f max %d, step %.4f' % (d, C, e0.elapsed_time(e1), float(ts.mean()), int(ts.max()), float(st[:, -1, _lib.NSTATS.index('step_size')].mean())), flush=True)
