"""Tuning: kernel time of launches of n iterations (HIP events) -> the fixed cost of a launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
dc = DeviceChains(DeviceDensity(spec, ctx), np.random.default_rng(1).normal(size=(4096, 64)), seed=3)
dc.run(750, 'NUTS', n_warmup=750, check=False)
for n in (1, 2, 5, 10, 25, 50, 100, 250, 500):
    s = st = None
    best = 1e9
    for rep in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        s, st = dc.run(n, 'NUTS', n_warmup=750, check=False, samples=s, stats=st, launch_iters=None)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print('%4d iterations: %.3f ms  (%.1f us per iteration)' % (n, best, best * 1e3 / n), flush=True)
