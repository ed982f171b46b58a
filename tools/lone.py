"""Tuning: per-leapfrog latency of the production NUTS kernel with K_ACT chains active per 16-chain workgroup."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bayesfast_amd.device import DeviceContext, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib

d, C = 64, 4096
spec, cov = correlated_gaussian_spec(d)
ctx = DeviceContext(0)
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(2024).normal(size=(C, d))
SAMPLER = os.environ.get('SAMPLER', 'NUTS')
for k_act in [int(v) for v in os.environ.get('K_ACTS', '1 4 16').split()]:
    ch = DeviceChains(dens, x0, seed=2024)
    ch.run(300, SAMPLER, n_warmup=300, check=False)
    if k_act < 16:
        parked = (torch.arange(C, device=ch.sc.device) % 16) >= k_act
        ch.sc[parked, _lib.SC_FIELDS.index('i_iter')] = 1e9
    ts_i = _lib.NSTATS.index('tree_size')
    res = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, st = ch.run(100, SAMPLER, n_warmup=300, check=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ts = st[:, :, ts_i].cpu().numpy() if SAMPLER == 'NUTS' else np.full((C, 100), 32.)
        act = (np.arange(C) % 16) < k_act
        tot = ts[act].sum(1)
        res.append((dt * 1e3, tot.sum() / dt, dt * 1e6 / tot.max(), tot.max(), tot.mean()))
    r = np.array(res)
    print('active %2d/16: %.2f ms, %.3e lf/s, %.2f us per leapfrog of the slowest chain (max %d, mean %.0f leapfrogs)'
          % (k_act, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].mean()))
