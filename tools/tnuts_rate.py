"""Tuning: tempered NUTS (bfhip_tnuts_run) on the headline surrogate with a Gaussian base density: tempered leapfrog steps/s.
usage: python tools/tnuts_rate.py [chains]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NI = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = get_context(0)
spec, cov = correlated_gaussian_spec(64)
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(1).normal(size=(C, 64))
ch = DeviceChains(dens, x0, seed=3)
base_cov = 1.3 * cov
ch.run_tempered(120, np.zeros(64), base_cov, n_warmup=100, check=False)
lf0 = ch.total_leapfrog
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(ctx.stream)
out = ch.run_tempered(NI, np.zeros(64), base_cov, n_warmup=100, check=False)
e1.record(ctx.stream)
torch.cuda.synchronize()
st = out[1]
print('tempered NUTS, %d chains: %.4g tempered leapfrog steps/s, %.1f ms per %d iterations, mean tree size %.1f' % (
    C, (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), e0.elapsed_time(e1), NI, st[:, :, _lib.NSTATS.index('tree_size')].mean().item()))
ts = st[:, :, _lib.NSTATS.index('tree_size')].cpu().numpy()
per_chain = ts.sum(1)
print('leapfrogs per chain in the launch: mean %.0f, max %.0f (launch tail %.2f); per workgroup of 8 chains, max over mean: %.2f; tree size '
      'quantiles 50 / 90 / 99 / max: %s' % (per_chain.mean(), per_chain.max(), per_chain.max() / per_chain.mean(),
                                           per_chain[:C // 8 * 8].reshape(-1, 8).max(1).mean() / per_chain.mean(),
                                           np.quantile(ts, [0.5, 0.9, 0.99, 1.0])))
