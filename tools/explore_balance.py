"""Exploration: how much of the NUTS launch time is load imbalance between chains?  Compares the NUTS rate with
the HMC rate (fixed work per chain) on the bench workload and prints the per-chain work distribution."""
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
ch = DeviceChains(dens, x0, seed=2024)
for k in range(3):
    ch.run(100, 'NUTS', n_warmup=300, check=False)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); lf0 = ch.total_leapfrog
    s, st = ch.run(100, 'NUTS', n_warmup=300, check=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('NUTS launch: %.2f ms, %.3e lf/s' % (dt * 1e3, (ch.total_leapfrog - lf0) / dt))
ts = st.cpu().numpy()[:, :, _lib.NSTATS.index('tree_size')]
td = st.cpu().numpy()[:, :, _lib.NSTATS.index('tree_depth')]
tot = ts.sum(1)
# trips ~ leapfrogs + 1 (init) + merge/end units; use leapfrogs + 1 + depth as a proxy
trips = (ts + 1 + td).sum(1)
print('per-chain leapfrogs per 100 iterations: mean %.0f, std %.0f, min %d, median %d, p90 %d, p99 %d, max %d'
      % (tot.mean(), tot.std(), tot.min(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
g = tot.reshape(-1, 16)
print('group-of-16 max: mean %.0f, max %.0f; ratio mean(group max)/mean = %.2f, max/mean = %.2f'
      % (g.max(1).mean(), g.max(1).max(), g.max(1).mean() / tot.mean(), tot.max() / tot.mean()))
e = np.exp(ch.field('log_bar').cpu().numpy())
print('step size: mean %.4f std %.4f; corr(log eps, leapfrogs) = %.2f' % (e.mean(), e.std(), np.corrcoef(np.log(e), tot)[0, 1]))
for depth in range(2, 8):
    print('  depth %d: %.1f%% of iterations' % (depth, 100 * (td == depth).mean()))
# balanced work: HMC with a fixed number of leapfrogs
for n_int in (16, 32):
    ch2 = DeviceChains(dens, x0, seed=7)
    ch2.run(100, 'HMC', n_warmup=100, n_int_step=n_int, check=False)
    for k in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); lf0 = ch2.total_leapfrog
        ch2.run(100, 'HMC', n_warmup=100, n_int_step=n_int, check=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('HMC n_int_step %d launch: %.2f ms, %.3e lf/s' % (n_int, dt * 1e3, (ch2.total_leapfrog - lf0) / dt))

# trip time against the number of chains active in a 16-chain workgroup: park the others (i_iter beyond iter_end)
for k_act in (1, 2, 4, 8, 12, 16):
    ch3 = DeviceChains(dens, x0, seed=9)
    ch3.run(20, 'HMC', n_warmup=20, n_int_step=32, check=False)
    parked = (torch.arange(C, device=ch3.sc.device) % 16) >= k_act
    ch3.sc[parked, _lib.SC_FIELDS.index('i_iter')] = 1e9
    torch.cuda.synchronize(); t0 = time.perf_counter(); lf0 = ch3.total_leapfrog
    ch3.run(50, 'HMC', n_warmup=20, n_int_step=32, check=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nl = ch3.total_leapfrog - lf0
    print('active %2d/16: %.2f ms, %.3e lf/s, %.2f us per trip' % (k_act, dt * 1e3, nl / dt, dt * 1e6 / (50 * 33)))
