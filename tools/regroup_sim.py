"""What regrouping chains by their tree sizes would buy on the heterogeneous workload of bench.py: a trip of the lane-per-chain
layouts costs the same for 1 or 16 active chains, so a group's iteration costs max(tree sizes); efficiency = sum of tree sizes /
(16 x sum of the groups' maxima), for the chains in index order and sorted by their mean tree size of the launch before."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
import bench
ctx = get_context(0)
d, C = 64, 4096
spec, _ = correlated_gaussian_spec(d, scales=np.logspace(-0.5, 0.5, d))
ch = DeviceChains(DeviceDensity(spec, ctx), np.random.default_rng(2025).normal(size=(C, d)), seed=2025)
kw = dict(n_warmup=bench.N_ADAPT, check=False, adapt_metric=False, target_accept=0.9, layout='wave')
ch.run(bench.N_ADAPT, 'NUTS', **kw)
ts = []
for _ in range(2):
    s, st = ch.run(250, 'NUTS', **kw)
    ts.append(st[:, :, _lib.NSTATS.index('tree_size')].cpu().numpy())
prev, cur = ts
def eff(t, order):
    g = t[order].reshape(C // 16, 16, -1)
    return t.sum() / (16 * g.max(1).sum())
ident = np.arange(C)
by_prev = np.argsort(prev.mean(1), kind='stable')
by_self = np.argsort(cur.mean(1), kind='stable')
own = np.array([np.mean(cur[i] == np.bincount(cur[i].astype(int)).argmax()) for i in range(C)])
print('mean tree size %.1f; a chain has its own most common size in %.0f %% of its iterations' % (cur.mean(), 100 * own.mean()))
print('efficiency of 16-chain groups: index order %.3f, sorted by the previous launch\'s mean tree size %.3f, by this launch\'s (oracle) %.3f'
      % (eff(cur, ident), eff(cur, by_prev), eff(cur, by_self)))
