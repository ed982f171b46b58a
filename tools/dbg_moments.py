"""Exploration: posterior moments and the share of samples outside the bound ellipsoid for a workload spec."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
d = 64
spec, cov = correlated_gaussian_spec(d)
ctx = get_context()
def report(name, draws):
    emp = np.cov(draws, rowvar=False)
    r = np.diag(emp) / np.diag(cov)
    ev = np.linalg.eigvalsh(np.linalg.solve(np.linalg.cholesky(cov), np.linalg.solve(np.linalg.cholesky(cov), emp).T))
    print('%-28s var ratio min %.3f median %.3f max %.3f | whitened eig min %.3f max %.3f | mean/sd max %.3f' % (name, r.min(), np.median(r), r.max(), ev.min(), ev.max(), np.abs(draws.mean(0) / np.sqrt(np.diag(cov))).max()))
for ub in (True,):
    sp = dict(spec); sp['poly'] = dict(spec['poly'], use_bound=ub)
    dc = DeviceChains(DeviceDensity(sp, ctx), np.random.default_rng(1).normal(size=(4096, d)), seed=3)
    dc.run(500, 'NUTS', n_warmup=500)
    s, st = dc.run(400, 'NUTS', n_warmup=500)
    s = s.cpu().numpy()
    report('spec use_bound=%s' % ub, s.reshape(-1, d))
    report('  (last iteration only)', s[:, -1])
    mu, H, al = spec['poly']['mu'], spec['poly']['hess'], spec['poly']['alpha']
    b = np.sqrt(np.einsum('ni,ij,nj->n', s[:, -1] - mu, H, s[:, -1] - mu))
    print('   beta of last samples: mean %.2f max %.2f ; alpha %.2f ; frac outside %.4f' % (b.mean(), b.max(), al, (b > al).mean()))
