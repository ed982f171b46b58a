"""Rates of the generic (non-plain) sampler instantiation on the headline shapes: decay on, constraint transforms on."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
d, C = 64, 4096
base, cov = correlated_gaussian_spec(d)
variants = {'plain': dict(base)}
dec = dict(base); dec.update(use_decay=True, decay_mu=base['poly']['mu'], decay_hess=base['poly']['hess'], decay_alpha2=float(base['poly']['alpha'])**2 * 2.25, decay_gamma=0.1)
variants['decay'] = dec
tr = dict(base); tr.update(ranges=np.stack([np.full(d, -12.), np.full(d, 12.)], 1), hard_bounds=np.ones((d, 2), dtype=np.uint8))
variants['hard_bounds (logistic transform)'] = tr
both = dict(dec); both.update(ranges=tr['ranges'], hard_bounds=tr['hard_bounds'])
variants['decay + hard_bounds'] = both
for name, spec in variants.items():
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(1).normal(size=(C, d)) * 0.1
    dc = DeviceChains(dens, x0, seed=3)
    for _ in range(3):
        dc.run(100, 'NUTS', n_warmup=300, check=False)
    ts = []
    s = st = None
    for _ in range(3):  # HIP events around the launch; output arrays reused
        l0 = dc.total_leapfrog
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        s, st = dc.run(100, 'NUTS', n_warmup=300, check=False, samples=s, stats=st)
        e1.record(); torch.cuda.synchronize(); dt = e0.elapsed_time(e1) * 1e-3
        ts.append(((dc.total_leapfrog - l0) / dt, dt * 1e3))
    dc.raise_on_error()
    r = np.array(ts)
    print('%-34s %.3e leapfrog/s, %.1f ms per 100 iterations, mean tree size %.1f' % (name, r[:, 0].mean(), r[:, 1].mean(), st[:, :, _lib.NSTATS.index('tree_size')].mean().item()), flush=True)
