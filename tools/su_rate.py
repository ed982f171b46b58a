"""Tuning: NUTS on a 64-d quadratic surrogate WITH input scales (4096 chains), folded at upload (default) or kept as a device-side
step (BFHIP_NO_SU_FOLD=1: the sliced kernel's generic instantiation); DECAY=1 adds a decay term in the original space;
BFHIP_NO_PROOF_WEIGHTS=1 keeps the plain norm in the bound / decay proofs.  usage: [BFHIP_NO_SU_FOLD=1] [DECAY=1] python tools/su_rate.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
d, C = 64, 4096
rng = np.random.default_rng(31)
lo, diff = rng.normal(size=d), rng.uniform(0.5, 3., size=d)
spec, _ = correlated_gaussian_spec(d)
spec = dict(spec, su_lo=lo, su_diff=diff)
if os.environ.get('DECAY'):   # a decay term (original space, as Density._set_decay makes it) around the same ellipsoid, never active
    po = spec['poly']
    dinv = 1. / diff
    spec = dict(spec, use_decay=True, decay_mu=lo + diff * po['mu'], decay_hess=po['hess'] * np.outer(dinv, dinv),
                decay_alpha2=(1.5 * po['alpha'])**2, decay_gamma=0.1)
ch = DeviceChains(DeviceDensity(spec, ctx), lo + diff * rng.normal(size=(C, d)), seed=3)
kw = dict(n_warmup=750, check=False)
ch.run(750, 'NUTS', **kw)
ch.run(250, 'NUTS', **kw)
lf0 = ch.total_leapfrog
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(ctx.stream)
for _ in range(3):
    s, st = ch.run(250, 'NUTS', **kw)
e1.record(ctx.stream)
torch.cuda.synchronize()
KN = _lib.last_kernel
print('input scales %s: %.4g leapfrog steps/s, mean tree size %.1f, %s' % ('kept on the device' if os.environ.get('BFHIP_NO_SU_FOLD') else 'folded at upload',
      (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), st[:, :, _lib.NSTATS.index('tree_size')].mean().item(), KN()))
