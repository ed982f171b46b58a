"""Tuning: NUTS on the plain Gaussian surrogate (config 2's family) or the funnel with the decay term, few chains -- the latency kernel
(bfhip_lone.h) against the pipelined kernel's few-chain instantiation (BFHIP_LONE=0).
usage: [BFHIP_LONE=0] python tools/lone_rate.py [gauss|funnel] [d] [chains] [iters]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import funnel_logp, correlated_gaussian_spec
from bayesfast_amd import _lib
what = sys.argv[1] if len(sys.argv) > 1 else 'gauss'
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
Cn = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
n_it = int(sys.argv[4]) if len(sys.argv) > 4 else 200
ctx = get_context(0)
rng = np.random.default_rng(2024)
if what == 'funnel':
    logp = funnel_logp(d)
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
    den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
    x_fit = rng.normal(size=(2 * su.n_param, d))
    den.fit(x_fit, logp(x_fit))
    x0 = x_fit[rng.integers(0, x_fit.shape[0], Cn)] * 0.5
    dd = den.device(ctx)
    kw = dict(n_warmup=300, check=False, target_accept=0.95, layout='wave')
else:
    spec, _ = correlated_gaussian_spec(d)
    dd = DeviceDensity(spec, ctx)
    x0 = rng.normal(size=(Cn, d))
    kw = dict(n_warmup=300, check=False, layout='wave')
ch = DeviceChains(dd, x0, seed=5)
ch.run(300, 'NUTS', **kw)
L = _lib.lib()
res = []
for rep in range(3):
    lf0 = ch.total_leapfrog
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s, st = ch.run(n_it, 'NUTS', **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ts = st[:, :, _lib.NSTATS.index('tree_size')].cpu().numpy().sum(1)
    res.append(((ch.total_leapfrog - lf0) / dt, dt * 1e6 / ts.max(), ts.mean() / n_it))
r = np.array(res)
print('%s d=%d chains=%d BFHIP_LONE=%s kernel %s: %.4g lf/s, %.2f us per leapfrog of the busiest chain, mean tree %.1f'
      % (what, d, Cn, os.environ.get('BFHIP_LONE'), _lib.last_kernel(), r[:, 0].max(), r[:, 1].min(), r[:, 2].mean()))
