import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.device import get_context
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import funnel_logp
ctx = get_context(0)
d, Cn = 64, 256
rng = np.random.default_rng(2024)
logp = funnel_logp(d)
su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
x_fit = rng.normal(size=(2 * su.n_param, d))
den.fit(x_fit, logp(x_fit))
x0 = x_fit[rng.integers(0, x_fit.shape[0], Cn)] * 0.5
ch = DeviceChains(den.device(ctx), x0, seed=5)
kw = dict(n_warmup=300, check=False, target_accept=0.95, layout='wave')
ch.run(300, 'NUTS', **kw)
lf0 = ch.total_leapfrog
torch.cuda.synchronize(); t0 = time.perf_counter()
s, st = ch.run(200, 'NUTS', **kw)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
from bayesfast_amd import _lib
ts = st[:, :, _lib.NSTATS.index('tree_size')].cpu().numpy().sum(1)
print('NO_QUAD', os.environ.get('BFHIP_NO_QUAD'), 'chains', Cn, '%.3g lf/s' % ((ch.total_leapfrog - lf0) / dt), 'us per leapfrog of the busiest chain %.2f' % (dt * 1e6 / ts.max()))
