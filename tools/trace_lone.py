"""Tuning: cycle stamps of the first trips of workgroup 0 of bf_lone_kernel (a -DBF_LTRACE=<n> build: tools/svariant.sh ltrace
-DBF_LTRACE=48, selected with BFHIP_LIBRARY).  Integrator: wait B0 | phase A | wait B1 | matvec (wait B2) | C sums | C rest.
Bookkeeper: wait B0 | bookkeeping (with its two barriers) | verdict.
usage: BFHIP_LIBRARY=bayesfast_amd/variants/libbfhip_s_ltrace.so python tools/trace_lone.py [gauss|funnel] [d] [chains]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import funnel_logp, correlated_gaussian_spec
from bayesfast_amd import _lib
what = sys.argv[1] if len(sys.argv) > 1 else 'gauss'
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Cn = int(sys.argv[3]) if len(sys.argv) > 3 else 256
NT = 48
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
buf = torch.zeros(NT * 32, dtype=torch.int64, device=ctx.device)
L = _lib.lib()
_lib.debug_buffer('stamps_lone', buf)
ch.run(8, 'NUTS', **kw)
torch.cuda.synchronize()
_lib.debug_buffer('stamps_lone', None)
print(_lib.last_kernel())
t = buf.cpu().numpy().reshape(NT, 2, 16).astype(np.float64)
print('trip |  I: total  waitB0      A  waitB1    tile  waitB2       C |  K: total  waitB0    load      dE   wait1 exp+mg0   wait2    rest verdict')
rows = []
for i in range(NT - 1):
    I, K, I2, K2 = t[i, 0], t[i, 1], t[i + 1, 0], t[i + 1, 1]
    if I[0] == 0 or I2[0] == 0:
        continue
    def df(a, b): return (b - a) if (a > 0 and b > 0) else float('nan')
    r = [df(I[0], I2[0]), df(I[0], I[1]), df(I[1], I[2]), df(I[2], I[3]), df(I[3], I[7]), df(I[7], I[4]), df(I[4], I[6]),
         df(K[0], K2[0]), df(K[0], K[1]), df(K[1], K[4]), df(K[4], K[11]), df(K[11], K[10]), df(K[10], K[6]), df(K[6], K[7]), df(K[7], K[2]), df(K[2], K[3])]
    rows.append(r)
    print('%4d | ' % i + ' '.join('%7.0f' % v for v in r[:7]) + ' | ' + ' '.join('%7.0f' % v for v in r[7:]))
rows = np.array(rows)
print('mean | ' + ' '.join('%7.0f' % v for v in np.nanmean(rows[:, :7], 0)) + ' | ' + ' '.join('%7.0f' % v for v in np.nanmean(rows[:, 7:], 0)))
