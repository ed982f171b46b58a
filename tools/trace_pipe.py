"""Tuning: cycle stamps of the first trips of wave 0, workgroup 0 of bf_nuts_pipe_kernel (a -DBF_TRACE=<n> build of
bfhip_sampler.hip: tools/svariant.sh trace -DBF_TRACE=24 (the stamps live in LDS: 3 KB is what the decay instantiation leaves), selected with BFHIP_LIBRARY) on config 4's funnel (decay
instantiation) or the plain Gaussian: phase A (0-1), barrier B1 (1-2), phase B = MFMA chain + pending bookkeeping (2-5), barrier
B2 (5-6), phase C up to / through its first reduction (6-7-8), rest of phase C (8 - next trip's 0).
usage:  BFHIP_LIBRARY=bayesfast_amd/variants/libbfhip_s_trace.so python tools/trace_pipe.py [funnel|plain] [chains]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.device import get_context
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import funnel_logp, correlated_gaussian_spec
from bayesfast_amd.device import DeviceDensity
from bayesfast_amd import _lib
what = sys.argv[1] if len(sys.argv) > 1 else 'funnel'
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
NT = 24
ctx = get_context(0)
d = 64
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
buf = torch.zeros(NT * 16, dtype=torch.int64, device=ctx.device)
L = _lib.lib()
_lib.debug_buffer('stamps', buf)
ch.run(4, 'NUTS', **kw)
torch.cuda.synchronize()
_lib.debug_buffer('stamps', None)
print(_lib.last_kernel())
t = buf.cpu().numpy().reshape(NT, 16).astype(np.float64)
names = ['A', 'wait B1', 'B: MF0', 'B: ->9', 'B: ->10', 'B: ->11', 'B: ->4', 'B: ->5', 'wait B2', 'C: gather', 'C: sums', 'C: rest']
pairs = [(0, 1), (1, 2), (2, 3), (3, 9), (9, 10), (10, 11), (11, 4), (4, 5), (5, 6), (6, 7), (7, 8)]
print('trip  total  ' + '  '.join('%9s' % n for n in names))
tot = []
for i in range(NT - 1):
    if t[i, 0] == 0 or t[i + 1, 0] == 0:
        continue
    row = [(t[i, b] - t[i, a]) if (t[i, a] > 0 and t[i, b] > 0) else float('nan') for a, b in pairs]
    row.append(t[i + 1, 0] - t[i, 8] if t[i, 8] > 0 else float('nan'))
    tot.append([t[i + 1, 0] - t[i, 0]] + row)
    if i < 24:
        print('%4d %6.0f  ' % (i, t[i + 1, 0] - t[i, 0]) + '  '.join('%9.0f' % v for v in row))
tot = np.array(tot)
print('mean %6.0f  ' % np.nanmean(tot[:, 0]) + '  '.join('%9.0f' % v for v in np.nanmean(tot[:, 1:], 0)))
