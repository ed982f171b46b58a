"""Tuning: cycle stamps of the first trips of wave 0, workgroup 0 of bf_sampler_kernel (a -DBF_TRACE=<n> build of
bfhip_sampler.hip, selected with BFHIP_LIBRARY) on config 5's shard: phase A (0-1), barrier B1 (1-2), MFMA jobs (3-5), barrier
B2 (5-6), phase C up to its reductions (6-7-8), rest of phase C (8-9), the tree unit (9-10).
build:  cd bayesfast_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DBF_TRACE=64 -c bfhip_sampler.hip
        -o _obj/bfhip_sampler_trace.o && hipcc -shared ... (tools/gvariant.sh shows the link line)
usage:  BFHIP_LIBRARY=bayesfast_amd/variants/libbfhip_strace.so python tools/trace_sliced.py [cubic 0|1]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.device import get_context
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import planck_like_logp
from bayesfast_amd import _lib
cubic = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NT = 24
ctx = get_context(0)
d, Cn = 128, 1024
rng = np.random.default_rng(2024)
logp, chol = planck_like_logp(d)
m16 = np.arange(16)
cfgs = [bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic')]
if cubic:
    cfgs += [bfa.PolyConfig('cubic-2', input_mask=m16), bfa.PolyConfig('cubic-3', input_mask=m16)]
su = bfa.PolyModel(cfgs, input_size=d, output_size=1)
den = bfa.SurrogateDensity(su)
x_fit = rng.normal(size=(2 * 9201, d)) @ chol.T
den.fit(x_fit, logp(x_fit))
x0 = x_fit[rng.integers(0, x_fit.shape[0], Cn)] * 0.5
ch = DeviceChains(den.device(ctx), x0, seed=5)
kw = dict(n_warmup=100, check=False, max_treedepth=6)
ch.run(100, 'NUTS', **kw)
buf = torch.zeros(NT * 16, dtype=torch.int64, device=ctx.device)
L = _lib.lib()
_lib.debug_buffer('stamps', buf)
ch.run(4, 'NUTS', **kw)
torch.cuda.synchronize()
_lib.debug_buffer('stamps', None)
t = buf.cpu().numpy().reshape(NT, 16).astype(np.float64)
names = ['A', 'wait B1', 'flags', 'jobs', 'jobs->5', 'wait B2', 'C: gather', 'C: sums', 'C: rest', 'unit']
pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10)]
print('trip  total  ' + '  '.join('%9s' % n for n in names))
tot = []
for i in range(NT - 1):
    if t[i, 0] == 0 or t[i + 1, 0] == 0:
        continue
    row = [(t[i, b] - t[i, a]) if (t[i, a] > 0 and t[i, b] > 0) else float('nan') for a, b in pairs]
    tot.append([t[i + 1, 0] - t[i, 0]] + row)
    if i < 24:
        print('%4d %6.0f  ' % (i, t[i + 1, 0] - t[i, 0]) + '  '.join('%9.0f' % v for v in row))
tot = np.array(tot)
print('mean %6.0f  ' % np.nanmean(tot[:, 0]) + '  '.join('%9.0f' % v for v in np.nanmean(tot[:, 1:], 0)))
print('(s_memtime ticks of 10 ns = 24 shader cycles at 2.4 GHz)')

if os.environ.get('CUBIC_STAMPS'):   # a -DBF_TRACE_CUBIC build: stamps 11 (masked inputs gathered), 12 (cubic-2 done), 13 (cubic-3 done)
    ok = (t[:, 11] > 0) & (t[:, 13] > 0)
    print('cubic_lds, mean cycles over %d trips: cubic-2 %.0f, cubic-3 %.0f' % (ok.sum(), np.mean((t[:, 12] - t[:, 11])[ok]), np.mean((t[:, 13] - t[:, 12])[ok])))
