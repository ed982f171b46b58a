"""Tuning: cycle stamps of the first trips of wave 0, workgroup 0 of the pipeline-density instantiation of bf_sampler_kernel
(a -DBF_TRACE=64 build of bfhip_sampler.hip, selected with BFHIP_LIBRARY):
build:  tools/svariant.sh trace -DBF_TRACE=64
usage:  BFHIP_LIBRARY=bayesfast_amd/variants/libbfhip_s_trace.so python3 tools/trace_pld.py [m d n_quad]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd import _lib
from pld_rate import spec_of
m = int(sys.argv[1]) if len(sys.argv) > 1 else 457
d = int(sys.argv[2]) if len(sys.argv) > 2 else 27
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 9
NT = 64
ctx = get_context(0)
dd = DeviceDensity(spec_of(m, d, nq), ctx)
x0 = np.random.default_rng(1).normal(size=(4096, d)) * 0.1
ch = DeviceChains(dd, x0, seed=1)
kw = dict(n_warmup=100, check=False)
ch.run(100, 'NUTS', **kw)
buf = torch.zeros(NT * 16, dtype=torch.int64, device=ctx.device)
L = _lib.lib()
_lib.debug_buffer('stamps', buf)
ch.run(8, 'NUTS', **kw)
torch.cuda.synchronize()
_lib.debug_buffer('stamps', None)
t = buf.cpu().numpy().reshape(NT, 16).astype(np.float64)
names = ['A', 'wait B1', 'flags', 'H jobs', 'wait B2', 'P0', 'wait P1', 'gemm1', 'wait P2', 'gemm2', 'wait P3', 'sums+grad', 'rest C', 'unit']
pairs = [(0, 1), (1, 2), (2, 3), (3, 5), (5, 6), (6, 7), (7, 8), (8, 11), (11, 12), (12, 13), (13, 14), (14, 15), (15, 9), (9, 10)]
print('trip  total  ' + '  '.join('%9s' % n for n in names))
tot = []
for i in range(NT - 1):
    if t[i, 0] == 0 or t[i + 1, 0] == 0:
        continue
    row = [(t[i, b] - t[i, a]) if (t[i, a] > 0 and t[i, b] > 0) else float('nan') for a, b in pairs]
    tot.append([t[i + 1, 0] - t[i, 0]] + row)
    if i < 20:
        print('%4d %6.0f  ' % (i, t[i + 1, 0] - t[i, 0]) + '  '.join('%9.0f' % v for v in row))
tot = np.array(tot)
print('mean %6.0f  ' % np.nanmean(tot[:, 0]) + '  '.join('%9.0f' % v for v in np.nanmean(tot[:, 1:], 0)))
print('(s_memtime ticks of 10 ns = 24 shader cycles at 2.4 GHz)')
