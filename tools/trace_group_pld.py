"""Tuning: cycle timeline of workgroup 0's trips in the group kernel's PIPELINE form (bf_group_kernel<W, true, 8 | 12>); needs a
build with -DBF_PTRACE=<n> (tools/gvariant.sh ptrace -DBF_PTRACE=64 -UBF_ONLY_HEADLINE is not enough: the variant script compiles
the headline instantiation only -- build with:  tools/gvariant.sh ptrace -DBF_PTRACE=64 -DBF_PTRACE_ALL), selected with BFHIP_LIBRARY.
usage:  BFHIP_LIBRARY=bayesfast_amd/variants/libbfhip_ptrace.so python3 tools/trace_group_pld.py [m d n_quad]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import random_pipeline_spec
from bayesfast_amd import _lib
m = int(sys.argv[1]) if len(sys.argv) > 1 else 457
d = int(sys.argv[2]) if len(sys.argv) > 2 else 27
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 9
N = 64
ctx = get_context(0)
dc = DeviceChains(DeviceDensity(random_pipeline_spec(m, d, nq), ctx), np.random.default_rng(1).normal(size=(4096, d)) * 0.1, seed=1)
kw = dict(n_warmup=100, check=False, layout='group')
dc.run(100, 'NUTS', **kw)
buf = torch.zeros((N * 16,), dtype=torch.int64, device='cuda')
_lib.debug_buffer('gstamps', buf)
dc.run(10, 'NUTS', **kw)
torch.cuda.synchronize()
_lib.debug_buffer('gstamps', None)
print(_lib.last_kernel())
t = buf.cpu().numpy().reshape(N, 16).astype(np.float64)
names = ['A', 'wait B1', 'H tiles', 'posts', 'wait B2', 'beta+XE', 'wait P0', 'monomials', 'wait P1', 'gemm1', 'wait P2', 'gemm2', 'wait P3',
         'gather+grad', 'late exch', 'state m.']
print('trip  total ' + ' '.join('%9s' % n for n in names))
rows = []
for i in range(2, N - 1):
    if t[i, 0] == 0 or t[i + 1, 0] == 0:
        continue
    seg = [t[i, k + 1] - t[i, k] for k in range(15)] + [t[i + 1, 0] - t[i, 15]]
    rows.append([t[i + 1, 0] - t[i, 0]] + seg)
    if len(rows) <= 16:
        print('%4d %6.0f ' % (i, rows[-1][0]) + ' '.join('%9.0f' % v for v in seg))
rows = np.array(rows)
print('mean %6.0f ' % rows[:, 0].mean() + ' '.join('%9.0f' % v for v in rows[:, 1:].mean(0)))
