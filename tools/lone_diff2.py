"""Debug: adapted scalars after k iterations, lone vs pipelined kernel."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
d = 64
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(d, fit_scale=1.5)
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(2).normal(size=(70, d))
L = _lib.lib()
_lib.debug_set('no_group', 1)
F = _lib.SC_FIELDS
for k in range(1, 6):
    out = {}
    for lone in (0, 2):
        _lib.debug_set('lone', lone)
        _lib.debug_set('wave_cpg', 16 if lone == 0 else 0)
        dc = DeviceChains(dens, x0, seed=11)
        s1, st1 = dc.run(k, 'NUTS', n_warmup=30, layout='wave')
        out[lone] = (dc.sc.cpu().numpy(), st1.cpu().numpy())
    a, b = out[0][0], out[2][0]
    bad = np.argwhere(a != b)
    print('after', k, 'iterations: differing (chain, field):', [(int(c), F[f]) for c, f in bad[:8]])
    for c, f in bad[:4]:
        print('   chain %d %s pipe %.17g lone %.17g  | accept %.17g' % (c, F[f], a[c, f], b[c, f], out[0][1][c, k - 1, _lib.NSTATS.index('mean_tree_accept')]))
        print('      pipe', {F[i]: float(a[c, i]) for i in range(5)})
        print('      lone', {F[i]: float(b[c, i]) for i in range(5)})
