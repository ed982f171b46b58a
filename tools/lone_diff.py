"""Debug: first difference between bf_lone_kernel and bf_nuts_pipe_kernel (same chains, same streams)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(d, fit_scale=1.5)
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(2).normal(size=(70, d))
L = _lib.lib()
out = {}
_lib.debug_set('no_group', 1)
for lone in (0, 2):
    _lib.debug_set('lone', lone)
    _lib.debug_set('wave_cpg', 16 if lone == 0 else 0)
    dc = DeviceChains(dens, x0, seed=11)
    s1, st1 = dc.run(45, 'NUTS', n_warmup=nw, layout='wave')
    out[lone] = (s1.cpu().numpy(), st1.cpu().numpy(), dc.sc.cpu().numpy(), dc.vec.cpu().numpy(), dc.rng.cpu().numpy())
a, b = out[0], out[2]
names = _lib.NSTATS
for c in range(70):
    for it in range(45):
        ds = a[1][c, it] != b[1][c, it]
        dq = a[0][c, it] != b[0][c, it]
        if ds.any() or dq.any():
            print('chain', c, 'first differing iteration', it, 'stats differing:', [names[i] for i in np.nonzero(ds)[0]], 'n dims differing', dq.sum())
            for k in range(max(0, it - 1), it + 1):
                print('  it', k, 'pipe', dict(zip(names, a[1][c, k])))
                print('  it', k, 'lone', dict(zip(names, b[1][c, k])))
            print('  max |dq|', np.abs(a[0][c, it] - b[0][c, it]).max())
            break
    if c >= 5:
        break
