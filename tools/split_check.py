import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
dens = DeviceDensity(spec, ctx)
rng = np.random.default_rng(3)
def run(layout, x0, n, nw, **kw):
    ch = DeviceChains(dens, x0, seed=5)
    s, st = ch.run(n, 'NUTS', n_warmup=nw, layout=layout, **kw)
    torch.cuda.synchronize()
    return s.cpu().numpy(), st.cpu().numpy(), ch.sc.cpu().numpy(), ch.vec.cpu().numpy(), ch.rng.cpu().numpy(), ch.total_leapfrog
for name, x0, n, nw, kw in (('small', rng.normal(size=(37, 64)), 60, 40, {}), ('far', rng.normal(size=(20, 64)) * 6., 30, 20, {}),
                            ('big', rng.normal(size=(4096, 64)), 300, 200, {}), ('cut', rng.normal(size=(100, 64)), 90, 50, dict(launch_iters=17))):
    a = run('group', x0, n, nw, **kw)
    b = run('split', x0, n, nw, **kw)
    kname = _lib.last_kernel
    print(name, kname(), [bool(np.array_equal(u, v, equal_nan=True)) if isinstance(u, np.ndarray) else u == v for u, v in zip(a, b)], a[1][:, :, 3].mean())
