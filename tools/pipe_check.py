"""Tuning / validation: the pipelined NUTS kernel against bf_sampler_kernel on the same chains (bitwise) and their launch times."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
L = _lib.lib()
L.bfhip_debug_no_pipe.argtypes = [C.c_int]
d = int(os.environ.get('DIM', 64)); Cn = int(os.environ.get('CHAINS', 4096)); FS = float(os.environ.get('FIT_SCALE', 1.5))
spec, _ = correlated_gaussian_spec(d, fit_scale=FS)
dens = DeviceDensity(spec, ctx)
out = {}
for name, flag in (('base', 1), ('pipe', 0)):
    L.bfhip_debug_no_pipe(flag)
    dc = DeviceChains(dens, np.random.default_rng(1).normal(size=(Cn, d)), seed=7)
    s1, st1 = dc.run(int(os.environ.get('N1', 150)), 'NUTS', n_warmup=int(os.environ.get('NWARM', 100)))
    NW = int(os.environ.get('NWARM', 100))
    s2, st2 = dc.run(100, 'NUTS', n_warmup=NW)
    s1 = torch.cat([s1, s2], 1); st1 = torch.cat([st1, st2], 1)
    del s2, st2
    best = None
    for rep in range(int(os.environ.get('REPS', 4))):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        s2, st2 = dc.run(100, 'NUTS', n_warmup=NW)
        e1.record(); torch.cuda.synchronize(); dt = e0.elapsed_time(e1) * 1e-3
        lf = float(st2[:, :, _lib.NSTATS.index('tree_size')].sum())
        print('   %s launch %d: %.2f ms, %.3e leapfrog/s' % (name, rep, dt * 1e3, lf / dt))
        if rep == 0:
            s1 = torch.cat([s1, s2], 1); st1 = torch.cat([st1, st2], 1)
    s2, st2 = s1[:, :0], st1[:, :0]
    out[name] = (torch.cat([s1, s2], 1).cpu().numpy(), torch.cat([st1, st2], 1).cpu().numpy())
L.bfhip_debug_no_pipe(0)
a, b = out['pipe'], out['base']
print('samples bitwise equal:', np.array_equal(a[0], b[0]), ' stats bitwise equal:', np.array_equal(a[1], b[1], equal_nan=True))
if not np.array_equal(a[0], b[0]):
    bad = np.argwhere((a[0] != b[0]).any(-1))
    print('first differing (chain, iteration):', bad[:5].tolist(), 'of', len(bad), 'max abs diff', np.abs(a[0] - b[0]).max())
    c, i = bad[0]
    print('stats pipe', a[1][c, i]); print('stats base', b[1][c, i])
if not np.array_equal(a[1], b[1], equal_nan=True):
    neq = ~((a[1] == b[1]) | (np.isnan(a[1]) & np.isnan(b[1])))
    it = np.argwhere(neq.any(-1))
    first = it[np.argsort(it[:, 1], kind='stable')][:6]
    print('earliest differing stats (chain, iteration):', first.tolist())
    for c, i in first[:3]:
        print(' chain %d it %d fields' % (c, i), [(_lib.NSTATS[k], a[1][c, i, k], b[1][c, i, k]) for k in np.argwhere(neq[c, i]).ravel()])
        print('   base dirs: depth %s size %s' % (b[1][c, max(i-1,0):i+1, 2], b[1][c, max(i-1,0):i+1, 3]))
