"""Tuning/verification: the plain kernel's VALU matvec (tail mode) must reproduce the MFMA path bit for bit."""
import sys, os, ctypes as C, time
os.environ.setdefault("BFHIP_NUTS_KERNEL", "sliced")  # the tail path belongs to the sliced kernel
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
L = _lib.lib()
for d in [int(v) for v in os.environ.get("DIMS", "64 32 10").split()]:
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(1).normal(size=(1000, d))
    out = {}
    for tm in (0, 1, 4):
        L.bfhip_debug_tail_max(tm)
        dc = DeviceChains(dens, x0, seed=5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, st = dc.run(60, 'NUTS', n_warmup=30)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[tm] = (s.cpu().numpy(), st.cpu().numpy(), dt)
    L.bfhip_debug_tail_max(4)
    for tm in (1, 4):
        same = np.array_equal(out[0][0], out[tm][0]) and np.array_equal(out[0][1], out[tm][1], equal_nan=True)
        print('d %d tail_max %d: bitwise identical to the MFMA-only run: %s   (%.1f ms vs %.1f ms)' % (d, tm, same, out[tm][2] * 1e3, out[0][2] * 1e3))
