"""Tuning: cycle timeline of workgroup 0's trips in the split kernel, both roles (needs a tools/gvariant.sh build with
-DBF_STRACE=<n>, selected with BFHIP_LIBRARY)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
N = int(os.environ.get("NTRACE", 48))
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
dc = DeviceChains(DeviceDensity(spec, ctx), np.random.default_rng(1).normal(size=(4096, 64)), seed=1)
dc.run(800, 'NUTS', n_warmup=750, layout='split')
buf = torch.zeros((N * 16,), dtype=torch.int64, device='cuda')
L = _lib.lib()
_lib.debug_buffer('gstamps', buf)
dc.run(20, 'NUTS', n_warmup=750, layout='split')
_lib.debug_buffer('gstamps', None)
t = buf.cpu().numpy().reshape(N, 2, 8).astype(np.int64)
print('integrator: top | ->B1 | B1 | tiles | posted | B2      bookkeeper: top | ->B1 | B1 | sums read | machine done | B2 || leaf and merges done | iteration end done (rows, adaptation)   (cycles from the integrator\'s top)')
for i in range(2, N - 1):
    t0 = t[i, 0, 0]
    if t0 == 0 or t[i + 1, 0, 0] == 0: continue
    a = ' '.join('%5d' % int(t[i, 0, k] - t0) for k in range(1, 6))
    b = ' '.join('%5d' % int(t[i, 1, k] - t0) for k in range(0, 6)) + ' || ' + ' '.join('%5d' % (int(t[i, 1, k] - t0) if t[i, 1, k] else 0) for k in (6, 7))
    print('  trip %3d total %6d | I %s | K %s' % (i, int(t[i + 1, 0, 0] - t0), a, b))
