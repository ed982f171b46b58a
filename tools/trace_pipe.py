"""Tuning: timeline of wave 0's trips in the pipelined NUTS kernel (needs a tools/variant.sh build with -DBF_TRACE=32 (the trace buffer shares the 160 KB of LDS with the tree vectors))."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
N = int(os.environ.get("NTRACE", 32))
K_ACT = int(os.environ.get('K_ACT', 16))
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
C_ = 4096
dc = DeviceChains(DeviceDensity(spec, ctx), np.random.default_rng(1).normal(size=(C_, 64)), seed=1)
dc.run(200, 'NUTS', n_warmup=200)
if K_ACT < 16:
    parked = (torch.arange(C_, device='cuda') % 16) >= K_ACT
    dc.sc[parked, _lib.SC_FIELDS.index('i_iter')] = 1e9
buf = torch.zeros((N * 16,), dtype=torch.int64, device='cuda')
L = _lib.lib()
L.bfhip_debug_stamps.argtypes = [C.c_void_p]
L.bfhip_debug_stamps(C.c_void_p(buf.data_ptr()))
dc.run(20, 'NUTS', n_warmup=200)
L.bfhip_debug_stamps(None)
t = buf.cpu().numpy().reshape(N, 16).astype(np.int64)
print('points: 1 A done | 2 after B1 | 3 MFMAs issued | 4 bookkeeping done | 5 GB written | 6 after B2 | 7 partials | 8 reduced | '
      '9 leaf logic done | 10 merges done | 11 tree ended | 12 iteration end done (before the momentum draw)')
for i in range(4, min(N - 1, 4 + int(os.environ.get("RAW", 30)))):
    tt = t[i]
    if tt[0] == 0: continue
    rel = [(int(tt[k] - tt[0]) if tt[k] else 0) for k in range(13)]
    print('  trip %3d total %6d | ' % (i, int(t[i + 1][0] - tt[0]) if t[i + 1][0] else -1) + ' '.join('%5d' % r for r in rel[1:]))
