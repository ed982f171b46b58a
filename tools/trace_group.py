"""Tuning: cycle timeline of workgroup 0's trips in the group kernel (needs a tools/gvariant.sh build with -DBF_GTRACE=<n>,
selected with BFHIP_LIBRARY)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
N = int(os.environ.get("NTRACE", 64))
ctx = get_context(0)
HET = int(os.environ.get('HETERO', 0))
spec, _ = correlated_gaussian_spec(64, scales=np.logspace(-0.5, 0.5, 64) if HET else None)
RKW = dict(adapt_metric=False, target_accept=0.9) if HET else {}
C_ = int(os.environ.get('CHAINS', 4096))
dc = DeviceChains(DeviceDensity(spec, ctx), np.random.default_rng(1).normal(size=(C_, 64)), seed=1)
dc.run(800, 'NUTS', n_warmup=750, **RKW)
buf = torch.zeros((N * 8,), dtype=torch.int64, device='cuda')
L = _lib.lib()
_lib.debug_buffer('gstamps', buf)
dc.run(20, 'NUTS', n_warmup=750, **RKW)
_lib.debug_buffer('gstamps', None)
t = buf.cpu().numpy().reshape(N, 8).astype(np.int64)
LBL = os.environ.get("GLABELS", "1 phase A done | 2 after B1 | 3 MFMAs + tile sums | 4 eval sums posted | 5 U-turn sums posted | 6 after B2 | 7 eval scalars | (next 0) state machine"); print("points:", LBL)
for i in range(2, N - 1):
    tt = t[i]
    if tt[0] == 0 or t[i + 1][0] == 0: continue
    rel = [int(tt[k] - tt[0]) for k in range(8)]
    print('  trip %3d total %6d | ' % (i, int(t[i + 1][0] - tt[0])) + ' '.join('%5d' % r for r in rel[1:]) + ' | state machine %5d' % int(t[i + 1][0] - tt[7]))
