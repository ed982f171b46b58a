"""Diagnostics: where do the sampler kernel's cycles go? (in-kernel stamps, separate from any timed run)"""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
C_ = 4096
x0 = np.random.default_rng(1).normal(size=(C_, 64))
dc = DeviceChains(DeviceDensity(spec, ctx), x0, seed=1)
SAMPLER = os.environ.get('SAMPLER', 'NUTS')
dc.run(200, SAMPLER, n_warmup=200)
K_ACT = int(os.environ.get('K_ACT', 16))  # chains active per 16-chain workgroup (the rest are parked)
if K_ACT < 16:
    parked = (torch.arange(C_, device='cuda') % 16) >= K_ACT
    dc.sc[parked, _lib.SC_FIELDS.index('i_iter')] = 1e9
buf = torch.zeros((C_ // 16, 16, 20), dtype=torch.int64, device='cuda')
L = _lib.lib()
L.bfhip_debug_stamps.argtypes = [C.c_void_p]
L.bfhip_debug_stamps(C.c_void_p(buf.data_ptr()))
torch.cuda.synchronize(); import time; t0 = time.time()
s, st = dc.run(50, SAMPLER, n_warmup=200)
torch.cuda.synchronize(); dt = time.time() - t0
L.bfhip_debug_stamps(None)
b = buf.cpu().numpy().astype(float)
acc, cnt = b[:, :, :10], b[:, :, 10:]
names = ['A (post x)', 'barrier waits', 'MFMA window', 'C (finish eval)', 'unit INIT', 'unit LEAF', 'unit MERGE_RUN', 'unit DBL_END', 'unit END1-3', 'idle (done)']
tot = acc.sum()
trips = cnt[:, :, 0].mean()
print('launch %.1f ms; leapfrogs %d; trips per wave %.0f; ticks per trip %.0f' % (dt * 1e3, st[:, :, 3].sum().item(), trips, acc.sum(-1).mean() / trips))
act = slice(0, K_ACT)
acc, cnt = acc[:, act], cnt[:, act]
tot = acc.sum()
for k in range(10):
    print('%-18s %5.1f%% of time | %8.0f events/wave | %7.0f ticks/event (mean) ' % (names[k], 100 * acc[:, :, k].sum() / tot, cnt[:, :, k].mean(), acc[:, :, k].sum() / max(cnt[:, :, k].sum(), 1)))
