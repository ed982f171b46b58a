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
dc.run(200, n_warmup=200)
buf = torch.zeros((C_ // 16, 16, 8), dtype=torch.int64, device='cuda')
L = _lib.lib()
L.bfhip_debug_stamps.argtypes = [C.c_void_p]
L.bfhip_debug_stamps(C.c_void_p(buf.data_ptr()))
torch.cuda.synchronize(); import time; t0 = time.time()
s, st = dc.run(50, n_warmup=200)
torch.cuda.synchronize(); dt = time.time() - t0
L.bfhip_debug_stamps(None)
b = buf.cpu().numpy().astype(float)
names = ['A(pre-B1)', 'wait B1', 'B(mfma)', 'wait B2', 'C(eval)', 'unit: eval-post', 'unit: other', 'trips']
tot = b[:, :, :7].sum(-1)
print('launch %.1f ms; leapfrogs %d; trips per wave mean %.0f' % (dt * 1e3, st[:, :, 3].sum().item(), b[:, :, 7].mean()))
print('cycles per wave total (clock64 ticks): mean %.3g' % tot.mean())
for k in range(7):
    print('%-18s %6.1f%%  per trip %8.1f' % (names[k], 100 * b[:, :, k].sum() / tot.sum(), b[:, :, k].sum() / b[:, :, 7].sum()))
print('MFMA waves (0-3) vs others, B phase per trip:', b[:, :4, 2].sum() / b[:, :4, 7].sum(), b[:, 4:, 2].sum() / b[:, 4:, 7].sum())
