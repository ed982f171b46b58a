"""Tuning: time of bfhip_polar_ns (FastICA's symmetric decorrelation on the device) and of the other pieces of a FastICA
iteration at SIT's sizes.  usage: python3 tools/polar_rate.py [d] [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, _ptr
from bayesfast_amd import _lib
from bayesfast_amd.transforms import ica
d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ctx = get_context(0)
rng = np.random.default_rng(0)
A = ctx.tensor(rng.normal(size=(d, d)) * 0.01)
X = torch.empty_like(A)
work = torch.empty(2 * d * d + 80, dtype=torch.float64, device=A.device)


def timed(f, reps=20):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for it in (0, 10, 40):
    t = timed(lambda: _lib.check(ctx._lib.bfhip_polar_ns(ctx.handle, d, _ptr(A), _ptr(X), it, _ptr(work), _ptr(work[-1:]))))
    print('bfhip_polar_ns d=%d n_iter=%d: %.1f us, resid %.2e' % (d, it, t, float(work[-1])))
# phase stamps of workgroup 0 (100 MHz clock): start, norms, then per step: T tile, workgroup barrier, update, grid barrier
st = torch.zeros(64, dtype=torch.int64, device=A.device)
_lib.check(ctx._lib.bfhip_debug_buffer(b'gstamps', _ptr(st)))
_lib.check(ctx._lib.bfhip_polar_ns(ctx.handle, d, _ptr(A), _ptr(X), 5, _ptr(work), _ptr(work[-1:])))
torch.cuda.synchronize()
_lib.check(ctx._lib.bfhip_debug_buffer(b'gstamps', None))
t = st.cpu().numpy()
t = t[t > 0]
print('stamps (us since start):', np.round((t - t[0]) / 100., 2))
u, s, vt = np.linalg.svd(A.cpu().numpy())
print('error against the SVD polar factor: %.2e' % np.abs(X.cpu().numpy() - u @ vt).max())
x1 = ctx.tensor(rng.normal(size=(n, d)))
W = ctx.tensor(np.linalg.qr(rng.normal(size=(d, d)))[0])
print('x1 @ W.T: %.1f us' % timed(lambda: x1 @ W.T))
g = torch.tanh(x1 @ W.T)
print('tanh: %.1f us' % timed(lambda: torch.tanh(g)))
print('1 - g*g mean: %.1f us' % timed(lambda: (1. - g * g).mean(0)))
print('tn_product: %.1f us' % timed(lambda: ica._tn_product(g, x1)))
