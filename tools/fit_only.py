"""Times the device fit of the headline surrogate (also the target of rocprofv3 --kernel-trace --stats): the whole
PolyModel.fit (host arrays in, coefficients and bound out) and, inside it, the least-squares solve alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd import PolyModel, _lib
from bayesfast_amd.device import get_context, _ptr
d = int(os.environ.get('DIM', 64))
su = PolyModel('quadratic', input_size=d, output_size=1)
P = su.n_param
x = np.random.default_rng(3).normal(size=(2 * P, d))
y = -0.5 * np.sum(x**2, 1) + 0.1 * np.sin(x[:, 0])
ts = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    su.fit(x, y[:, None], logp=y)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print('PolyModel.fit n = {} P = {}: {} ms'.format(2 * P, P, ' '.join('%.2f' % t for t in ts)))
# the solve alone, on resident arrays
ctx = get_context()
n = 2 * P
A = ctx.empty((n, P))
xt = ctx.tensor(x, torch.float64)
_lib.check(ctx._lib.bfhip_design_block(ctx.handle, 0, n, d, _ptr(xt), None, _ptr(A), P, 0))       # [1 | x]
_lib.check(ctx._lib.bfhip_design_block(ctx.handle, 1, n, d, _ptr(xt), None, _ptr(A), P, d + 1))   # quadratic
B = ctx.tensor(y[:, None], torch.float64)
G, r, work = ctx.empty((P, P)), ctx.empty((P, 1)), ctx.empty((n + P,))
info = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
for nref in (0, 2):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for _ in range(5):
        ev[0].record(torch.cuda.current_stream())
        _lib.check(ctx._lib.bfhip_lstsq(ctx.handle, n, P, 1, _ptr(A), P, _ptr(B), _ptr(G), _ptr(r), nref, _ptr(work), _ptr(info)))
        ev[1].record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ms.append(ev[0].elapsed_time(ev[1]))
    c = r.cpu().numpy()[:, 0]
    res = (A @ r - B).abs().max().item()
    print('bfhip_lstsq n_refine = {}: {} ms   info {}  max |A c - b| {:.3e}'.format(nref, ' '.join('%.2f' % t for t in ms), int(info.item()), res))
if P <= 3000 or os.environ.get('HOST_CHECK'):   # (the host's gelsd of a 16770 x 8385 matrix takes minutes on a loaded box)
    ref = np.linalg.lstsq(A.cpu().numpy(), y, rcond=None)[0]
    print('max |c - lstsq| / max |c|: {:.3e}'.format(np.abs(c - ref).max() / np.abs(ref).max()))
# where the host time of PolyModel.fit goes
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    su.fit(x, y[:, None], logp=y)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
