"""Micro-benchmarks of the other entry points of the C ABI against their rooflines (MI355X): batched logp+grad,
batched leapfrog step, and the pieces of the surrogate fit.  Writes one JSON object to stdout."""
import sys, os, json, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity, _ptr
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib, PolyModel

ctx = get_context(0)
L, h = ctx._lib, ctx.handle
out = {}


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(ctx.stream)
    for _ in range(reps):
        fn()
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for d in (64, 32, 128):
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    dens.upload_if_needed()
    n = 1 << 20 if d <= 64 else 1 << 19
    x = ctx.tensor(np.random.default_rng(0).normal(size=(n, d)), torch.float64)
    lp, g = ctx.empty((n,)), ctx.empty((n, d))
    t = timeit(lambda: _lib.check(L.bfhip_logp_grad(h, n, _ptr(x), 0, _ptr(lp), _ptr(g))))
    byt = n * (16 * d + 8)
    flop = n * 4 * d * d
    out['logp_grad_d%d' % d] = dict(points=n, ms=t * 1e3, points_per_s=n / t, GBps_algorithmic=byt / t / 1e9, hbm_frac=byt / t / 8e12,
                                   TFLOPs=flop / t / 1e12, mfma_frac=flop / t / 78.6e12)
    eps = ctx.tensor(np.full(n, 0.1), torch.float64)
    var = ctx.tensor(np.ones((n, d)), torch.float64)
    q = x.clone(); p = ctx.tensor(np.random.default_rng(1).normal(size=(n, d)), torch.float64)
    e = ctx.empty((n,))
    L.bfhip_logp_grad(h, n, _ptr(q), 0, _ptr(lp), _ptr(g))
    t = timeit(lambda: _lib.check(L.bfhip_leapfrog(h, n, _ptr(eps), _ptr(var), _ptr(q), _ptr(p), _ptr(g), _ptr(lp), _ptr(e), None)))
    byt = n * (48 * d + 32 + 8 * d + 8)  # state in/out (48 d + 32) + var row + eps
    out['leapfrog_d%d' % d] = dict(chains=n, ms=t * 1e3, steps_per_s=n / t, GBps_algorithmic=byt / t / 1e9, hbm_frac=byt / t / 8e12,
                                  TFLOPs=flop / t / 1e12, mfma_frac=flop / t / 78.6e12)

# fit pieces at the headline size (n = 4290, P = 2145)
d = 64
su = PolyModel('quadratic', input_size=d, output_size=1)
P = su.n_param
n = 2 * P
rng = np.random.default_rng(3)
xh = rng.normal(size=(n, d))
x = ctx.tensor(xh, torch.float64)
A = ctx.empty((n, P))


def design():
    _lib.check(L.bfhip_design_block(h, 0, n, d, _ptr(x), None, _ptr(A), P, 0))       # (1 | x)
    _lib.check(L.bfhip_design_block(h, 1, n, d, _ptr(x), None, _ptr(A), P, 1 + d))   # quadratic products


sig = _lib.SYMBOLS['bfhip_design_block'][1]
try:
    t = timeit(design, reps=10)
    out['design_block'] = dict(ms=t * 1e3, GBps=n * P * 8 / t / 1e9, hbm_frac=n * P * 8 / t / 8e12)
except Exception as ex:
    out['design_block'] = dict(error=repr(ex), argtypes=str(sig))
y = ctx.tensor(rng.normal(size=(n, 1)), torch.float64)
G, r = ctx.empty((P, P)), ctx.empty((P, 1))
try:
    t = timeit(lambda: _lib.check(L.bfhip_gram(h, n, P, 1, _ptr(A), P, _ptr(y), _ptr(G), _ptr(r))), reps=10)
    out['gram'] = dict(ms=t * 1e3, TFLOPs=2. * n * P * P / t / 1e12, mfma_frac=2. * n * P * P / t / 78.6e12)
except Exception as ex:
    out['gram'] = dict(error=repr(ex), argtypes=str(_lib.SYMBOLS['bfhip_gram'][1]))
info = torch.zeros((1,), dtype=torch.int32, device=ctx.device)
G0, r0 = G.clone(), r.clone()


def solve():
    G.copy_(G0); r.copy_(r0)
    _lib.check(L.bfhip_solve_spd(h, P, 1, _ptr(G), _ptr(r), _ptr(info)))


try:
    t_copy = timeit(lambda: (G.copy_(G0), r.copy_(r0)), reps=10)
    t = timeit(solve, reps=10) - t_copy
    out['solve_spd'] = dict(ms=t * 1e3, GFLOPs=P**3 / 3. / t / 1e9, info=int(info.item()))
except Exception as ex:
    out['solve_spd'] = dict(error=repr(ex))
yh = -0.5 * np.sum(xh**2, 1)
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    su.fit(xh, yh[:, None], logp=yh)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
out['fit_total_host_in_host_out'] = dict(ms=min(ts[1:]) * 1e3, n=n, P=P)
print(json.dumps(out, indent=1))

# multi-output module at a cosmology-like size (m = 457 outputs)
try:
    from bayesfast_amd.device import DevicePolyModel
    d, m, n = 64, 457, 4096
    rng = np.random.default_rng(5)
    iu = np.triu_indices(d)
    cq = np.zeros((m, d, d))
    cq[:, iu[0], iu[1]] = rng.normal(size=(m, iu[0].size)) * 0.05
    poly = dict(input_size=d, output_size=m, use_bound=False,
                configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.arange(m), coef=rng.normal(size=(m, d + 1))),
                         dict(order='quadratic', input_mask=np.arange(d), output_mask=np.arange(m), coef=cq)])
    dm = DevicePolyModel(poly, ctx)
    dm.upload_if_needed()
    x = ctx.tensor(rng.normal(size=(n, d)), torch.float64)
    f, j = ctx.empty((n, m)), ctx.empty((n, m, d))
    t = timeit(lambda: _lib.check(L.bfhip_polymodel_eval(h, n, _ptr(x), _ptr(f), _ptr(j))), reps=5)
    t2 = timeit(lambda: _lib.check(L.bfhip_polymodel_eval(h, n, _ptr(x), _ptr(f), None)), reps=5)
    out2 = dict(points=n, outputs=m, ms_with_jacobian=t * 1e3, jac_GBps=n * m * d * 8 / t / 1e9, hbm_frac=n * m * d * 8 / t / 8e12,
                TFLOPs=2. * n * m * d * d / t / 1e12, mfma_frac=2. * n * m * d * d / t / 78.6e12, ms_values_only=t2 * 1e3,
                TFLOPs_values_only=2. * n * m * d * d / t2 / 1e12)
    print(json.dumps({'polymodel_eval_m457_d64': out2}, indent=1))
except Exception as ex:
    print(json.dumps({'polymodel_eval_m457_d64': {'error': repr(ex)}}))
