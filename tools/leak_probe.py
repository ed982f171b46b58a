"""Tuning: the in-step layouts against the wave layout when chains leak through the surrogate's bound (the headline family with a
training set FIT_SCALE times the posterior's width: below ~1.25 some chains sit outside the alpha-ellipsoid): rates, the share of
samples outside, and the group kernels' trip counters.  usage: FIT_SCALE=1.2 python tools/leak_probe.py [chains] [dim]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(D, fit_scale=float(os.environ.get('FIT_SCALE', 1.2)))
po = spec['poly']
mu, H, alpha = np.asarray(po['mu']), np.asarray(po['hess']), float(po['alpha'])
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(1).normal(size=(C, D))
for layout in ('split', 'group', 'wave', 'auto'):
    ch = DeviceChains(dens, x0, seed=3)
    kw = dict(n_warmup=750, check=False, layout=layout)
    ch.run(750, 'NUTS', **kw)
    gc = torch.zeros(4, dtype=torch.int64, device=ctx.device)
    _lib.debug_buffer('group_counters', gc)
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    s, st = ch.run(250, 'NUTS', **kw)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    _lib.debug_buffer('group_counters', None)
    x = s.reshape(-1, D).cpu().numpy()
    beta = np.sqrt(np.sum(((x - mu) @ H) * (x - mu), 1))
    out_chain = (beta.reshape(C, -1) > alpha).mean(1)
    print('layout %-5s %-34s %.3g leapfrog steps/s, tree %.1f; samples outside %.2f %%, chains ever outside %d of %d; trips %s' % (
        layout, _lib.last_kernel(), (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), st[:, :, _lib.NSTATS.index('tree_size')].mean().item(),
        100 * np.mean(beta > alpha), int((out_chain > 0).sum()), C, [int(v) for v in gc.cpu().numpy()]), flush=True)
