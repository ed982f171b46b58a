"""Tuning: what config 5's shard (1024 chains x 128-d, cubic-cross surrogate) spends its trip on: the same chains with and
without the cubic configs, at tree depth limits 10 and 6 (the deep levels of the subtree stack are in global scratch).
usage: python tools/config5_ab.py [chains]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.device import get_context
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import planck_like_logp
from bayesfast_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = get_context(0)
d = 128
rng = np.random.default_rng(2024)
logp, chol = planck_like_logp(d, amp=float(os.environ.get("AMP", "0.02")))
m16 = np.arange(16)
kname = _lib.last_kernel
for cubic in (True, False):
    cfgs = [bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic')]
    if cubic:
        cfgs += [bfa.PolyConfig('cubic-2', input_mask=m16), bfa.PolyConfig('cubic-3', input_mask=m16)]
    su = bfa.PolyModel(cfgs, input_size=d, output_size=1)
    den = bfa.SurrogateDensity(su)
    x_fit = rng.normal(size=(2 * 9201, d)) @ chol.T
    den.fit(x_fit, logp(x_fit))
    x0 = x_fit[rng.integers(0, x_fit.shape[0], C)] * 0.5
    dd = den.device(ctx)
    for depth in (10, 6):
        ch = DeviceChains(dd, x0, seed=5)
        kw = dict(n_warmup=150, check=False, max_treedepth=depth)
        ch.run(150, 'NUTS', **kw)
        iters = 20 if depth == 10 else 200
        ch.run(iters, 'NUTS', **kw)
        lf0 = ch.total_leapfrog
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)
        ch.run(iters, 'NUTS', **kw)
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        n = ch.total_leapfrog - lf0
        print('chains %d cubic %d max_treedepth %d: %.3g leapfrog steps/s, mean tree %.1f, %s' % (
            C, cubic, depth, n / (e0.elapsed_time(e1) * 1e-3), n / (C * iters), kname()), flush=True)
