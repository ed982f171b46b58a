"""Tuning: the launches of the default run (4096 chains x 64-d, 1500 iterations, 500 of them warm-up) one by one: time, leapfrog
steps, rate, layout.  usage: python tools/launch_times.py [chains] [dim] [layout of every launch, comma-separated; default auto]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lays = sys.argv[3].split(',') if len(sys.argv) > 3 else ['auto'] * 9
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(d)
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(1).normal(size=(C, d))
kname = _lib.last_kernel
for rep in range(2):
    ch = DeviceChains(dens, x0, seed=3)
    rows = []
    for n, lay_in in zip([100] * 5 + [250] * 4, lays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lf0 = ch.total_leapfrog
        e0.record(ctx.stream)
        s, st = ch.run(n, 'NUTS', n_warmup=500, check=False, launch_iters=n, layout=lay_in)
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        ts = st[:, :, _lib.NSTATS.index('tree_size')]
        tot = ts.sum(1)                                   # leapfrog steps per chain in this launch
        wg = tot.view(-1, 16).max(1).values               # ... of the slowest chain of every 16-chain workgroup
        rows.append((n, e0.elapsed_time(e1), ch.total_leapfrog - lf0, ch.last_layout, kname(), float(ts.mean()), float(ts.max()),
                     float(tot.mean() / wg.mean()), float(tot.mean() / tot.max())))
tot = 0.
for n, ms, lf, lay, kn, tm, tx, e16, eall in rows:
    tot += ms
    print('%4d iterations: %6.2f ms  %.3g leapfrogs  %.3g /s  layout %-5s %-28s tree mean %.1f max %d  chain totals: mean / mean of 16-chain maxima %.2f, mean / max %.2f' % (
        n, ms, lf, lf / ms * 1e3, lay, kn, tm, tx, e16, eall))
print('total %.1f ms' % tot)
