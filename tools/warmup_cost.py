"""Tuning: what the adaptation costs during warm-up: the first two launches (250 iterations each) of a fresh chain set on the
headline surrogate with the step-size / metric adaptation on and off.  usage: python tools/warmup_cost.py [layout]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
layout = sys.argv[1] if len(sys.argv) > 1 else 'wave'
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
dens = DeviceDensity(spec, ctx)
x0 = np.random.default_rng(1).normal(size=(4096, 64))
for am, asz in ((True, True), (False, True), (True, False), (False, False)):
    for rep in range(2):
        ch = DeviceChains(dens, x0, seed=3)
        ts, ms = [], []
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lf0 = ch.total_leapfrog
            e0.record(ctx.stream)
            s, st = ch.run(250, 'NUTS', n_warmup=500, adapt_metric=am, adapt_step_size=asz, check=False, layout=layout)
            e1.record(ctx.stream)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1)); ts.append((ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3))
    print('layout %s adapt_metric %d adapt_step %d: launch 1 %.1f ms (%.3g steps/s), launch 2 %.1f ms (%.3g steps/s)' % (layout, am, asz, ms[0], ts[0], ms[1], ts[1]))
