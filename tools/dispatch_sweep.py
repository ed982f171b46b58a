"""The automatic layout dispatch (DeviceChains.run(layout='auto'): chains._small_problem, _trees_in_step and the library's own
rules) against every forced layout, over workload families it was NOT tuned on one by one: dimension x chain count x tree size
(target_accept) x tree heterogeneity (per-dimension scales with the metric adaptation off) x feature set (plain, decay term,
constraint transform).  For every cell: leapfrog steps/s of group / split / wave / auto in post-adaptation launches, the kernel
'auto' ran in its last launch, and auto / best.  usage: python tools/dispatch_sweep.py [quick]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
ctx = get_context(0)
KN = _lib.last_kernel


def spec_of(d, family):
    if family == 'hetero':   # trees that differ from chain to chain and from iteration to iteration
        spec, _ = correlated_gaussian_spec(d, scales=np.logspace(-0.5, 0.5, d))
    else:
        spec, _ = correlated_gaussian_spec(d)
    if family == 'decay':    # core/density.py:740-746 around the bound's ellipsoid, never active
        po = spec['poly']
        spec = dict(spec, use_decay=True, decay_mu=po['mu'], decay_hess=po['hess'], decay_alpha2=(1.5 * po['alpha'])**2, decay_gamma=0.1)
    if family == 'bounded':  # behind the constraint transform: all four kinds of bounds (density.py:92-140)
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        spec = dict(spec, ranges=np.stack([lo, lo + 18.], 1), hard_bounds=np.array([[1, 1], [1, 0], [0, 1], [0, 0]] * (d // 4), dtype=np.uint8))
    return spec


def rate(dens, x0, ta, layout, family):
    ch = DeviceChains(dens, x0, seed=3)
    kw = dict(n_warmup=600, target_accept=ta, check=False, layout=layout, adapt_metric=family != 'hetero')
    ch.run(600, 'NUTS', **kw)
    s, st = ch.run(200, 'NUTS', **kw)
    lf0 = ch.total_leapfrog
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    for _ in range(3):
        s, st = ch.run(200, 'NUTS', **kw)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    return (ch.total_leapfrog - lf0) / (e0.elapsed_time(e1) * 1e-3), st[:, :, _lib.NSTATS.index('tree_size')].mean().item(), KN()


cells = []
for family in ('plain', 'hetero', 'decay', 'bounded'):
    for d in ((64,) if quick else (16, 32, 64)):
        for C in ((1024, 4096) if quick else (512, 1024, 2048, 4096)):
            for ta in ((0.8,) if quick else ((0.8, 0.95) if family != 'hetero' else (0.9,))):
                cells.append((family, d, C, ta))
print('%-8s %3s %5s %5s %6s | %9s %9s %9s | %9s  auto/best  kernel of auto' % ('family', 'd', 'C', 'ta', 'tree', 'group', 'split', 'wave', 'auto'))
worst = 1.
for family, d, C, ta in cells:
    dens = DeviceDensity(spec_of(d, family), ctx)
    x0 = np.random.default_rng(1).normal(size=(C, d)) * (0.3 if family == 'bounded' else 1.)
    r = {}
    for layout in ('group', 'split', 'wave', 'auto'):
        r[layout] = rate(dens, x0, ta, layout, family)
    best = max(r[k][0] for k in ('group', 'split', 'wave'))
    ratio = r['auto'][0] / best
    worst = min(worst, ratio)
    print('%-8s %3d %5d %5.2f %6.1f | %9.3g %9.3g %9.3g | %9.3g  %5.2f      %s' % (family, d, C, ta, r['auto'][1], r['group'][0], r['split'][0], r['wave'][0],
                                                                             r['auto'][0], ratio, r['auto'][2]), flush=True)
print('worst auto / best over %d cells: %.2f' % (len(cells), worst))
