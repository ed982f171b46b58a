#!/usr/bin/env python3
"""Pipeline density (bfhip_pld.h) on its own: the stand-alone evaluation's rate and the fused sampler's, on a synthetic
DES-shaped density (random coefficients, no fit), to separate the cost of the two contractions from the sampler's.

  python3 tools/pld_rate.py [m] [d] [n_quad_inputs] [chains]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


from bayesfast_amd.workloads import random_pipeline_spec as spec_of  # noqa: E402


def main():
    import torch
    from bayesfast_amd.device import get_context, DeviceDensity
    from bayesfast_amd.chains import DeviceChains
    from bayesfast_amd.workloads import flops_per_leapfrog_spec
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 457
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 27
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    C = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    ctx = get_context(0)
    spec = spec_of(m, d, nq)
    dd = DeviceDensity(spec, ctx)
    fl = flops_per_leapfrog_spec(spec)
    rng = np.random.default_rng(1)
    out = {'m': m, 'd': d, 'n_quad': nq, 'flops_per_eval': fl}
    n = 65536
    x = ctx.tensor(rng.normal(size=(n, d)) * 0.2)
    dd.logp_and_grad(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(ctx.stream)
    for _ in range(5):
        dd.logp_and_grad(x)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 5
    out['eval'] = {'points_per_s': n / t, 'TFLOPs': n * fl / t / 1e12, 'us_per_16_points_per_cu': t * 1e6 / (n / 16 / 256)}
    x0 = rng.normal(size=(C, d)) * 0.1
    ch = DeviceChains(dd, x0, seed=1)
    ch.run(150, 'NUTS', n_warmup=150, check=False)
    for _ in range(3):   # (the layout of a launch follows the trees of the launches before it: let 'auto' settle)
        ch.run(50, 'NUTS', n_warmup=150, check=False)
    lf0 = ch.total_leapfrog
    torch.cuda.synchronize()
    e0.record(ctx.stream)
    _, st = ch.run(50, 'NUTS', n_warmup=150, check=False)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    ch.raise_on_error()
    t = e0.elapsed_time(e1) * 1e-3
    nl = ch.total_leapfrog - lf0
    out['nuts'] = {'leapfrog_per_s': nl / t, 'TFLOPs': nl * fl / t / 1e12, 'mean_tree_size': float(st[:, :, 3].mean()),
                   'us_per_trip': t * 1e6 / (50 * (float(st[:, :, 3].mean()) + 1)), 'kernel': __import__('bayesfast_amd')._lib.last_kernel()}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
