"""Leapfrog rates of the fused sampler on the other BASELINE configs' shapes (parity-test cases, not the headline)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
res = {}
for name, d, C, kw in (('config2_32d_1024', 32, 1024, {}), ('config2_32d_4096', 32, 4096, {}), ('headline_64d_4096', 64, 4096, {}),
                       ('config4_64d_4096_ta95', 64, 4096, dict(target_accept=0.95)), ('config5_128d_1024', 128, 1024, {}), ('config5_128d_4096', 128, 4096, {}),
                       ('full_metric_64d_1024', 64, 1024, dict(metric='full'))):
    spec, _ = correlated_gaussian_spec(d)
    dens = DeviceDensity(spec, ctx)
    x0 = np.random.default_rng(1).normal(size=(C, d))
    ckw = {k: v for k, v in kw.items() if k == 'metric'}
    rkw = {k: v for k, v in kw.items() if k != 'metric'}
    dc = DeviceChains(dens, x0, seed=3, **ckw)
    for _ in range(3):
        dc.run(100, 'NUTS', n_warmup=300, check=False, **rkw)
    ts = []
    s = st = None
    for _ in range(3):  # HIP events around the launch; the output arrays of the previous launch are reused
        l0 = dc.total_leapfrog
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        s, st = dc.run(100, 'NUTS', n_warmup=300, check=False, samples=s, stats=st, **rkw)
        e1.record(); torch.cuda.synchronize(); dt = e0.elapsed_time(e1) * 1e-3
        ts.append(((dc.total_leapfrog - l0) / dt, dt * 1e3))
    dc.raise_on_error()
    r = np.array(ts)
    res[name] = dict(leapfrog_per_s=float(r[:, 0].mean()), ms_per_100_iterations=float(r[:, 1].mean()),
                     mean_tree_size=float(st[:, :, _lib.NSTATS.index('tree_size')].mean().item()))
    print(name, res[name], flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'configs.json'), 'w'), indent=1)
