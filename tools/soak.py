"""Stability soak: repeated full-size runs of every sampler path; checks that statistics stay finite, chains end without error
flags and posterior moments of the Gaussian workloads stay right."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
ctx = get_context(0)
t00 = time.time()
for rep in range(int(os.environ.get('REPS', 3))):
    for name, d, C, smp, kw, mod in (('nuts64', 64, 4096, 'NUTS', {}, None), ('nuts32', 32, 4096, 'NUTS', {}, None), ('hmc64', 64, 4096, 'HMC', {}, None),
                                     ('nuts64_bounded', 64, 4096, 'NUTS', {}, 'bounded'), ('nuts128', 128, 1024, 'NUTS', {}, None),
                                     ('nuts64_ta95', 64, 4096, 'NUTS', dict(target_accept=0.95), None), ('nuts10', 10, 1000, 'NUTS', {}, None)):
        spec, cov = correlated_gaussian_spec(d)
        if mod == 'bounded':
            spec = dict(spec, ranges=np.stack([np.full(d, -12.), np.full(d, 12.)], 1), hard_bounds=np.ones((d, 2), dtype=np.uint8))
        dc = DeviceChains(DeviceDensity(spec, ctx), np.random.default_rng(rep).normal(size=(C, d)) * 0.5, seed=100 + rep)
        t0 = time.time()
        s, st = dc.run(900, smp, n_warmup=400, **kw)
        torch.cuda.synchronize(); dt = time.time() - t0
        fin = torch.isfinite(st); fin[:, :, 9] = True  # (max_energy_change is +inf for a divergent step with a NaN energy, as in the reference)
        assert bool(torch.isfinite(s).all()) and bool(fin.all()) and not bool(torch.isnan(st).any()), name
        assert int((dc.sc[:, _lib.SC_FIELDS.index('error')] != 0).sum()) == 0, name
        x = s[:, 400:].reshape(-1, d)
        x = x.cpu().numpy()
        if mod == 'bounded':  # samples live in the transformed space (logistic onto (-12, 12)); compare in the original one
            x = -12. + 24. / (1. + np.exp(-x))
        vr = x.var(0) / np.diag(cov)
        print('rep %d %-15s %6.1f s  leapfrogs %.3e  variance ratio %.3f..%.3f  |mean|/sd max %.4f' % (
            rep, name, dt, dc.total_leapfrog, vr.min(), vr.max(), np.abs(x.mean(0) / np.sqrt(np.diag(cov))).max()), flush=True)
        del s, st, dc
print('soak finished in %.0f s' % (time.time() - t00))
