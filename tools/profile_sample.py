"""Host-side profile of sample(): where does the wall time go around the kernel launches?"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.workloads import correlated_gaussian_spec
d = 64
spec, cov = correlated_gaussian_spec(d)
prec = np.linalg.inv(cov)
su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
den = bfa.SurrogateDensity(su)
rng = np.random.default_rng(7)
x_fit = 1.5 * rng.normal(size=(2 * su.n_param, d))
den.fit(x_fit, -0.5 * np.einsum('ni,ij,nj->n', x_fit, prec, x_fit))
kw = {'n_chain': 4096, 'n_iter': 1500, 'n_warmup': 500, 'random_generator': 5}
bfa.sample(den, dict(kw, n_iter=40, n_warmup=20), verbose=False)  # warm the allocator
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
tt = bfa.sample(den, kw, verbose=False)
pr.disable()
dt = time.perf_counter() - t0
nl = tt.stat('tree_size').sum()
print('sample(): %.3f s wall, %d leapfrogs -> %.3e leapfrog steps/s end to end' % (dt, nl, nl / dt))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14)
print('\n'.join(s.getvalue().splitlines()[:40]))
