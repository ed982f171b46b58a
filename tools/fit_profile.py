"""Tuning: cProfile of SurrogateDensity.fit on the config-5 surrogate (P = 9201, 18402 points x 128-d), host side by cumulative
time.  usage: python tools/fit_profile.py"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.workloads import planck_like_logp
rng = np.random.default_rng(0)
d = 128
logp, chol = planck_like_logp(d, amp=0.)
m16 = np.arange(16)
su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic'), bfa.PolyConfig('cubic-2', input_mask=m16),
                    bfa.PolyConfig('cubic-3', input_mask=m16)], input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
den = bfa.SurrogateDensity(su)
x_fit = rng.normal(size=(2 * su.n_param, d)) @ chol.T * 1.3
y = logp(x_fit)
for rep in range(3):
    t0 = time.perf_counter(); den.fit(x_fit, y); torch.cuda.synchronize()
    print('fit %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
pr = cProfile.Profile()
pr.enable(); den.fit(x_fit, y); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
