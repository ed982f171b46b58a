"""Soak test: the reference's default run length (1500 iterations, 500 warm-up) for 4096 chains on the headline
surrogate through the public sample() entry point; checks the posterior moments against the analytic Gaussian."""
import sys, os, time
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
x_fit = rng.normal(size=(2 * su.n_param, d))
t0 = time.perf_counter()
den.fit(x_fit, -0.5 * np.einsum('ni,ij,nj->n', x_fit, prec, x_fit))
t_fit = time.perf_counter() - t0
t0 = time.perf_counter()
tt = bfa.sample(den, {'n_chain': 4096, 'n_iter': 1500, 'n_warmup': 500, 'random_generator': 5}, verbose=False, iters_per_launch=250)
t_s = time.perf_counter() - t0
draws = tt.get().reshape(-1, d)
nl = tt.stat('tree_size').sum()
emp = np.cov(draws[::7], rowvar=False)
sd = np.sqrt(np.diag(cov))
print('fit %.1f ms; sample() %.2f s wall for %d leapfrogs (%.3e/s incl. host copies); divergences %d' % (t_fit * 1e3, t_s, nl, nl / t_s, int(tt.stat('diverging')[:, 500:].sum())))
print('max |mean| / sd = %.4f ; max relative covariance error = %.4f ; mean accept %.3f' % (np.abs(draws.mean(0) / sd).max(),
      np.max(np.abs(emp - cov) / np.sqrt(np.outer(np.diag(cov), np.diag(cov)))), tt.stat('mean_tree_accept')[:, 500:].mean()))
