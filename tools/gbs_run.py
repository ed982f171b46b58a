"""Tuning: GBS end to end on a 64-d correlated Gaussian (known log Z) with a cProfile of the host side.
usage: python tools/gbs_run.py [n_samples] [dim]"""
import sys, os, time, warnings, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.workloads import correlated_gaussian_spec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
_, cov = correlated_gaussian_spec(d)
prec = np.linalg.inv(cov)
L = np.linalg.cholesky(cov)
rng = np.random.default_rng(3)
x = (rng.normal(size=(n, d)) @ L.T).reshape(8, n // 8, d)
logp = lambda z: -0.5 * np.einsum('...i,ij,...j->...', z, prec, z)
logz_true = 0.5 * d * np.log(2 * np.pi) + 0.5 * np.linalg.slogdet(cov)[1]
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    for rep in range(2):
        gbs = bfa.GBS(sit=dict(n_iter=10, random_generator=5), n_q=n // 2)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        if rep:
            pr.enable()
        logz, err = gbs.run(x, logp)
        torch.cuda.synchronize()
        pr.disable()
        print('GBS %d x %d: %.2f s, log Z = %.3f +- %.3f (true %.3f)' % (n, d, time.perf_counter() - t0, logz, err, logz_true), flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
