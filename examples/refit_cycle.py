#!/usr/bin/env python3
"""One sampling round of a BayesFast-style recipe on the GPU path: fit a quadratic surrogate to an expensive
log-density, sample it with NUTS over many chains, pick refit points out of the samples, evaluate the true model
there, refit, sample again, and weight the final samples by importance.

Mirrors what ``Recipe._sam_step`` / ``_pos_step`` do around ``sample`` (bayesfast/core/recipe.py:986-1185,1270-1297)
with the pieces this package provides; the "expensive" model here is a cheap stand-in (a 32-d Gaussian with a mild
quartic term), evaluated on the host like a user's likelihood would be.

    python examples/refit_cycle.py [--chains 2048] [--dim 32]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesfast_amd as bfa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=2048)
    ap.add_argument('--dim', type=int, default=32)
    a = ap.parse_args()
    d = a.dim
    rng = np.random.default_rng(0)
    L = np.eye(d) + 0.3 * np.tril(rng.normal(size=(d, d)), -1) / np.sqrt(d)
    prec = L @ L.T

    def logp_true(x):  # the "expensive" model
        x = np.atleast_2d(x)
        return -0.5 * np.einsum('ni,ij,nj->n', x, prec, x) - 0.002 * np.sum(x**4, axis=1)

    su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
    den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
    n_eval = 2 * su.n_param
    x_fit = 1.5 * rng.normal(size=(n_eval, d))          # round 0: a broad training set
    for rnd in range(2):
        t0 = time.perf_counter()
        den.fit(x_fit, logp_true(x_fit))
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        tt = bfa.sample(den, {'n_chain': a.chains, 'n_iter': 600, 'n_warmup': 300, 'random_generator': 10 + rnd},
                        verbose=False)
        t_s = time.perf_counter() - t0
        x = tt.get()                                      # post-warm-up samples, original space, all chains
        logq = tt.get(return_type='logp')
        n_lf = int(tt.stat('tree_size').sum())
        print('round %d: fit %.0f ms (%d points, %d parameters); sample %.2f s, %d leapfrog steps, mean tree size %.1f, '
              '%d divergences' % (rnd, t_fit * 1e3, x_fit.shape[0], su.n_param, t_s, n_lf, tt.stat('tree_size').mean(),
                                  int(tt.stat('diverging').sum())))
        if rnd == 0:   # refit on points taken from the samples (SystematicResampler + logp_cutoff)
            x_fit, lp_fit, n_calls = bfa.select_fit_points(x, logq, logp_true, n_eval)
            print('         refit set: %d points kept of %d true-model evaluations' % (x_fit.shape[0], n_calls))
    sub = slice(None, None, max(1, x.shape[0] // 20000))
    w, wt = bfa.importance_weights(logp_true(x[sub]), logq[sub])
    ess = wt.sum()**2 / np.sum(wt**2)
    print('importance weights of the final samples: effective sample size %.0f of %d (%.1f %%)' % (ess, wt.size, 100 * ess / wt.size))


if __name__ == '__main__':
    main()
