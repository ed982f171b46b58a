#!/usr/bin/env python3
"""A BayesFast cosmology-style pipeline on the GPU path, in the shape of examples/des-y1-w-cosmosis.ipynb of the reference:

    x (27 parameters in a box)  ->  theory vector m (457 whitened data points; the expensive model)
                                ->  like = -|m - d|^2 / 2 + norm
                                ->  logp = like + Gaussian prior on 13 of the parameters

The expensive model is replaced by a multi-output polynomial surrogate (linear in all parameters, quadratic in 9), the
likelihood and the prior are analytic, and NUTS runs on the whole pipeline inside one fused kernel: per gradient two dense
FP64-MFMA contractions with the chains as columns, the (457, 27) Jacobian is never formed.  The
"theory" here is a synthetic stand-in evaluated on the host (bayesfast_amd.workloads.des_like_pipeline); a real run would
call CosmoSIS at the fit points.

    python examples/des_like_pipeline.py [--chains 4096]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesfast_amd as bfa  # noqa: E402
from bayesfast_amd.workloads import des_like_pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=4096)
    ap.add_argument('--rounds', type=int, default=2)
    a = ap.parse_args()
    w = des_like_pipeline()
    d, m = w['d'], w['m']
    lo, hi = w['para_range'][:, 0], w['para_range'][:, 1]
    rng = np.random.default_rng(0)

    su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic', input_mask=w['nonlinear'])], input_size=d, output_size=m,
                       input_scales=w['para_range'])
    den = bfa.Chi2PipelineDensity(su, w['data'], prec_diag=np.ones(m), logp0=w['norm'], prior_mu=w['prior_mu'], prior_prec=w['prior_prec'],
                                  prior_c0=w['prior_c0'], input_scales=w['para_range'], hard_bounds=True)
    n_eval = 4 * su.n_param
    u0 = (w['x_true'] - lo) / (hi - lo)
    x_fit = lo + (hi - lo) * np.clip(u0 + 0.08 * rng.normal(size=(n_eval, d)), 0.02, 0.98)   # round 0: a cloud around a fiducial point
    tt = None
    for r in range(a.rounds):
        t0 = time.perf_counter()
        y_fit = w['model'](x_fit)                     # the "expensive" calls: n_eval of them per round
        den.fit(x_fit, w['logp'](x_fit), y=y_fit)     # all 457 outputs from one factorisation, on the device
        t1 = time.perf_counter()
        x0 = x_fit[rng.integers(0, n_eval, a.chains)]
        tt = bfa.sample(den, dict(n_chain=a.chains, n_iter=600, n_warmup=300, x_0=x0, random_generator=r), verbose=False)
        s = tt.get()                                   # original space, post-warm-up
        t2 = time.perf_counter()
        print('round %d: fit %.0f ms, sample %d x 600 in %.0f ms (%d leapfrog steps); posterior mean offset %.2f sigma, '
              'width / prior range %.3f' % (r, (t1 - t0) * 1e3, a.chains, (t2 - t1) * 1e3, tt.n_call,
                                             np.max(np.abs(s.mean(0) - w['x_true']) / s.std(0)), np.mean(s.std(0) / (hi - lo))))
        # next round's fit points: a thinned subset of the samples (the reference's recipe picks them by logq)
        x_fit = s[rng.choice(s.shape[0], n_eval, replace=False)]
    err = np.max(np.abs(den.logp(s[:200]) - w['logp'](s[:200])))
    print('surrogate vs true log-density on 200 posterior samples: max |diff| = %.3f' % err)


if __name__ == '__main__':
    main()
