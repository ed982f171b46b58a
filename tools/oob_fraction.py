"""How often the chains of the config blocks of bench.py are outside the surrogate's bound (where every evaluation takes a
second pass at the projected point, modules/poly.py:480-503) and inside the decay region: from the samples of a timed block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from bayesfast_amd.device import get_context
ctx = get_context(0)
import bayesfast_amd as bfa
from bayesfast_amd.workloads import banana_logp, funnel_logp
rng = np.random.default_rng(2024)
for name, logp_f, ta in (('funnel', funnel_logp, 0.95), ('banana_decay', banana_logp, 0.8)):
    d, C = 64, 4096
    logp = logp_f(d)
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
    den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=True))
    x_fit = rng.normal(size=(2 * su.n_param, d))
    den.fit(x_fit, logp(x_fit))
    x0 = x_fit[rng.integers(0, x_fit.shape[0], C)] * 0.5
    r, s, st = bench._sampler_block(ctx, den, x0, 2024, ta, 300, 50, 1, 0., name)
    x = s.reshape(-1, d).cpu().numpy()
    beta = np.sqrt(np.sum(((x - su._mu) @ su._hess) * (x - su._mu), 1))
    bd = np.sqrt(np.sum(((x - den._mu) @ den._hess) * (x - den._mu), 1))
    print('%s: %.3g steps/s, %s; samples outside the bound (beta > alpha = %.2f): %.1f %%; in the decay region (beta_d > %.2f): %.1f %%; '
          'median beta %.2f' % (name, r['value'], r['roofline']['kernel'], su._alpha, 100 * np.mean(beta > su._alpha), den._alpha,
                                100 * np.mean(bd > den._alpha), np.median(beta)))
