"""Tuning: bench.py's config-5 evidence block (fit -> sample() -> GBS at 1024 chains x 128-d) with a cProfile of the GBS stage, host
side by cumulative time.  usage: python tools/evidence_profile.py [sit_iter]"""
import sys, os, time, warnings, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.workloads import planck_like_logp
sit_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(0)
d, chains, n_iter, n_warmup = 128, 1024, 340, 120
logp, chol = planck_like_logp(d, amp=0.)
m16 = np.arange(16)
su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic'), bfa.PolyConfig('cubic-2', input_mask=m16),
                    bfa.PolyConfig('cubic-3', input_mask=m16)], input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
den = bfa.SurrogateDensity(su)
x_fit = rng.normal(size=(2 * su.n_param, d)) @ chol.T * 1.3
for rep in range(2):
    t0 = time.perf_counter(); den.fit(x_fit, logp(x_fit)); torch.cuda.synchronize()
    print('fit %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
t0 = time.perf_counter()
tt = bfa.sample(den, {'n_chain': chains, 'n_iter': n_iter, 'n_warmup': n_warmup, 'random_generator': 0}, verbose=False)
torch.cuda.synchronize()
print('sample %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
from bayesfast_amd.transforms import ica as _ica
_ica._TIME_REPLAYS = True
gbs = bfa.GBS(sit=dict(n_iter=sit_iter, random_generator=5), n_q=chains * (n_iter - n_warmup) // 2)
pr = cProfile.Profile()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    t0 = time.perf_counter()
    pr.enable()
    logz, err = gbs(tt, den.logp)
    torch.cuda.synchronize()
    pr.disable()
print('gbs %.1f ms, logZ %.4f +- %.4f (exact %.4f)' % ((time.perf_counter() - t0) * 1e3, logz, err,
      0.5 * d * np.log(2. * np.pi) + float(np.sum(np.log(np.diag(chol))))), flush=True)
from bayesfast_amd.transforms import ica
print('FastICA chunks:', ica.GRAPH_STATS)
ts = [a.elapsed_time(b) for a, b in ica._REPLAY_EVENTS]
print('replays: %d, device ms each: min %.2f median %.2f max %.2f, total %.1f ms' % (len(ts), min(ts), sorted(ts)[len(ts) // 2], max(ts), sum(ts)))
pstats.Stats(pr).sort_stats('cumulative').print_stats(40)
