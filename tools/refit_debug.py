"""Debug: the refit cycle of bench.py with diagnostics of both sampling rounds and of the refit's accuracy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bayesfast_amd as bfa
from bayesfast_amd.core.refit import select_fit_points
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
d, C, seed = 64, 4096, 2024
spec, cov = correlated_gaussian_spec(d)
prec = np.linalg.inv(cov)
logp_true = lambda x: -0.5 * np.einsum('ij,jk,ik->i', x, prec, x)
su = bfa.PolyModel('quadratic', input_size=d, output_size=1, bound_options=dict(alpha_p=150.))
dens = bfa.SurrogateDensity(su)
n_eval = 2 * su.n_param
x = 1.5 * np.random.default_rng(seed).normal(size=(n_eval, d))
dens.fit(x, logp_true(x))
kw = dict(n_chain=C, n_iter=1500, n_warmup=500, random_generator=seed)
def run(tag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tt = bfa.sample(dens, dict(kw), verbose=False)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    st = tt.device('stats')
    ts = st[:, :, _lib.NSTATS.index('tree_size')]
    print(tag, '%.1f ms' % ms, 'mean tree size warm-up %.2f after %.2f' % (ts[:, :500].mean().item(), ts[:, 500:].mean().item()),
          'max depth', int(st[:, :, _lib.NSTATS.index('tree_depth')].max().item()), 'divergent %.4f' % st[:, :, _lib.NSTATS.index('diverging')].mean().item(),
          'alpha %.3f' % su._alpha, 'step %.4f' % st[:, -1, _lib.NSTATS.index('step_size_bar')].mean().item())
    return tt
run('warm'); tt = run('sample_0')
xf, lf, n_true = select_fit_points(tt, None, logp_true, n_eval, logp_cutoff=False)
dens.fit(xf, lf)
# accuracy of the refit against numpy's lstsq on the same design
from bayesfast_amd.modules.poly import PolyModel
A = np.concatenate([np.ones((xf.shape[0], 1)), xf] + [np.stack([xf[:, i] * xf[:, j] for i in range(d) for j in range(i, d)], 1)], 1)
print('design cond (column-equilibrated): %.3e' % np.linalg.cond(A / np.linalg.norm(A, axis=0)))
ref = np.linalg.lstsq(A, lf, rcond=None)[0]
res_ref = np.abs(A @ ref - lf).max()
f = np.array([np.ravel(su.fun(r)[0])[0] for r in xf[:200]]); lf_h, xf_h = lf, xf; 
try:
    print('max residual on fit points: device fit %.3e, lstsq %.3e' % (np.abs(f - lf[:200]).max(), res_ref))
except Exception as e:
    print('residual check failed', e)
del tt
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
run('sample_1')
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
run('sample_1 again')
time.sleep(0.03); run('after 30 ms idle')
time.sleep(0.2); run('after 200 ms idle')
run('back to back')
dens.fit(xf, lf); run('right after a fit')
dens.fit(xf, lf); torch.cuda.synchronize(); time.sleep(0.05); run('fit, 50 ms idle')
