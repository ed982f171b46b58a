"""Exploration: behaviour of the config-3 banana workload over refit rounds (tree sizes, divergences, rates)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bayesfast_amd as bfa
from bayesfast_amd.workloads import banana_logp, sobol_normal
from bayesfast_amd.utils import SystematicResampler
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd import _lib

d, C = 64, int(os.environ.get('C', 4096))
logp = banana_logp(d)
decay = os.environ.get('DECAY', '0') == '1'
su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
den = bfa.SurrogateDensity(su, decay_options=dict(use_decay=decay))
P = su.n_param
x_fit = sobol_normal(2 * P, d, seed=1)
for rnd in range(int(os.environ.get('ROUNDS', 3))):
    t0 = time.perf_counter()
    den.fit(x_fit, logp(x_fit))
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    dd = den.device()
    x0 = x_fit[np.arange(C) % x_fit.shape[0]]
    ch = DeviceChains(dd, x0, seed=11 + rnd)
    out = []
    for k in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lf0 = ch.total_leapfrog
        s, st = ch.run(100, 'NUTS', n_warmup=200, check=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        stn = st.cpu().numpy()
        ts = stn[:, :, _lib.NSTATS.index('tree_size')]
        dv = stn[:, :, _lib.NSTATS.index('diverging')]
        td = stn[:, :, _lib.NSTATS.index('tree_depth')]
        print('round %d launch %d: %.1f ms, %.3e lf/s, tree_size mean %.1f max %d, depth max %d, diverging %.4f, fit %.0f ms'
              % (rnd, k, dt * 1e3, (ch.total_leapfrog - lf0) / dt, ts.mean(), ts.max(), td.max(), dv.mean(), t_fit * 1e3), flush=True)
    ch.raise_on_error()
    sn = s.cpu().numpy().reshape(-1, d)
    lq = stn[:, :, 0].reshape(-1)
    ok = np.all(np.isfinite(sn), axis=1) & np.isfinite(lq)
    sn, lq = sn[ok], lq[ok]
    print('  sample |x| max %.2f, logq range %.1f .. %.1f, true logp median %.1f' % (np.abs(sn).max(), lq.min(), lq.max(), np.median(logp(sn))))
    idx = SystematicResampler(require_unique=False)(lq, 2 * P)
    x_fit = sn[idx]
