"""Tuning: one SIT iteration at 64 dimensions x 400 000 points with a cProfile of the host side (for rocprofv3 --kernel-trace
--stats -- python3 tools/sit_iter.py).  usage: python tools/sit_iter.py [n_points] [dim]"""
import sys, os, time, warnings, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.transforms import SIT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(2)
x = rng.laplace(size=(n, d)) @ (np.eye(d) + 0.2 * rng.normal(size=(d, d)))
sit = SIT(n_iter=4, random_generator=3)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    sit.fit(x, n_run=1)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        sit.fit(n_run=1)
        torch.cuda.synchronize()
        print('SIT iteration, %d x %d: %.3f s' % (n, d, time.perf_counter() - t0), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    sit.fit(n_run=1)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
