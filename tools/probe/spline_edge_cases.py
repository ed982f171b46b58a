import sys, os, warnings
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bayesfast_amd') else os.getcwd())
import numpy as np, torch
from bayesfast_amd.transforms import SIT
from bayesfast_amd.device import get_context
ctx = get_context(0)
rng = np.random.default_rng(0)
n, d = 20000, 6
y = rng.normal(size=(n, d))
y[:, 1] = rng.integers(0, 5, size=n)            # five distinct values
y[:, 2] = np.round(y[:, 2], 1)                  # ~70 distinct values
y[:, 3] = np.where(rng.uniform(size=n) < 0.5, -50., 50.) + 0.01 * rng.normal(size=n)   # two far clusters
sit = SIT(n_iter=1, random_generator=1)
sit._weights = np.ones(n) / n
for name, cols in (('all', list(range(d))), ('normal only', [0, 4, 5]), ('discrete', [0, 1]), ('rounded', [0, 2]), ('clusters', [0, 3])):
    yd = ctx.tensor(y[:, cols])
    for route in ('device', 'host'):
        try:
            if route == 'host':
                sit.cubic_options = dict(bins=100.0)     # (a float: the device builder takes it too?  force the host: unknown key)
                sit.cubic_options = {}
                old = SIT._build_on_device
                SIT._build_on_device = lambda self, *a: None
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter('always')
                t = sit._gaussianize(yd)
            print(name, route, 'ok knots', [s.x.size for s in t.splines], 'warnings', sorted({str(x.message)[:40] for x in w}))
        except Exception as e:
            print(name, route, 'raised', type(e).__name__, str(e)[:80])
        finally:
            if route == 'host':
                SIT._build_on_device = old
