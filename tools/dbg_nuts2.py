import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, 'tests'), os.path.join(R, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np
import test_gpu_sampler as T
from bayesfast_amd.device import get_context
ctx = get_context(0)
samp = np.load(os.path.join(T.G, 'sampler.npz'))
np.set_printoptions(linewidth=200, precision=6)
name = 'plain16'
spec = T._spec(samp, name + '.')
spec['poly']['use_bound'] = False
rng = np.random.default_rng(3)
n_chain = 16
x0 = rng.normal(size=(n_chain, spec['d'])) * 0.5
for mtd in (1, 2, 3, 4, 10):
    for step in (0.3, 1.0):
        kw = dict(max_treedepth=mtd, step_size=step)
        s, st, dc = T._device_chains(ctx, spec, x0, 30, 0, **kw)
        runs = T._oracle_chains(spec, x0, 30, 0, **kw)
        nbad_d, nbad_q, first = 0, 0, []
        for i, (so, sto, ch) in enumerate(runs):
            bad = np.nonzero((st['tree_size'][i] != sto['tree_size']) | (st['tree_depth'][i] != sto['tree_depth']))[0]
            err = np.abs(s[i] - so).max(1)
            badq = np.nonzero(err > 1e-6)[0]
            nbad_d += bad.size > 0
            nbad_q += badq.size > 0
            if badq.size:
                k = badq[0]
                first.append((i, int(k), int(bad[0]) if bad.size else -1, st['tree_depth'][i][k], sto['tree_depth'][k], st['tree_size'][i][k], sto['tree_size'][k],
                              round(float(st['energy'][i][k]), 4), round(float(sto['energy'][k]), 4), round(float(st['mean_tree_accept'][i][k]), 4), round(float(sto['mean_tree_accept'][k]), 4)))
        print('max_treedepth', mtd, 'step', step, 'chains with shape mismatch', nbad_d, 'with q mismatch', nbad_q)
        for f in first[:4]:
            print('    chain,iter_q,iter_shape,depth dev/orc,size dev/orc,energy dev/orc,acc dev/orc:', f)
