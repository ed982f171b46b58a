import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import numpy as np
import test_gpu_sampler as T
from bayesfast_amd.device import get_context
ctx = get_context(0)
samp = np.load(os.path.join(T.G, 'sampler.npz'))
name = sys.argv[1] if len(sys.argv) > 1 else 'plain16'
n_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_iter, n_warmup = 40, 25
spec = T._spec(samp, name + '.')
if len(sys.argv) > 3 and sys.argv[3] == 'nobound':
    spec['poly']['use_bound'] = False
if len(sys.argv) > 4:
    n_warmup = int(sys.argv[4])
rng = np.random.default_rng(3)
x0 = rng.normal(size=(n_chain, spec['d'])) * 0.5
s, st, dc = T._device_chains(ctx, spec, x0, n_iter, n_warmup)
runs = T._oracle_chains(spec, x0, n_iter, n_warmup)
np.set_printoptions(linewidth=200, precision=6)
for i, (so, sto, ch) in enumerate(runs):
    bad = np.nonzero((st['tree_size'][i] != sto['tree_size']) | (st['tree_depth'][i] != sto['tree_depth']))[0]
    err = np.abs(s[i] - so).max(1)
    print('chain', i, 'first mismatch', bad[:1], 'max q err before', err[:bad[0]].max() if bad.size and bad[0] > 0 else err.max())
    if bad.size and len(sys.argv) > 5:
        k = bad[0]
        for f in ('tree_depth', 'tree_size', 'mean_tree_accept', 'energy', 'step_size', 'max_energy_change', 'diverging'):
            print('   ', f, st[f][i][k - 1:k + 2], sto[f][k - 1:k + 2])
        print('    qerr', err[max(0, k - 2):k + 2])
