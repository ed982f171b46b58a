import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, 'tests'), os.path.join(R, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np, torch
import test_gpu_sampler as T
from bayesfast_amd.device import get_context
from bayesfast_amd import _lib
from oracle import oracle as orc
ctx = get_context(0)
samp = np.load(os.path.join(T.G, 'sampler.npz'))
np.set_printoptions(linewidth=220, precision=5, suppress=True)
spec = T._spec(samp, 'plain16.')
spec['poly']['use_bound'] = False
rng = np.random.default_rng(3)
x0 = rng.normal(size=(16, 16)) * 0.5
chain, it_bad = 1, 21
kw = dict(max_treedepth=3, step_size=1.0)
cap = 4000
buf = torch.zeros((cap, 32), dtype=torch.float64, device='cuda')
L = _lib.lib()
L.bfhip_debug_trace.argtypes = [C.c_void_p, C.c_int, C.c_int]
L.bfhip_debug_trace(C.c_void_p(buf.data_ptr()), chain, cap)
s, st, dc = T._device_chains(ctx, spec, x0, 30, 0, **kw)
tr = buf.cpu().numpy()
ob = np.zeros(20000)
OL = orc.lib()
OL.bfo_set_trace.argtypes = [C.c_void_p, C.c_long]
OL.bfo_set_trace(ob.ctypes.data_as(C.c_void_p), ob.size)
runs = T._oracle_chains(spec, x0[chain:chain + 1], 30, 0, first_stream=chain, **kw)
so, sto, ch = runs[0]
rec = ob[:OL.bfo_set_trace and 20000].reshape(-1, 8)
sizes = sto['tree_size'].astype(int)
# walk records: leaves (kind 0) counted per iteration
k = it_bad - 1
n_before = sizes[:k].sum()
cnt = 0
out = []
for r in rec:
    if r[0] == 0 and r[1] == 0 and r[2] == 0:
        break
    if r[0] == 0:
        cnt += 1
    if n_before < cnt <= n_before + sizes[k] or (r[0] == 1 and n_before < cnt <= n_before + sizes[k]):
        out.append(r[:4])
print('oracle records of iter', k, '(kind, energy|depth, eps|dot_left, dot_right):')
print(np.array(out))
sel = np.nonzero(tr[:, 13] == k + 1)[0]
print('device:')
print(tr[sel[0] - 8: sel[0] + 1, :24])
