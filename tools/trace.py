"""Tuning: timeline of one wave's trips (needs a tools/variant.sh build with -DBF_TRACE=64)."""
import sys, os, ctypes as C
os.environ.setdefault("BFHIP_NUTS_KERNEL", "sliced")  # this timeline is the sliced kernel's (tools/trace_pipe.py: pipelined)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, DeviceDensity
from bayesfast_amd.chains import DeviceChains
from bayesfast_amd.workloads import correlated_gaussian_spec
from bayesfast_amd import _lib
N = int(os.environ.get('NTRACE', 64))
SAMPLER = os.environ.get('SAMPLER', 'NUTS')
K_ACT = int(os.environ.get('K_ACT', 1))
ctx = get_context(0)
spec, _ = correlated_gaussian_spec(64)
C_ = 4096
dc = DeviceChains(DeviceDensity(spec, ctx), np.random.default_rng(1).normal(size=(C_, 64)), seed=1)
dc.run(200, SAMPLER, n_warmup=200)
if K_ACT < 16:
    parked = (torch.arange(C_, device='cuda') % 16) >= K_ACT
    dc.sc[parked, _lib.SC_FIELDS.index('i_iter')] = 1e9
buf = torch.zeros((N * 16,), dtype=torch.int64, device='cuda')
L = _lib.lib()
L.bfhip_debug_stamps.argtypes = [C.c_void_p]
L.bfhip_debug_stamps(C.c_void_p(buf.data_ptr()))
dc.run(20, SAMPLER, n_warmup=200)
L.bfhip_debug_stamps(None)
t = buf.cpu().numpy().reshape(N, 16).astype(np.int64)
names = ['0 loop top', '1 A done (x posted)', '2 after B1', '3 alive checked', '4 MFMAs done', '5 GB written', '6 after B2',
         '7 partials ready', '8 reductions done', '9 C done', '10 unit done']
leaf_names = ['11 leaf: dE known', '12 leaf: weights (exp) done', '13 merge0: partial dots', '14 merge0: reduced', '15 merge0: draw done']
rows = []
for i in range(4, N - 1):
    tt = t[i].copy(); nxt = t[i + 1][0]
    if tt[0] == 0 or nxt == 0 or tt[7] == 0 or tt[4] == 0: continue   # need an evaluating trip with a job
    seq = [tt[k] for k in range(11)] + [nxt]
    rows.append(np.diff(seq))
rows = np.array(rows)
print('%d evaluating trips traced; mean ticks per segment (median in brackets):' % len(rows))
for k in range(11):
    print('  %-22s -> next: %7.0f  [%6.0f]' % (names[k], rows[:, k].mean(), np.median(rows[:, k])))
print('  trip total: %.0f [%.0f]' % (rows.sum(1).mean(), np.median(rows.sum(1))))

# finer points inside the NUTS leaf unit (slots 11-15), relative to '9 C done'
rows2 = []
for i in range(4, N - 1):
    tt = t[i]
    if tt[9] == 0 or tt[11] == 0 or tt[12] == 0: continue
    seq = [tt[9], tt[11], tt[12]] + ([tt[13], tt[14], tt[15]] if tt[13] and tt[14] and tt[15] else []) + [tt[10]]
    rows2.append((len(seq), np.diff(seq)))
for ln, lab in ((4, ['C done -> dE', 'dE -> weights', 'weights -> unit done (even leaf)']),
                (7, ['C done -> dE', 'dE -> weights', 'weights -> partial dots', 'partial dots -> reduced', 'reduced -> draw', 'draw -> unit done (odd leaf)'])):
    sel = np.array([r[1] for r in rows2 if r[0] == ln])
    if len(sel):
        print('leaf unit, %d trips:' % len(sel))
        for k, nm in enumerate(lab):
            print('  %-34s %7.0f [%6.0f]' % (nm, sel[:, k].mean(), np.median(sel[:, k])))

# raw timeline of consecutive trips (all kinds): ticks from the trip's loop top to every recorded point
if os.environ.get('RAW'):
    print('raw trips (ticks since loop top; 0 = point not reached):')
    for i in range(8, min(N - 1, 8 + int(os.environ['RAW']))):
        tt = t[i]
        if tt[0] == 0: continue
        rel = [(int(tt[k] - tt[0]) if tt[k] else 0) for k in range(16)]
        print('  trip %3d total %6d | ' % (i, int(t[i + 1][0] - tt[0]) if t[i + 1][0] else -1) + ' '.join('%5d' % r for r in rel[1:]))
