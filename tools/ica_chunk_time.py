"""Tuning: the device-resident FastICA chunk (transforms/ica.py: _ChunkState) at SIT's size -- the chunk eager and as a HIP graph, and
its pieces one by one.  usage: python3 tools/ica_chunk_time.py [n] [d] [gaussian]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesfast_amd.device import get_context, _ptr
from bayesfast_amd import _lib
from bayesfast_amd.transforms import ica
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ctx = get_context(0)
rng = np.random.default_rng(0)
x = rng.normal(size=(n, d)) if len(sys.argv) > 3 else rng.laplace(size=(n, d))
st = ica._ChunkState(ctx, n, d, torch.device('cuda', 0))
st.x1[:n].copy_(ctx.tensor(x))
W0 = ctx.tensor(np.linalg.qr(rng.normal(size=(d, d)))[0])


def wall(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


st.W.copy_(W0)
print('eager chunk of %d iterations: %.2f ms' % (ica._CHUNK, wall(st.chunk)))
st.warm = True
st.run_chunk()
print('graph captured:', st.graph is not None)
print('graph replay: %.2f ms' % wall(st.graph.replay))
print('meas', st.meas.cpu().numpy()[:, :4])
lib, h = ctx._lib, ctx.handle
res = st.work[-1:]
Yb = st.Y.view(st.nb, ica._ROWS, d).transpose(1, 2)
Xb = st.x1.view(st.nb, ica._ROWS, d)
pieces = [('mm', lambda: torch.mm(st.x1, st.W.T, out=st.Y)),
          ('tanh', lambda: _lib.check(lib.bfhip_ica_tanh(h, st.n, st.n_pad, d, _ptr(st.Y), _ptr(st.partial)))),
          ('bmm', lambda: torch.bmm(Yb, Xb, out=st.P)),
          ('assemble', lambda: _lib.check(lib.bfhip_ica_assemble(h, d, st.nb, _ptr(st.P), st.n, st.n_pad, _ptr(st.partial), _ptr(st.W), _ptr(st.A), _ptr(st.meas[0, 0:])))),
          ('polar', lambda: _lib.check(lib.bfhip_polar_ns(h, d, _ptr(st.A), _ptr(st.W1), ica._NS_ITERS, _ptr(st.work), _ptr(res)))),
          ('post', lambda: _lib.check(lib.bfhip_ica_post(h, d, _ptr(st.W1), _ptr(st.W), _ptr(res), 0, ica._CHUNK, _ptr(st.Wbuf), _ptr(st.meas))))]
for name, f in pieces:
    print('%-9s %.1f us' % (name, wall(f, 20) * 1e3))
# replay + the host's look at the measures, as _ica_par_device alternates them
for label, f in (('replay; meas.cpu()', lambda: (st.graph.replay(), st.meas.cpu())),
                 ('W.clone(); replay; meas.cpu()', lambda: (st.W.clone(), st.graph.replay(), st.meas.cpu())),
                 ('replay; synchronize', lambda: (st.graph.replay(), torch.cuda.synchronize())),
                 ('eager chunk; meas.cpu()', lambda: (st.chunk(), st.meas.cpu()))):
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    print('%-32s %.2f ms each' % (label, (time.perf_counter() - t0) / 10 * 1e3))
# two seconds of replay + look: the individual times (stalls that end on a 100 ms grid were seen inside GBS runs)
ts, t_end = [], time.perf_counter() + 2.
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    st.graph.replay()
    st.meas.cpu()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print('%d replays in 2 s: median %.2f ms, %d above 8 ms: %s' % (ts.size, np.median(ts), (ts > 8).sum(), np.round(ts[ts > 8], 1)))
