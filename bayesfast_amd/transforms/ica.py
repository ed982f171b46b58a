"""FastICA on the device, for the rotations of the Sliced Iterative Transform (transforms/sit.py:235-251).

The reference calls ``sklearn.decomposition.FastICA(**ica_options).fit(x[subset])`` on the host: at 64 dimensions and 20 000
points that is 100 fixed-point iterations of two (d x d)(d x n) products and a tanh over d x n numbers, 2.5 s of the 3.6 s a SIT
iteration took.  This is the same algorithm, statement by statement (scikit-learn 1.7: ``FastICA._fit_transform`` with its
defaults algorithm='parallel', fun='logcosh' (alpha = 1), whiten='unit-variance', whiten_solver='svd', tol 1e-4; ``_ica_par``,
``_sym_decorrelation``, ``_logcosh``), with the data on the GPU:

* centring, the Gram matrix of the whitening (``bfhip_gram``: the same MFMA kernel as the surrogate fit's normal equations)
  and the projections ``K X``, ``W X1``, ``g(W X1) X1^T`` as device products; tanh and its row means on the device;
* the fixed-point iteration is device-resident (round 6, ``_ica_par``): the symmetric decorrelation of every iteration is a
  Newton-Schulz polar iteration of d x d device products, the iterations run ahead in chunks (a HIP graph on the GPU) and the
  host reads the chunk's convergence measures once -- one synchronisation per ten iterations instead of one per iteration with
  a host ``eigh`` each; only the eigen-decomposition of the whitening (once per fit) is NumPy's;
* ``w_init`` from ``np.random.RandomState(random_state).normal(size=(d, d))``, exactly the draw ``FastICA`` makes, so the
  same ``random_state`` starts both from the same matrix.

The singular vectors of the whitening come from the eigen-decomposition of the d x d Gram matrix instead of an SVD of the
d x n data (scikit-learn's 'eigh' solver; its 'svd' default differs by rounding), with the same sign convention.  Agreement
with ``sklearn.decomposition.FastICA`` on the same data and seed is tested to 1e-8 on well-conditioned data
(tests/test_evidence.py); the fixed point of a run that has not converged after ``max_iter`` iterations (the reference warns
about those) is as sensitive to rounding here as it is there.
"""
import time
import warnings

import numpy as np

__all__ = ['fastica_device']


def _sym_decorrelation(w):
    """W <- (W W^T)^{-1/2} W  (scikit-learn ``_sym_decorrelation``)."""
    s, u = np.linalg.eigh(w @ w.T)
    s = np.clip(s, a_min=np.finfo(w.dtype).tiny, a_max=None)
    return np.linalg.multi_dot([u * (1. / np.sqrt(s)), u.T, w])


def _gram(ctx, xc):
    """X^T X (d, d) of the centred data (n, d) on the device: ``bfhip_gram``, the MFMA kernel of the surrogate fit."""
    from ..device import _ptr
    from .. import _lib
    n, d = xc.shape
    gram = ctx.empty((d, d))
    dummy_b, dummy_r = ctx.zeros((n, 1)), ctx.empty((d, 1))
    _lib.check(ctx._lib.bfhip_gram(ctx.handle, n, d, 1, _ptr(xc), d, _ptr(dummy_b), _ptr(gram), _ptr(dummy_r)))
    return gram


def _tn_product(a, b, rows=400):
    """a^T b for tall a, b (n, d): the sum over the n rows split into batches of ``rows`` (one batched product and a sum over
    the batches).  rocBLAS runs the plain (d x n)(n x d) product at 20 000 x 64 on a handful of workgroups: 1.07 ms against
    21 us this way -- it was 70 % of a FastICA iteration."""
    import torch
    n, d = a.shape
    m = (n // rows) * rows
    if m == 0:
        return a.T @ b
    out = torch.bmm(a[:m].view(m // rows, rows, d).transpose(1, 2), b[:m].view(m // rows, rows, b.shape[1])).sum(0)
    if m < n:
        out = out + a[m:].T @ b[m:]
    return out


# ---- the fixed-point iteration: host form (one round trip per iteration) and device-resident form -------------------------

def _ica_step_host(ctx, x1, W):
    """One iteration of scikit-learn's ``_ica_par`` with the d x d symmetric decorrelation on the host: (W1, lim)."""
    import torch
    d = W.shape[0]
    p_ = float(x1.shape[0])
    wt = ctx.tensor(W.T.copy())
    gwtx = torch.tanh(x1 @ wt)                                   # (n, d) = g(W X1)^T, alpha = 1
    g_wtx = (1. - gwtx * gwtx).mean(0)                           # (d,)   mean of g'(W X1) over the samples
    both = torch.cat([_tn_product(gwtx, x1) / p_, g_wtx[None]], 0).cpu().numpy()   # one copy to the host
    W1 = _sym_decorrelation(both[:d] - both[d][:, None] * W)
    lim = np.max(np.abs(np.abs(np.einsum('ij,ij->i', W1, W)) - 1))
    return W1, lim


_NS_ITERS = 60      # Newton-Schulz steps of the polar factor (covers singular values down to ~1e-5 of the largest; d^3 products of
                    # 20 us each at d = 128: the iteration costs about what the host's eigh did -- the gain is the missing round trips)
_NS_RESID = 1e-11   # ... accepted when max |X X^T - I| ends below this; otherwise the chunk is redone with the host's eigh
_CHUNK = 10         # iterations run ahead between two looks at the convergence test
GRAPH_STATS = {'captured': 0, 'failed': 0, 'replayed': 0, 'capture_s': 0., 'eager_s': 0., 'wait_s': 0.}   # chunk graphs of this process (tests and tools look at it)


def _polar_newton_schulz(A, eye, ctx=None, work=None):
    """(A A^T)^{-1/2} A -- scikit-learn's ``_sym_decorrelation``, the orthogonal polar factor of A -- by the Newton-Schulz iteration
    X <- 1.5 X - 0.5 X X^T X from X_0 = A / sqrt(|A|_1 |A|_inf) (singular values in (0, 1]: monotone, finally quadratic
    convergence to 1): no eigen-decomposition, no host round trip.  On the GPU ``bfhip_polar_ns`` (FP64-MFMA tiles, one wave per
    16 x 16 tile of a product: the 64 small rocBLAS products of the torch form below were 20 us each); the torch form serves CPU
    tensors (tests).  Returns (X, max |X X^T - I|)."""
    import torch
    if ctx is not None and A.is_cuda:
        from .. import _lib
        from ..device import _ptr
        d = A.shape[0]
        A = A.contiguous()
        X = torch.empty_like(A)
        if work is None:
            work = torch.empty((2 * d * d + _NS_ITERS + 11,), dtype=torch.float64, device=A.device)
        res = work[-1:]
        _lib.check(ctx._lib.bfhip_polar_ns(ctx.handle, d, _ptr(A), _ptr(X), _NS_ITERS, _ptr(work), _ptr(res)))
        return X, res[0].clone()
    s = torch.sqrt(A.abs().sum(0).max() * A.abs().sum(1).max())
    X = A / s
    for _ in range(_NS_ITERS):
        X = torch.addmm(X, X @ X.T, X, beta=1.5, alpha=-0.5)
    return X, (X @ X.T - eye).abs().max()


_ROWS = 400         # rows per batch of the G^T X1 product (see _tn_product)
_TIME_REPLAYS = False   # tools: HIP events around every replay
_REPLAY_EVENTS = []
_DEVICE_STATE = {}  # (device index, n, d) -> the chunk's buffers and its HIP graph (kept: a SIT fit calls FastICA once per iteration)


class _ChunkState:
    """Buffers of the device-resident iteration for one shape, and the chunk of ``_CHUNK`` iterations as a HIP graph."""

    def __init__(self, ctx, n, d, dev):
        import torch
        self.ctx, self.n, self.d = ctx, n, d
        self.n_pad = -(-n // _ROWS) * _ROWS                  # zero rows pad the last batch of the G^T X1 product
        self.nb = self.n_pad // _ROWS
        f = dict(dtype=torch.float64, device=dev)
        self.x1 = torch.zeros((self.n_pad, d), **f)
        self.Y = torch.empty((self.n_pad, d), **f)
        self.P = torch.empty((self.nb, d, d), **f)
        self.partial = torch.empty((-(-self.n_pad // 32), d), **f)
        self.A, self.W1, self.W = (torch.empty((d, d), **f) for _ in range(3))
        self.work = torch.empty((2 * d * d + _NS_ITERS + 11,), **f)    # bfhip_polar_ns: 2 d^2 + n_iter + 10, and the residual
        self.Wbuf = torch.empty((_CHUNK, d, d), **f)
        self.meas = torch.zeros((2, _CHUNK), **f)
        self.graph = None
        self.warm = False

    def chunk(self):
        """``_CHUNK`` iterations: two library products, three glue kernels and the polar kernel each -- no copies, no allocations."""
        import torch
        from .. import _lib
        from ..device import _ptr
        ctx, lib, d, h = self.ctx, self.ctx._lib, self.d, self.ctx.handle
        Yb = self.Y.view(self.nb, _ROWS, d).transpose(1, 2)
        Xb = self.x1.view(self.nb, _ROWS, d)
        res = self.work[-1:]
        for k in range(_CHUNK):
            torch.mm(self.x1, self.W.T, out=self.Y)                                               # Y = X1 W^T
            _lib.check(lib.bfhip_ica_tanh(h, self.n, self.n_pad, d, _ptr(self.Y), _ptr(self.partial)))   # G = tanh(Y), sums of g'
            torch.bmm(Yb, Xb, out=self.P)                                                          # G^T X1 by row batches
            _lib.check(lib.bfhip_ica_assemble(h, d, self.nb, _ptr(self.P), self.n, self.n_pad, _ptr(self.partial), _ptr(self.W),
                                              _ptr(self.A), _ptr(self.meas[0, k:])))
            _lib.check(lib.bfhip_polar_ns(h, d, _ptr(self.A), _ptr(self.W1), _NS_ITERS, _ptr(self.work), _ptr(res)))
            _lib.check(lib.bfhip_ica_post(h, d, _ptr(self.W1), _ptr(self.W), _ptr(res), k, _CHUNK, _ptr(self.Wbuf), _ptr(self.meas)))

    def run_chunk(self):
        import torch
        t_0 = time.perf_counter()
        if not self.warm:                       # the first chunk of a shape in this process: eager (the libraries' workspaces)
            self.chunk()
            self.warm = True
            GRAPH_STATS['eager_s'] += time.perf_counter() - t_0
            return
        if self.graph is None and GRAPH_STATS['failed'] <= 2:
            keep = self.ctx.stream
            saved = self.W.clone()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    self.ctx.set_stream(torch.cuda.current_stream(self.x1.device))   # (the library's kernels: nodes of the graph)
                    self.chunk()
            except Exception:
                g = None
            finally:
                self.ctx.set_stream(keep)
            self.W.copy_(saved)
            GRAPH_STATS['captured' if g is not None else 'failed'] += 1
            GRAPH_STATS['capture_s'] += time.perf_counter() - t_0
            self.graph = g
        if self.graph is not None:
            if _TIME_REPLAYS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.graph.replay()
                e1.record()
                _REPLAY_EVENTS.append((e0, e1))
            else:
                self.graph.replay()
            GRAPH_STATS['replayed'] += 1
        else:
            self.chunk()


def _ica_par_device(ctx, x1, W, max_iter, tol):
    """``_ica_par`` on the GPU: see ``_ica_par``.  The chunk's buffers and graph live in ``_DEVICE_STATE`` per shape."""
    import torch
    n, d = x1.shape
    key = (x1.device.index, n, d)
    st = _DEVICE_STATE.get(key)
    if st is None:
        if len(_DEVICE_STATE) >= 4:             # (a handful of shapes at most: each holds ~3 copies of the data)
            _DEVICE_STATE.pop(next(iter(_DEVICE_STATE)))
        st = _DEVICE_STATE[key] = _ChunkState(ctx, n, d, x1.device)
    st.ctx = ctx
    st.x1[:n].copy_(x1)
    st.W.copy_(torch.as_tensor(W, dtype=torch.float64, device=x1.device))
    n_iter = 0
    while n_iter < max_iter:
        start = st.W.clone()
        st.run_chunk()
        t_0 = time.perf_counter()
        m = st.meas.cpu().numpy()
        GRAPH_STATS['wait_s'] += time.perf_counter() - t_0
        left = min(_CHUNK, max_iter - n_iter)
        if not np.all(m[1, :left] < _NS_RESID) or not np.all(np.isfinite(m[:, :left])):
            # (rare) the polar iteration fell short somewhere in this chunk: the same iterations with the host's eigh
            Wh = start.cpu().numpy()
            for k in range(left):
                Wh, lim = _ica_step_host(ctx, x1, Wh)
                n_iter += 1
                if lim < tol:
                    return Wh, n_iter, True
            st.W.copy_(torch.as_tensor(Wh, dtype=torch.float64, device=x1.device))
            continue
        hit = np.flatnonzero(m[0, :left] < tol)
        if hit.size:
            k = int(hit[0])
            return st.Wbuf[k].cpu().numpy(), n_iter + k + 1, True
        if left < _CHUNK:   # max_iter fell inside the chunk: the iterate it ends on
            return st.Wbuf[left - 1].cpu().numpy(), max_iter, False
        n_iter += _CHUNK
    return st.W.cpu().numpy(), n_iter, False


def _ica_par(ctx, x1, W, max_iter, tol):
    """scikit-learn's ``_ica_par`` (W numpy (d, d), x1 (n, d) white data on the device): (W, n_iter, converged).

    Device-resident (round 6): the iterations run ahead in chunks of ``_CHUNK`` -- products, tanh, the symmetric decorrelation
    as a Newton-Schulz polar iteration, the convergence measure, every iterate kept -- and the host looks at the chunk's
    convergence measures ONCE (one synchronisation per chunk instead of one per iteration plus a host eigh each: 612 of them were
    0.57 s of a config-5 GBS run, profiles/r05b_evidence_profile.log); it stops at the FIRST iterate below ``tol`` exactly as
    the sequential loop does, so the iteration count is scikit-learn's.  A chunk whose polar iteration did not reach ``_NS_RESID`` is
    redone with the host's eigen-decomposition (``_ica_step_host``).  On the GPU (``_ica_par_device``) an iteration is two library
    products, three glue kernels (``bfhip_ica_tanh / _assemble / _post``) and ``bfhip_polar_ns``, and the chunk is ONE HIP graph
    captured once per shape with the library's context pointed at the capturing stream (the ~25 framework launches of an iteration
    were 0.7 ms of host time, and the device-to-device copies among them stalled the graph's replay).  CPU tensors (the tests'
    stand-in context) take the framework form below."""
    import torch
    from ..utils.threads import blas_single_thread
    if x1.device.type == 'cuda' and hasattr(ctx, 'handle'):
        return _ica_par_device(ctx, x1, W, max_iter, tol)
    d = W.shape[0]
    p_ = float(x1.shape[0])
    dev = x1.device
    Wd = torch.as_tensor(W, dtype=torch.float64, device=dev).clone()
    eye = torch.eye(d, dtype=torch.float64, device=dev)
    Wbuf = torch.empty((_CHUNK, d, d), dtype=torch.float64, device=dev)
    meas = torch.zeros((2, _CHUNK), dtype=torch.float64, device=dev)   # convergence measure and polar residual of every iterate

    def chunk(n_it=_CHUNK):
        for k in range(n_it):
            gwtx = torch.tanh(x1 @ Wd.T)
            g_wtx = (1. - gwtx * gwtx).mean(0)
            W1, res = _polar_newton_schulz(_tn_product(gwtx, x1) / p_ - g_wtx[:, None] * Wd, eye)
            meas[0, k] = ((W1 * Wd).sum(1).abs() - 1.).abs().max()
            meas[1, k] = res
            Wbuf[k].copy_(W1)
            Wd.copy_(W1)

    n_iter = 0
    with blas_single_thread():
        while n_iter < max_iter:
            start = Wd.clone()
            chunk()
            m = meas.cpu().numpy()
            left = min(_CHUNK, max_iter - n_iter)
            if not np.all(m[1, :left] < _NS_RESID) or not np.all(np.isfinite(m[:, :left])):
                Wh = start.cpu().numpy()
                for k in range(left):
                    Wh, lim = _ica_step_host(ctx, x1, Wh)
                    n_iter += 1
                    if lim < tol:
                        return Wh, n_iter, True
                Wd.copy_(torch.as_tensor(Wh, dtype=torch.float64, device=dev))
                continue
            hit = np.flatnonzero(m[0, :left] < tol)
            if hit.size:
                k = int(hit[0])
                return Wbuf[k].cpu().numpy(), n_iter + k + 1, True
            if left < _CHUNK:   # max_iter fell inside the chunk: the iterate it ends on
                return Wbuf[left - 1].cpu().numpy(), max_iter, False
            n_iter += _CHUNK
    return Wd.cpu().numpy(), n_iter, False


def fastica_device(x, random_state=None, max_iter=200, tol=1e-4, w_init=None, ctx=None):
    """``FastICA(max_iter=..., tol=..., random_state=...).fit(x)`` for x (n, d) (array or device tensor).

    Returns (components_ (d, d), mean_ (d,), n_iter) as NumPy arrays, like the fitted estimator's attributes."""
    import torch
    if ctx is None:
        from ..device import get_context
        ctx = get_context()
    xt = ctx.tensor(x, torch.float64)
    n, d = xt.shape
    if n < 2:
        raise ValueError('FastICA needs at least two samples.')
    mean = xt.mean(0)
    xc = (xt - mean).contiguous()                                   # (n, d): XT^T of scikit-learn
    # whitening: singular values / left vectors of XT (d, n) from the Gram matrix XT XT^T (MFMA)
    ev, u = np.linalg.eigh(_gram(ctx, xc).cpu().numpy())
    order = np.argsort(ev)[::-1]
    eps = np.finfo(np.float64).eps * 10
    if np.any(ev < eps):
        warnings.warn('There are some small singular values in the whitening of FastICA.')
    sv = np.sqrt(np.where(ev < eps, eps, ev))[order]
    u = u[:, order]
    u = u * np.sign(u[0])                                            # consistent eigenvectors, as scikit-learn
    K = (u / sv).T                                                   # (d, d), see (6.33) p.140
    x1 = (xc @ ctx.tensor(K.T.copy())) * np.sqrt(n)                  # (n, d) = (K XT)^T sqrt(n): white data, points as rows
    if w_init is None:
        rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
        w_init = np.asarray(rs.normal(size=(d, d)), dtype=np.float64)
    else:
        w_init = np.asarray(w_init, dtype=np.float64)
        if w_init.shape != (d, d):
            raise ValueError('w_init has invalid shape -- should be {}'.format((d, d)))
    # _ica_par: parallel FastICA with the logcosh contrast
    W = _sym_decorrelation(w_init)
    W, n_iter, converged = _ica_par(ctx, x1, W, int(max_iter), float(tol))
    if not converged:
        warnings.warn('FastICA did not converge. Consider increasing tolerance or the maximum number of iterations.')
    # whiten='unit-variance': the sources get unit variance, the rows of W are scaled accordingly
    comp = W @ K
    s_std = ((xc @ ctx.tensor(comp.T.copy())).std(0, unbiased=False)).cpu().numpy()
    comp = comp / s_std[:, None]
    return comp, mean.cpu().numpy(), n_iter
