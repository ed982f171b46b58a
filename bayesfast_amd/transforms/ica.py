"""FastICA on the device, for the rotations of the Sliced Iterative Transform (transforms/sit.py:235-251).

The reference calls ``sklearn.decomposition.FastICA(**ica_options).fit(x[subset])`` on the host: at 64 dimensions and 20 000
points that is 100 fixed-point iterations of two (d x d)(d x n) products and a tanh over d x n numbers, 2.5 s of the 3.6 s a SIT
iteration took.  This is the same algorithm, statement by statement (scikit-learn 1.7: ``FastICA._fit_transform`` with its
defaults algorithm='parallel', fun='logcosh' (alpha = 1), whiten='unit-variance', whiten_solver='svd', tol 1e-4; ``_ica_par``,
``_sym_decorrelation``, ``_logcosh``), with the data on the GPU:

* centring, the Gram matrix of the whitening (``bfhip_gram``: the same MFMA kernel as the surrogate fit's normal equations)
  and the projections ``K X``, ``W X1``, ``g(W X1) X1^T`` as device products; tanh and its row means on the device;
* the d x d pieces -- the symmetric eigen-decompositions of the whitening and of every ``_sym_decorrelation``, the
  convergence test -- on the host in NumPy, as scikit-learn does them (32 KB across PCIe per iteration);
* ``w_init`` from ``np.random.RandomState(random_state).normal(size=(d, d))``, exactly the draw ``FastICA`` makes, so the
  same ``random_state`` starts both from the same matrix.

The singular vectors of the whitening come from the eigen-decomposition of the d x d Gram matrix instead of an SVD of the
d x n data (scikit-learn's 'eigh' solver; its 'svd' default differs by rounding), with the same sign convention.  Agreement
with ``sklearn.decomposition.FastICA`` on the same data and seed is tested to 1e-8 on well-conditioned data
(tests/test_evidence.py); the fixed point of a run that has not converged after ``max_iter`` iterations (the reference warns
about those) is as sensitive to rounding here as it is there.
"""
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
    p_ = float(n)
    n_iter, lim = 0, np.inf
    # (the host's share of an iteration is one d x d eigh: with the BLAS pool's threads spinning on a GPU host whose cgroup gives
    # the process a fraction of the cores it sees, 6.2 ms at d = 128 -- half of a config-5 GBS run -- against 1 ms on one thread)
    from ..utils.threads import blas_single_thread
    with blas_single_thread():
        for ii in range(int(max_iter)):
            wt = ctx.tensor(W.T.copy())
            gwtx = torch.tanh(x1 @ wt)                                   # (n, d) = g(W X1)^T, alpha = 1
            g_wtx = (1. - gwtx * gwtx).mean(0)                           # (d,)   mean of g'(W X1) over the samples
            both = torch.cat([_tn_product(gwtx, x1) / p_, g_wtx[None]], 0).cpu().numpy()   # one copy to the host
            W1 = _sym_decorrelation(both[:d] - both[d][:, None] * W)
            lim = np.max(np.abs(np.abs(np.einsum('ij,ij->i', W1, W)) - 1))
            W = W1
            n_iter = ii + 1
            if lim < tol:
                break
        else:
            warnings.warn('FastICA did not converge. Consider increasing tolerance or the maximum number of iterations.')
    # whiten='unit-variance': the sources get unit variance, the rows of W are scaled accordingly
    comp = W @ K
    s_std = ((xc @ ctx.tensor(comp.T.copy())).std(0, unbiased=False)).cpu().numpy()
    comp = comp / s_std[:, None]
    return comp, mean.cpu().numpy(), n_iter
