"""Refit glue either side of the sampler inside ``Recipe._sam_step`` / ``_pos_step``: choosing
the points at which the true model is evaluated for the next surrogate fit, with the ``logp_cutoff`` filter
(bayesfast/core/recipe.py:1060-1155), and the truncated importance weights of the post-processing step
(core/recipe.py:1270-1297).

Two forms of the selection:

* host arrays in, host arrays out (``select_fit_points`` on NumPy inputs): the reference's control flow around a user
  callable, with ``SystematicResampler`` doing the picking;
* device-resident and sharded (``select_fit_points`` on a ``TraceTuple``; ``select_rows_sharded``): every rank keeps
  only its own chains' samples on its GPU.  The resampler needs n order statistics of the logq of ALL chains
  (``np.argsort(logq)[ranks]``, utils/misc.py:108).  They are found without moving the samples: each rank sorts its
  shard once (``bfhip_sort_keys``), the ranks bisect together on the 64-bit order-preserving keys -- per step one
  ``bfhip_count_keys`` on the local shard and one all-reduce of n counters -- ties are assigned in global index order
  from an all-gather of per-rank tie counts, and only the n selected rows cross the links, as one all-reduce of an
  (n, d + 1) array in which every row has exactly one non-zero contributor (SURVEY section 8e option ii: 2.3 MB at
  n = 4290, d = 64, instead of the 25 GB of all samples).  The result is bit-identical to the single-rank selection.
"""
import warnings

import numpy as np

from ..utils.resample import SystematicResampler
from .. import parallel

__all__ = ['select_fit_points', 'importance_weights', 'select_rows_sharded', 'device_argsort', 'device_sort']

_SIGN = -2**63  # int64 with only the top bit set: uint64 key <-> int64 with the same order


def _ctx_of(t):
    from ..device import get_context
    return get_context(t.device.index)


def device_sort(a):
    """Stable ascending sort of a 1-d float64 device tensor: (keys_sorted as order-preserving int64, order int64)."""
    import torch
    from .. import _lib
    from ..device import _ptr
    if not a.is_cuda:
        raise RuntimeError('device_sort needs a tensor on the GPU (there is no CPU fallback).')
    a = a.contiguous().to(torch.float64)
    ctx = _ctx_of(a)
    keys = torch.empty(a.shape[0], dtype=torch.int64, device=a.device)
    order = torch.empty(a.shape[0], dtype=torch.int64, device=a.device)
    _lib.check(ctx._lib.bfhip_sort_keys(ctx.handle, a.shape[0], _ptr(a), _ptr(keys), _ptr(order)))
    return keys ^ _SIGN, order  # (uint64 bit patterns -> signed values with the same order)


def device_argsort(a):
    return device_sort(a)[1]


def _device_count(keys_sorted_signed, q_signed, upper):
    """#{keys < q} (upper False) or #{keys <= q} (upper True) for every query, on the device (bfhip_count_keys)."""
    import torch
    from .. import _lib
    from ..device import _ptr
    if not keys_sorted_signed.is_cuda:
        raise RuntimeError('the sharded selection needs device tensors (there is no CPU fallback).')
    ctx = _ctx_of(keys_sorted_signed)
    ku = (keys_sorted_signed ^ _SIGN).contiguous()
    qu = (q_signed ^ _SIGN).contiguous()
    out = torch.empty(q_signed.shape[0], dtype=torch.int64, device=q_signed.device)
    _lib.check(ctx._lib.bfhip_count_keys(ctx.handle, ku.shape[0], _ptr(ku), qu.shape[0], _ptr(qu), int(bool(upper)), _ptr(out)))
    return out


def select_rows_sharded(local_values, local_rows, ranks, sort_fn=None, count_fn=None, stats=None):
    """Rows of the globally ``ranks``-th smallest values (stable order: value, then global index = rank-major position).

    local_values (n_loc,) float64 and local_rows (n_loc, k) float64: this rank's shard, in global index order across
    ranks; ranks (n,) int64 global 0-based ranks, identical on every rank.  Returns (rows (n, k), values (n,)) identical
    on every rank.  ``sort_fn`` / ``count_fn`` default to the device kernels (the CPU tests inject torch stand-ins to
    exercise the collective logic under gloo).  ``stats``, if a dict, receives the bytes this rank put on the wire."""
    import torch
    import torch.distributed as dist
    sort_fn = sort_fn or device_sort
    count_fn = count_fn or _device_count
    rank, ws = parallel.world()
    dev = local_values.device
    ranks = torch.as_tensor(np.asarray(ranks, dtype=np.int64), device=dev)
    n = ranks.shape[0]
    keys, order = sort_fn(local_values)
    wire = 0
    if ws == 1:
        idx = order[ranks]
        if stats is not None:
            stats['wire_bytes'] = 0
        return local_rows[idx], local_values[idx]

    def allsum(t):
        nonlocal wire
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        wire += t.numel() * t.element_size()
        return t

    # 1. the key of every requested rank: smallest K with #{keys <= K} >= rank + 1, by bisection on the int64 key space
    lo = torch.full((n,), -2**63, dtype=torch.int64, device=dev)
    hi = torch.full((n,), 2**63 - 1, dtype=torch.int64, device=dev)
    for _ in range(64):
        mid = (lo & hi) + ((lo ^ hi) >> 1)  # floor((lo + hi) / 2) without overflow
        cnt = allsum(count_fn(keys, mid, True))
        ok = cnt >= ranks + 1
        hi = torch.where(ok, mid, hi)
        lo = torch.where(ok, lo, mid + 1)
    key = lo
    # 2. ties: the element is number t = rank - #{keys < K} among the elements equal to K, counted in global index order
    c_lt_loc = count_fn(keys, key, False)
    c_le_loc = count_fn(keys, key, True)
    t = ranks - allsum(c_lt_loc.clone())
    ties = [torch.empty_like(c_lt_loc) for _ in range(ws)]
    dist.all_gather(ties, (c_le_loc - c_lt_loc).contiguous())
    wire += n * 8 * ws
    before = torch.zeros_like(t)
    mine = torch.zeros_like(t, dtype=torch.bool)
    off = torch.zeros_like(t)
    for r in range(ws):
        owns = (t >= before) & (t < before + ties[r])
        if r == rank:
            mine, off = owns, t - before
        before = before + ties[r]
    # 3. only the selected rows travel: every row has exactly one owner, the others add zeros
    k = local_rows.shape[1]
    out = torch.zeros((n, k + 1), dtype=torch.float64, device=dev)
    sel = mine.nonzero().reshape(-1)
    if sel.numel():
        idx = order[(c_lt_loc + off)[sel]]
        out[sel, :k] = local_rows[idx]
        out[sel, k] = local_values[idx]
    allsum(out)
    if stats is not None:
        stats['wire_bytes'] = wire
    return out[:, :k], out[:, k]


def _resampled_rows(source, resampler, n, stats=None):
    """n resampled (x, logq) rows of the previous round, from a TraceTuple (device-resident, sharded) or host arrays."""
    from ..samplers.sample_trace import TraceTuple
    if isinstance(source, TraceTuple):
        x_loc, logq_loc, n_total = source.refit_shard()
        ranks = resampler.ranks(n_total, n)
        rows, vals = select_rows_sharded(logq_loc, x_loc, ranks, stats=stats)
        return rows.cpu().numpy(), vals.cpu().numpy()
    prev_samples, prev_logq = source
    i = resampler(prev_logq, n)
    return prev_samples[i], prev_logq[i]


def select_fit_points(prev_samples, prev_logq=None, logp_true=None, n_eval=None, resampler=None, logp_cutoff=True,
                      alpha_min=0.75, alpha_supp=1.25, stats=None):
    """Points and true log-densities for the next ``Density.fit``.

    prev_samples (N, d), prev_logq (N,): the previous round's samples (original space) and the surrogate log-density
    they were drawn from -- or ``prev_samples`` is the ``TraceTuple`` of the previous ``sample`` call (``prev_logq``
    None): the rows then stay on the GPUs that produced them and only the selected ones are exchanged (module
    docstring).  ``logp_true(x (k, d)) -> (k,)`` evaluates the true model; ``n_eval`` points are picked by
    ``resampler`` (``SystematicResampler`` by default, core/recipe.py:1074-1075).

    With ``logp_cutoff`` (recipe.py:1097-1155) points whose true logp falls below the smallest logq among the
    resampled points are dropped, and supplementary points are drawn until ``alpha_min * n_eval`` good points
    remain (``alpha_supp`` oversamples each supplement).  Returns ``(x_fit, logp_fit, n_true_evaluations)``."""
    from ..samplers.sample_trace import TraceTuple
    if isinstance(prev_samples, TraceTuple):
        source = prev_samples
        n_avail = source.n_refit_rows()
    else:
        prev_samples = np.asarray(prev_samples, dtype=np.float64)
        prev_logq = np.asarray(prev_logq, dtype=np.float64).reshape(-1)
        if prev_samples.ndim != 2 or prev_samples.shape[0] != prev_logq.size:
            raise ValueError('prev_samples (N, d) and prev_logq (N,) do not match.')
        source = (prev_samples, prev_logq)
        n_avail = prev_samples.shape[0]
    n_eval = int(n_eval)
    if n_eval <= 0:
        raise ValueError('n_eval should be a positive int.')
    if n_avail < n_eval:  # recipe.py:1060-1065
        raise RuntimeError('I need {} points to fit the surrogate model, but I can find at most {} points in the '
                           'previous step.'.format(n_eval, n_avail))
    if resampler is None:
        resampler = SystematicResampler()
    x_fit, logq_fit = _resampled_rows(source, resampler, n_eval, stats)
    logp_fit = np.asarray(logp_true(x_fit), dtype=np.float64).reshape(-1)
    n_calls = x_fit.shape[0]
    if not logp_cutoff:
        return x_fit, logp_fit, n_calls
    logq_min = np.min(logq_fit)
    is_good = logp_fit > logq_min
    f_good = np.sum(is_good) / logp_fit.size
    if f_good < 0.5:
        warnings.warn('more than half of the samples are abandoned because their logp < logq_min.', RuntimeWarning)
    if f_good == 0.:
        raise RuntimeError('f_good is 0, indicating that the samples seem very bad. Please check your recipe setup. '
                           'You may also want to try logp_cutoff=False for the SampleStep.')
    x_fit, logp_fit = x_fit[is_good], logp_fit[is_good]
    n_eval_min = int(alpha_min * n_eval)
    # (the reference calls np.delete without keeping the result, so resampled points stay in the pool; same here)
    while x_fit.shape[0] < n_eval_min:
        n_supp = max(int((n_eval_min - x_fit.shape[0]) / f_good * alpha_supp), 4)
        if n_avail < n_supp:
            raise RuntimeError('I do not have enough supplementary points.')
        x_supp, _ = _resampled_rows(source, resampler, n_supp)
        logp_supp = np.asarray(logp_true(x_supp), dtype=np.float64).reshape(-1)
        n_calls += x_supp.shape[0]
        good = logp_supp > logq_min
        if np.sum(good) < logp_supp.size / 2:
            warnings.warn('more than half of the samples are abandoned because their logp < logq_min.', RuntimeWarning)
        x_fit = np.concatenate((x_fit, x_supp[good]))
        logp_fit = np.concatenate((logp_fit, logp_supp[good]))
    return x_fit, logp_fit, n_calls


def importance_weights(logp, logq, k_trunc=0.25):
    """Truncated importance weights of ``PostStep`` (core/recipe.py:1289-1296): ``w = exp(logp - logq)`` clipped at
    ``mean(w) * n ** k_trunc`` (no clipping for ``k_trunc < 0``).  Returns ``(weights, weights_trunc)``: device tensors
    for device tensors (``bfhip_importance_weights``), NumPy arrays for host arrays (as in the reference)."""
    if hasattr(logp, 'is_cuda') and logp.is_cuda:
        import torch
        from .. import _lib
        from ..device import _ptr
        logp = logp.contiguous().to(torch.float64).reshape(-1)
        logq = logq.to(logp.device).contiguous().to(torch.float64).reshape(-1)
        if logp.shape != logq.shape:
            raise ValueError('logp and logq should have the same size.')
        ctx = _ctx_of(logp)
        w, wt = torch.empty_like(logp), torch.empty_like(logp)
        _lib.check(ctx._lib.bfhip_importance_weights(ctx.handle, logp.shape[0], _ptr(logp), _ptr(logq), float(k_trunc), _ptr(w),
                                                     _ptr(wt)))
        return w, wt
    logp = np.asarray(logp, dtype=np.float64).reshape(-1)
    logq = np.asarray(logq, dtype=np.float64).reshape(-1)
    if logp.shape != logq.shape:
        raise ValueError('logp and logq should have the same size.')
    weights = np.exp(logp - logq)
    if k_trunc < 0:
        return weights, weights.copy()
    return weights, np.clip(weights, 0, np.mean(weights) * logp.size**k_trunc)
