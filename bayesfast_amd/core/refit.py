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
  shard once (``bfhip_sort_keys``); the ranks exchange local quantile keys (all-gather), count their shards against all
  of them (``bfhip_count_keys``, one all-reduce), exchange the few candidates each requested rank is then pinned to
  (all-gather; ties in global index order from per-rank counts), and only the n selected rows cross the links, as one
  all-reduce of an (n, d + 1) array in which every row has exactly one non-zero contributor (SURVEY section 8e option
  ii): four collectives, no host synchronisation in between.  The result is bit-identical to the single-rank selection.
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


def select_rows_sharded(local_values, local_rows, ranks, sort_fn=None, count_fn=None, stats=None, n_splitter=None, n_loc_max=None):
    """Rows of the globally ``ranks``-th smallest values (stable order: value, then global index = rank-major position).

    local_values (n_loc,) float64 and local_rows (n_loc, k) float64: this rank's shard, in global index order across
    ranks; ranks (n,) int64 global 0-based ranks, identical on every rank.  Returns (rows (n, k), values (n,)) identical
    on every rank.  ``sort_fn`` / ``count_fn`` default to the device kernels (the CPU tests inject torch stand-ins to
    exercise the collective logic under gloo).  ``stats``, if a dict, receives the bytes this rank put on the wire and the
    number of collectives.

    A splitter exchange, FOUR collectives whatever the sizes, nothing in between waits for the host:
      1. all-gather of S local quantile keys per rank (positions spread evenly over the rank's sorted shard, ends included);
      2. all-reduce of the global counts #{keys < b}, #{keys <= b} for every gathered key b.  Between two neighbouring
         boundaries a rank holds fewer elements than its own splitter spacing g (none of its splitters falls inside), so a
         requested rank r is now pinned to an interval with at most g candidates per rank -- or to a boundary value itself;
      3. all-gather of those candidates, (n, g) keys per rank, with the per-rank counts of the boundary (ties in global
         index order are assigned from them);
      4. all-reduce of the (n, k + 1) selected rows, every row contributed by its one owner.
    ``n_loc_max``: the row count of the largest shard (``_resampled_rows`` knows it); without it, it is read from the first
    gather's header (one small device-to-host read).  S defaults to the value that minimises the bytes,
    sqrt(n * n_loc_max / 2): at the headline (n = 4290, 4.1 M rows per rank,
    8 ranks) 94 k splitters, 6 + 12 + 1.5 + 2.2 MB per rank instead of the 25 GB of all samples, and 4 collectives
    instead of the 68 of a bisection over the 64-bit key space."""
    import torch
    sort_fn = sort_fn or device_sort
    count_fn = count_fn or _device_count
    rank, ws = parallel.world()
    dev = local_values.device
    ranks_host = np.asarray(ranks.cpu() if hasattr(ranks, 'cpu') else ranks, dtype=np.int64)
    ranks = torch.as_tensor(ranks_host, device=dev)
    n = ranks.shape[0]
    keys, order = sort_fn(local_values)
    if ws == 1:
        idx = order[ranks]
        if stats is not None:
            stats['wire_bytes'] = 0
            stats['collectives'] = 0
        return local_rows[idx], local_values[idx]
    wire, n_coll = 0, 0
    n_loc = int(keys.shape[0])
    i64max = 2**63 - 1

    def gather(t):
        nonlocal wire, n_coll
        wire += t.numel() * t.element_size() * ws
        n_coll += 1
        return parallel.all_gather_stack(t)

    def allsum(t):
        nonlocal wire, n_coll
        wire += t.numel() * t.element_size()
        n_coll += 1
        return parallel.all_reduce_sum(t)

    # 1. splitters.  S and the candidate width g must be the same on every rank without asking: S is a pure function of
    # (n, n_loc_max), g the splitter spacing of the largest shard
    if n_loc_max is None:
        n_loc_max = -(-(int(np.max(ranks_host)) + 1) // ws) if n > 0 else 1   # (sizes all ranks can derive: ceil(n_total' / ws))
        exact_sizes = False
    else:
        exact_sizes = True
    S = int(n_splitter) if n_splitter is not None else int(max(64, min(2**20, round((max(n, 1) * max(n_loc_max, 1) / 2.)**0.5))))
    if n_loc > 0:
        pos = torch.div(torch.arange(S, device=dev, dtype=torch.int64) * (n_loc - 1), max(S - 1, 1), rounding_mode='floor')
        spl = keys[pos]
    else:
        spl = torch.full((S,), i64max, dtype=torch.int64, device=dev)
    got = gather(torch.cat([torch.tensor([n_loc], dtype=torch.int64, device=dev), spl]))
    if not exact_sizes:   # (the caller did not say how large the largest shard is: read it from the gathered header)
        n_loc_max = int(got[:, 0].max())
    g = -(-max(int(n_loc_max) - 1, 0) // max(S - 1, 1)) + 1   # more than any rank holds between two of its own splitters
    B = torch.sort(got[:, 1:].reshape(-1)).values             # (ws * S,) boundaries, ascending (duplicates are harmless)
    # 2. global counts below / up to every boundary
    c_lt_loc, c_le_loc = count_fn(keys, B, False), count_fn(keys, B, True)
    cnt = allsum(torch.stack([c_lt_loc, c_le_loc]))
    cnt_lt, cnt_le = cnt[0], cnt[1]
    # the interval of every requested rank: first boundary j with #{keys <= B_j} >= r + 1
    j = torch.searchsorted(cnt_le, ranks + 1, right=False).clamp_(max=B.shape[0] - 1)
    jm = (j - 1).clamp_(min=0)
    base = torch.where(j > 0, cnt_le[jm], torch.zeros_like(ranks))          # #{keys <= B_{j-1}}
    lo_loc = torch.where(j > 0, c_le_loc[jm], torch.zeros_like(ranks))      # my candidates: sorted positions [lo_loc, hi_loc)
    hi_loc = c_lt_loc[j]
    in_open = ranks < cnt_lt[j]                                              # else the element equals B_j (a tie run)
    # 3. candidates of the open intervals + my count of elements equal to the boundary
    ar = torch.arange(g, device=dev, dtype=torch.int64)
    cpos = lo_loc[:, None] + ar[None, :]
    valid = cpos < hi_loc[:, None]
    ckey = torch.where(valid, keys[cpos.clamp(max=max(n_loc - 1, 0))] if n_loc > 0 else torch.full_like(cpos, i64max),
                       torch.full_like(cpos, i64max))
    payload = torch.cat([ckey, valid.to(torch.int64).sum(1, keepdim=True), (c_le_loc[j] - c_lt_loc[j])[:, None]], 1)
    allp = gather(payload)                                                   # (ws, n, g + 2)
    ck = allp[:, :, :g].permute(1, 0, 2).reshape(n, ws * g)                  # rank-major, then local order: global index order
    ntie = allp[:, :, g + 1]                                                 # (ws, n) elements equal to B_j per rank
    # open interval: the t-th smallest candidate in (key, rank, local position) order.  Valid candidates lie strictly below
    # B_j, hence below the int64-max padding, and a stable sort keeps equal keys in global index order
    t_open = (ranks - base).clamp_(min=0, max=ws * g - 1)
    pick = torch.gather(torch.sort(ck, dim=1, stable=True).indices, 1, t_open[:, None]).reshape(-1)
    own_open, idx_open = torch.div(pick, g, rounding_mode='floor'), pick % g
    # tie run: element number t_tie among the elements equal to B_j, in rank-major order
    t_tie = (ranks - cnt_lt[j]).clamp_(min=0)
    cum = torch.cumsum(ntie, 0)                                              # (ws, n)
    own_tie = (cum <= t_tie[None, :]).sum(0).clamp_(max=ws - 1)
    before = torch.where(own_tie > 0, torch.gather(cum, 0, (own_tie - 1).clamp(min=0)[None, :]).reshape(-1), torch.zeros_like(t_tie))
    owner = torch.where(in_open, own_open, own_tie)
    mypos = torch.where(in_open, lo_loc + idx_open, c_lt_loc[j] + (t_tie - before))
    mine = owner == rank
    # 4. only the selected rows travel: every row has exactly one owner, the others add zeros (no host round trip)
    k = local_rows.shape[1]
    if n_loc > 0:
        src = order[mypos.clamp(min=0, max=n_loc - 1)]
        mine_f = mine[:, None]
        out = torch.cat([torch.where(mine_f, local_rows[src], torch.zeros((), dtype=local_rows.dtype, device=dev)),
                         torch.where(mine, local_values[src], torch.zeros((), dtype=local_values.dtype, device=dev))[:, None]], 1)
        out = out.to(torch.float64).contiguous()
    else:
        out = torch.zeros((n, k + 1), dtype=torch.float64, device=dev)
    allsum(out)
    if stats is not None:
        stats['wire_bytes'] = wire
        stats['collectives'] = n_coll
        stats['n_splitter'] = S
        stats['candidates_per_rank'] = g
    return out[:, :k], out[:, k]


def _resampled_rows(source, resampler, n, stats=None):
    """n resampled (x, logq) rows of the previous round, from a TraceTuple (device-resident, sharded) or host arrays."""
    from ..samplers.sample_trace import TraceTuple
    if isinstance(source, TraceTuple):
        x_loc, logq_loc, n_total = source.refit_shard()
        ranks = resampler.ranks(n_total, n)
        ws = parallel.world()[1]
        per_chain = n_total // max(source.n_chain, 1)
        n_loc_max = max(parallel.shard_range(source.n_chain, r, ws)[1] - parallel.shard_range(source.n_chain, r, ws)[0]
                        for r in range(ws)) * per_chain
        rows, vals = select_rows_sharded(logq_loc, x_loc, ranks, stats=stats, n_loc_max=n_loc_max)
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
