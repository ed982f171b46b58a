"""``sample``: the reference's sampling entry point (bayesfast/core/sample.py:26-220) for surrogate densities,
running all chains of this rank in fused device launches instead of one worker process per chain."""
import numpy as np

from ..samplers.sample_trace import NTrace, HTrace, TraceTuple
from .density import SurrogateDensity
from .. import parallel

__all__ = ['sample']


def _pinned_like(shape, dtype):
    """Pinned host buffer or None (page-locking gigabytes takes a few 100 ms: sample() does it while the kernels run)."""
    import torch
    try:
        return torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
    except RuntimeError:
        return None


def _to_host(t, h=None):
    """Device tensor -> NumPy array through a pinned staging buffer (pageable copies run at a third of the PCIe rate)."""
    import torch
    if not t.is_cuda:
        return t.numpy()
    t = t.contiguous()
    try:
        if h is None or tuple(h.shape) != tuple(t.shape) or h.dtype != t.dtype:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return h.numpy()
    except RuntimeError:  # no pinned memory left: plain copy
        return t.cpu().numpy()


def sample(density, sample_trace=None, sampler='NUTS', n_run=None, parallel_backend=None, verbose=True,
           iters_per_launch=None):
    """Sample a surrogate density.

    density : SurrogateDensity
    sample_trace : NTrace / HTrace / dict of their keyword arguments / TraceTuple (continue the chains)
    sampler : 'NUTS' or 'HMC' (ignored when ``sample_trace`` is a trace object)
    n_run : number of iterations to run now (default: up to ``n_iter``)
    parallel_backend : accepted for signature compatibility and ignored; chains shard over the ranks of the
        default ``torch.distributed`` process group instead (one process per GPU)
    Returns a ``TraceTuple`` holding ALL chains on every rank (one all-gather over RCCL when world_size > 1).
    """
    import torch
    from ..chains import DeviceChains
    if not isinstance(density, SurrogateDensity):
        raise ValueError('density should be a SurrogateDensity.')
    prev = None
    if isinstance(sample_trace, TraceTuple):
        prev = sample_trace
        trace = prev._trace
    elif isinstance(sample_trace, (NTrace, HTrace)):
        trace = sample_trace
    elif sample_trace is None or isinstance(sample_trace, dict):
        kw = {} if sample_trace is None else sample_trace
        if sampler == 'NUTS':
            trace = NTrace(**kw)
        elif sampler == 'HMC':
            trace = HTrace(**kw)
        elif sampler in ('TNUTS', 'THMC', 'Ensemble'):
            raise NotImplementedError
        else:
            raise ValueError('unexpected value for sampler.')
    else:
        raise ValueError('unexpected value for sample_trace.')

    d = density.input_size
    rank, ws = parallel.world()
    if ws > 1:  # one process per GPU: bind this rank to its own device before anything touches device memory
        dev = parallel.local_device(torch.cuda.device_count())
        if dev is not None and torch.cuda.current_device() != dev:
            torch.cuda.set_device(dev)
    b, e = parallel.shard_range(trace.n_chain, rank, ws)
    if prev is None:
        if trace.x_0 is None:  # core/sample.py:106-113 (N(0, I) starts; the reference draws them from a Sobol sequence)
            trace.x_0 = np.random.default_rng(trace.seed() ^ 0x5bd1e995).normal(size=(trace.n_chain, d))
            trace._x_0_transformed = True
        elif not trace.x_0_transformed:  # :114-116
            trace.x_0 = density.from_original(trace.x_0)
            trace._x_0_transformed = True
        x0 = np.asarray(trace.x_0, dtype=np.float64).reshape((-1, d))
        if x0.shape[0] != trace.n_chain:  # samplers/sample_trace.py:195-199
            x0 = x0[np.random.default_rng(trace.seed()).integers(0, x0.shape[0], trace.n_chain)]
        dd = density.device()
        lp0, g0 = dd.logp_and_grad(x0[b:e])
        if not (bool(torch.isfinite(lp0).all()) and bool(torch.isfinite(g0).all())):  # base_hmc.py:42-46
            raise ValueError('failed to get finite logp and/or grad at x_0.')
        chains = DeviceChains(dd, x0[b:e], seed=trace.seed(), first_stream=b, step_size=trace._step_size,
                              metric=trace._metric, initial_mean=trace._initial_mean,
                              initial_weight=trace._initial_weight, adapt_window=trace._adapt_window)
        done = 0
        old_s = old_st = None
    else:
        chains = prev._chains
        if chains is None:
            raise ValueError('this TraceTuple cannot be continued on this rank.')
        done = prev.i_iter
        old_s, old_st = prev._samples[b:e], prev._stats[b:e]
    if n_run is None:
        n_run = trace.n_iter - done
    n_run = int(n_run)
    if n_run <= 0:
        raise ValueError('invalid value for n_run.')
    if done + n_run > trace.n_iter:
        trace.n_iter = done + n_run
    step = n_run if not iters_per_launch else int(iters_per_launch)
    ss, sts = [], []
    left = n_run
    while left > 0:
        k = min(step, left)
        s, st = chains.run(k, trace._sampler, check=False, **trace.run_kwargs())  # queued; errors are raised below
        ss.append(s)
        sts.append(st)
        left -= k
    # the host staging buffers are page-locked while the launches run
    n_loc = ss[0].shape[0]
    h_s = _pinned_like((n_loc if ws == 1 else trace.n_chain, n_run, d), torch.float64) if ss[0].is_cuda else None
    h_st = _pinned_like((n_loc if ws == 1 else trace.n_chain, n_run, sts[0].shape[2]), torch.float64) if ss[0].is_cuda else None
    chains.raise_on_error()
    s = ss[0] if len(ss) == 1 else torch.cat(ss, 1)
    st = sts[0] if len(sts) == 1 else torch.cat(sts, 1)
    # boundary conversions on device (core/sample.py:175-177), then ONE pass of device-to-host copies
    s_orig = density.to_original_device(s)
    lp_orig = density.to_original_density_device(st[:, :, 0], s)
    if ws > 1:
        s_orig = s_orig if s_orig is s else parallel.all_gather_chains(s_orig, trace.n_chain)
        lp_orig = parallel.all_gather_chains(lp_orig.contiguous(), trace.n_chain)
        s = parallel.all_gather_chains(s, trace.n_chain)
        st = parallel.all_gather_chains(st, trace.n_chain)
        if density._input_scales is None:
            s_orig = s
    same = s_orig is s
    s, st, lp_orig = _to_host(s, h_s), _to_host(st, h_st), _to_host(lp_orig)
    s_orig = s if same else _to_host(s_orig)
    if prev is not None:
        shared = same and prev._samples_original is prev._samples
        if not shared:
            s_orig = np.concatenate([prev._samples_original, s_orig], 1)
        s = np.concatenate([prev._samples, s], 1)
        if shared:
            s_orig = s
        st = np.concatenate([prev._stats, st], 1)
        lp_orig = np.concatenate([prev._logp_original, lp_orig], 1)
    if verbose and rank == 0:
        nl = st[:, :, 3].sum() if trace._sampler == 'NUTS' else st[:, :, 2].sum()
        print(' sampling finished [ {} / {} ], {} chains, {} leapfrog steps.'.format(s.shape[1], trace.n_iter,
                                                                                     trace.n_chain, int(nl)))
    return TraceTuple(trace, s, st, s_orig, lp_orig, chains)
