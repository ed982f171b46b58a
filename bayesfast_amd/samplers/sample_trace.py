"""Sampler configuration and result containers with the reference's names and semantics
(bayesfast/samplers/sample_trace.py).  In the reference one ``NTrace``/``HTrace`` object per chain holds the
configuration AND that chain's adaptive state and samples; here the configuration object is shared and the
state of all chains lives in device tensors (``bayesfast_amd.chains.DeviceChains``), while ``TraceTuple``
exposes the per-chain views the callers (``Recipe``, users) read."""
import warnings
from collections import OrderedDict

import numpy as np

from .. import _lib

__all__ = ['NTrace', 'HTrace', 'TNTrace', 'GaussianBase', 'TraceTuple', 'ChainView', '_get_step_size', '_get_metric']


class _HTrace:
    """Options shared by HTrace and NTrace (samplers/sample_trace.py:157-172, defaults identical)."""

    _sampler = None

    def __init__(self, n_chain=4, n_iter=1500, n_warmup=500, x_0=None, random_generator=None, step_size=1.,
                 adapt_step_size=True, metric='diag', adapt_metric=True, max_change=1000., target_accept=0.8,
                 gamma=0.05, k=0.75, t_0=10., initial_mean=None, initial_weight=10., adapt_window=60,
                 update_window=1, doubling=True):
        def _pos_int(v, name):
            try:
                v = int(v)
                assert v > 0
            except Exception:
                raise ValueError('{} should be a positive int, instead of {}.'.format(name, v))
            return v
        self.n_chain = _pos_int(n_chain, 'n_chain')
        self.n_iter = _pos_int(n_iter, 'n_iter')
        self.n_warmup = _pos_int(n_warmup, 'n_warmup')
        if self.n_warmup >= self.n_iter:
            raise ValueError('n_iter is {}, so n_warmup should be smaller than this number.'.format(self.n_iter))
        self.x_0 = None if x_0 is None else np.atleast_1d(x_0).copy()
        self._x_0_transformed = False
        self.random_generator = random_generator
        if step_size is not None:  # None: 1 at use (_set_step_size_2, samplers/sample_trace.py:365-373); Recipe fills it in
            try:                   # from the previous round when it is None (core/recipe.py:1033-1038)
                step_size = float(step_size)
                assert step_size > 0
            except Exception:
                raise ValueError('invalid value for step_size.')
        self._step_size = step_size
        self._adapt_step_size = bool(adapt_step_size)
        if isinstance(metric, str):
            if metric not in ('diag', 'full'):
                raise ValueError('invalid value for metric.')
            self._metric = metric  # kept as given: Recipe tests `_metric == 'diag'` / `'full'` (core/recipe.py:971-974,1040-1045)
        else:
            metric = np.asarray(metric, dtype=np.float64)
            n = metric.shape[0] if metric.ndim else 0
            if not (metric.shape == (n,) or metric.shape == (n, n)):  # samplers/sample_trace.py:382-388
                raise ValueError('invalid value for metric.')
            if metric.ndim == 1 and not np.all(metric > 0):
                raise ValueError('invalid value for metric.')
            self._metric = metric
        self._adapt_metric = bool(adapt_metric)
        try:
            self.max_change = float(max_change)
            assert self.max_change > 0
        except Exception:
            raise ValueError('max_change should be a positive float, instead of {}.'.format(max_change))
        try:
            self._target_accept = float(target_accept)
            assert 0 < self._target_accept < 1
        except Exception:
            raise ValueError('invalid value for target_accept.')
        self._gamma, self._k, self._t_0 = float(gamma), float(k), float(t_0)
        if self._gamma == 0 or self._t_0 < 0:
            raise ValueError('invalid value for gamma or t_0.')
        self._initial_mean = None if initial_mean is None else np.atleast_1d(initial_mean).astype(np.float64)
        self._initial_weight = float(initial_weight)
        self._adapt_window = _pos_int(adapt_window, 'adapt_window')
        self._update_window = _pos_int(update_window, 'update_window')
        self._doubling = bool(doubling)

    x_0_transformed = property(lambda self: self._x_0_transformed)

    @property
    def input_size(self):
        try:
            return self.x_0.shape[-1]
        except Exception:
            return None

    def seed(self):
        """Integer seed of the per-chain xoshiro streams, from ``random_generator`` (None: OS entropy).  Resolved ONCE per
        trace and, under ``torch.distributed``, taken from rank 0, so that every rank derives the same starting points
        and stream keys from it (results do not depend on the number of ranks)."""
        if getattr(self, '_seed', None) is None:
            g = self.random_generator
            if g is None:
                v = int(np.random.SeedSequence().generate_state(1, np.uint64)[0] >> 1)
            elif isinstance(g, (int, np.integer)):
                v = int(g)
            else:
                v = int(np.random.default_rng(g).integers(0, 2**63 - 1))
            from .. import parallel
            self._seed = parallel.broadcast_int(v)
        return self._seed

    def run_kwargs(self):
        kw = dict(n_warmup=self.n_warmup, max_change=self.max_change, target_accept=self._target_accept,
                  gamma=self._gamma, k=self._k, t_0=self._t_0, adapt_step_size=self._adapt_step_size,
                  adapt_metric=self._adapt_metric, update_window=self._update_window, doubling=self._doubling)
        return kw


class NTrace(_HTrace):
    """Trace options of the NUTS sampler (samplers/sample_trace.py:499-537)."""
    _sampler = 'NUTS'

    def __init__(self, n_chain=4, n_iter=1500, n_warmup=500, x_0=None, random_generator=None, step_size=1.,
                 adapt_step_size=True, metric='diag', adapt_metric=True, max_change=1000., max_treedepth=10,
                 target_accept=0.8, gamma=0.05, k=0.75, t_0=10., initial_mean=None, initial_weight=10.,
                 adapt_window=60, update_window=1, doubling=True):
        super().__init__(n_chain, n_iter, n_warmup, x_0, random_generator, step_size, adapt_step_size, metric,
                         adapt_metric, max_change, target_accept, gamma, k, t_0, initial_mean, initial_weight,
                         adapt_window, update_window, doubling)
        try:
            self.max_treedepth = int(max_treedepth)
            assert 0 < self.max_treedepth <= _lib.MAX_TREEDEPTH
        except Exception:
            raise ValueError('max_treedepth should be a positive int not larger than {}, instead of '
                             '{}.'.format(_lib.MAX_TREEDEPTH, max_treedepth))

    def run_kwargs(self):
        return dict(super().run_kwargs(), max_treedepth=self.max_treedepth)


class GaussianBase:
    """Base density of the tempered samplers (``TNTrace(density_base=...)``): the Gaussian N(mean, cov) in the sampler's
    space.  (The reference accepts any Density there, samplers/sample_trace.py:540-557; the device path covers quadratic
    log-densities, of which this is the one in use: a Gaussian approximation of the posterior.)"""

    def __init__(self, mean, cov):
        self.mean = np.atleast_1d(np.asarray(mean, dtype=np.float64))
        self.cov = np.atleast_2d(np.asarray(cov, dtype=np.float64))
        if self.cov.shape != (self.mean.size, self.mean.size):
            raise ValueError('invalid value for density_base.')

    def logp(self, x):
        r = np.asarray(x, dtype=np.float64) - self.mean
        d = self.mean.size
        return -0.5 * np.einsum('...i,ij,...j->...', r, np.linalg.inv(self.cov), r) - 0.5 * (
            d * np.log(2 * np.pi) + np.linalg.slogdet(self.cov)[1])


class TNTrace(NTrace):
    """Trace options of the tempered NUTS sampler (samplers/sample_trace.py:540-567,607-629): ``NTrace`` plus the base
    density and ``logxi``."""
    _sampler = 'TNUTS'

    def __init__(self, density_base, logxi=0., **kwargs):
        if not isinstance(density_base, GaussianBase):
            raise ValueError('invalid value for density_base.')
        self.density_base = density_base
        try:
            self.logxi = float(logxi)
        except Exception:
            raise ValueError('invalid value for logxi.')
        super().__init__(**kwargs)


class HTrace(_HTrace):
    """Trace options of the static HMC sampler (samplers/sample_trace.py:458-497)."""
    _sampler = 'HMC'

    def __init__(self, n_chain=4, n_iter=1500, n_warmup=500, n_int_step=32, x_0=None, random_generator=None,
                 step_size=1., adapt_step_size=True, metric='diag', adapt_metric=True, max_change=1000.,
                 target_accept=0.8, gamma=0.05, k=0.75, t_0=10., initial_mean=None, initial_weight=10.,
                 adapt_window=60, update_window=1, doubling=True):
        super().__init__(n_chain, n_iter, n_warmup, x_0, random_generator, step_size, adapt_step_size, metric,
                         adapt_metric, max_change, target_accept, gamma, k, t_0, initial_mean, initial_weight,
                         adapt_window, update_window, doubling)
        try:
            self.n_int_step = int(n_int_step)
            assert self.n_int_step > 0
        except Exception:
            raise ValueError('n_int_step should be a positive int, instead of {}.'.format(n_int_step))

    def run_kwargs(self):
        return dict(super().run_kwargs(), n_int_step=self.n_int_step)


class _Stats:
    """``NStats``/``HStats`` look-alike of one chain (samplers/hmc_utils/stats.py:39-94)."""

    def __init__(self, items, table, n_warmup):
        self.stats_items = items
        self._n_warmup = n_warmup
        for i, k in enumerate(items):
            col = table[:, i]
            if k in ('tree_depth', 'tree_size', 'n_int_step'):
                col = col.astype(int)
            elif k in ('warmup', 'diverging', 'accepted'):
                col = col.astype(bool)
            setattr(self, '_' + k, list(col))

    n_iter = property(lambda self: len(self._logp))
    n_warmup = property(lambda self: self._n_warmup)

    def get(self, since_iter=None, include_warmup=False):
        if since_iter is None:
            since_iter = 0 if include_warmup else self._n_warmup
        return OrderedDict((k, getattr(self, '_' + k)[int(since_iter):]) for k in self.stats_items)

    __call__ = get


class ChainView:
    """What one element of the reference's ``TraceTuple`` offers to its readers."""

    def __init__(self, tt, i):
        self._tt, self.chain_id = tt, i

    n_iter = property(lambda self: self._tt.n_iter)
    n_warmup = property(lambda self: self._tt.n_warmup)
    i_iter = property(lambda self: self._tt.i_iter)
    input_size = property(lambda self: self._tt.input_size)
    samples = property(lambda self: self._tt.samples[self.chain_id])
    samples_original = property(lambda self: self._tt.samples_original[self.chain_id])
    logp = property(lambda self: self._tt.logp[self.chain_id])
    logp_original = property(lambda self: self._tt.logp_original[self.chain_id])

    @property
    def stats(self):
        t = self._tt
        return _Stats(t._stat_items, t._stats[self.chain_id], t.n_warmup)

    @property
    def n_call(self):
        t = self._tt
        if t.sampler in ('NUTS', 'TNUTS'):  # samplers/sample_trace.py:529-530
            return int(t._stats[self.chain_id, 1:, t._stat_items.index('tree_size')].sum()) + t.i_iter + 1
        return t.i_iter * (t._trace.n_int_step + 1) + 1  # :487-489

    def get(self, since_iter=None, include_warmup=False, original_space=True, return_type='samples', flatten=True):
        return self._tt.get(since_iter, include_warmup, original_space, return_type, flatten=False)[self.chain_id]

    __call__ = get


class TraceTuple:
    """All chains of one ``sample`` call (samplers/sample_trace.py:631-801).

    The arrays stay where the sampler wrote them: on this rank's GPU, for this rank's chains.  The reference's array
    views (``samples``, ``logp``, ``stats``, ``get`` ...) materialise them on the host on first use and are cached.  Under
    ``torch.distributed`` with more than one rank they need every rank's chains: call ``gather()`` -- a collective, on
    EVERY rank -- first (or ``sample(..., gather=True)``); a view asked for before that raises instead of starting a
    collective behind the caller's back (``if rank == 0: save(tt.samples)`` would otherwise hang).  The refit path does
    not need them: ``refit_shard`` hands the device-resident rows to
    ``bayesfast_amd.core.refit.select_fit_points``, and the warm-start helpers reduce over the ranks on the device."""

    _FIELDS = ('samples', 'stats', 'samples_original', 'logp_original')

    def __init__(self, trace, samples, stats, samples_original, logp_original, chains=None, stats_t=None):
        self._trace = trace
        self.sampler = trace._sampler
        self._stat_items = _lib.HSTATS if self.sampler == 'HMC' else _lib.NSTATS
        self._parts = dict(samples=samples, stats=stats, samples_original=samples_original, logp_original=logp_original)
        if stats_t is not None:  # TNUTS: (u, weight) of every sample (TNStepStats)
            self._parts['stats_t'] = stats_t
        self._host = {}
        self._chains = chains  # DeviceChains of this rank

    # ---- lazily materialised host arrays (all chains) ----
    def _array(self, name, collective=False):
        if name not in self._host:
            t = self._parts[name]
            if not isinstance(t, np.ndarray):  # this rank's shard as a tensor
                from .. import parallel
                if parallel.world()[1] > 1:
                    if not collective:
                        raise RuntimeError('this TraceTuple holds the chains of one rank only: call gather() on every rank '
                                           '(or sample(..., gather=True)) before using its host views.')
                    t = parallel.all_gather_chains(t.contiguous(), self._trace.n_chain)
                t = t.cpu().numpy()
            self._host[name] = np.asarray(t)
        return self._host[name]

    def gather(self):
        """Materialise the host arrays of ALL chains (one all-gather per field over the ranks: a collective, to be called
        by every rank; a plain device-to-host copy without a process group).  Returns self."""
        for name in self._parts:
            if not (name == 'samples_original' and self._parts[name] is self._parts['samples']):
                self._array(name, collective=True)
        return self

    def _adapted_state(self):
        """Host arrays, all chains, of what the reference keeps per chain next to its samples: the starting points and the
        adapted step-size (DualAverageAdaptation, step_size.py:10-22) and metric state.  Under ``torch.distributed`` a
        collective (call it on every rank); cached."""
        if 'adapted' not in self._host:
            from .. import parallel
            ch = self._chains
            if ch is None:
                raise ValueError('this TraceTuple does not carry its chains.')
            out = {}
            for k in ('log_step', 'log_bar', 'hbar', 'count'):
                out[k] = parallel.all_gather_chains(ch.field(k).contiguous(), self.n_chain).cpu().numpy()
            if ch.full_metric:
                out['cov'] = parallel.all_gather_chains(ch.covariance(), self.n_chain).cpu().numpy()
            else:
                out['var'] = parallel.all_gather_chains(ch.field('var').contiguous(), self.n_chain).cpu().numpy()
            x0 = np.asarray(self._trace.x_0, dtype=np.float64)
            out['x_0'] = x0.reshape(-1, x0.shape[-1])
            if out['x_0'].shape[0] != self.n_chain:   # (x_0 was a pool the chains drew their starts from)
                out['x_0'] = self._samples[:, 0]
            self._host['adapted'] = out
        return self._host['adapted']

    _samples = property(lambda self: self._array('samples'))
    _stats = property(lambda self: self._array('stats'))
    _logp_original = property(lambda self: self._array('logp_original'))

    @property
    def _samples_original(self):
        if self._parts['samples_original'] is self._parts['samples']:
            return self._samples
        return self._array('samples_original')

    def device(self, name):
        """This rank's shard of one of ``samples``, ``stats``, ``samples_original``, ``logp_original`` as stored
        (a device tensor after ``sample``; rows = this rank's chains)."""
        return self._parts[name]

    def _local_tensor(self, name):
        import torch
        t = self._parts[name]
        return t if hasattr(t, 'is_cuda') else torch.as_tensor(np.asarray(t))

    def n_refit_rows(self, since_iter=None):
        since = self.n_warmup if since_iter is None else int(since_iter)
        return self.n_chain * max(self.i_iter - since, 0)

    def refit_shard(self, since_iter=None):
        """(x (n_loc, d) original-space samples, logq (n_loc,), n_total): this rank's rows of what
        ``get(flatten=True)`` / ``get(return_type='logp', flatten=True)`` return, in the same (chain-major) order."""
        since = self.n_warmup if since_iter is None else int(since_iter)
        x = self._local_tensor('samples_original')[:, since:]
        lq = self._local_tensor('logp_original')[:, since:]
        return x.reshape(-1, x.shape[-1]), lq.reshape(-1), self.n_refit_rows(since)

    sample_traces = property(lambda self: tuple(ChainView(self, i) for i in range(self.n_chain)))
    n_chain = property(lambda self: self._trace.n_chain)
    n_iter = property(lambda self: self._trace.n_iter)
    i_iter = property(lambda self: int(self._parts['samples'].shape[1]))
    n_warmup = property(lambda self: self._trace.n_warmup)
    input_size = property(lambda self: int(self._parts['samples'].shape[-1]))
    finished = property(lambda self: self.i_iter >= self.n_iter)
    samples = property(lambda self: self._samples)
    samples_original = property(lambda self: self._samples_original)
    logp = property(lambda self: self._stats[:, :, 0])
    logp_original = property(lambda self: self._logp_original)
    stats = property(lambda self: [t.stats for t in self.sample_traces])

    @property
    def n_call(self):
        st = self._stats
        if self.sampler in ('NUTS', 'TNUTS'):  # samplers/sample_trace.py:529-530, summed over chains
            return int(st[:, 1:, self._stat_items.index('tree_size')].sum()) + self.n_chain * (self.i_iter + 1)
        return self.n_chain * (self.i_iter * (self._trace.n_int_step + 1) + 1)  # :487-489

    def stat(self, name):
        """(n_chain, i_iter) array of one statistic by its reference name ('u' and 'weight' for TNUTS included)."""
        if name in ('u', 'weight') and 'stats_t' in self._parts:
            return self._array('stats_t')[:, :, ('u', 'weight').index(name)]
        return self._stats[:, :, self._stat_items.index(name)]

    def get(self, since_iter=None, include_warmup=False, original_space=True, return_type='samples', flatten=True):
        if return_type == 'all':
            return [self.get(since_iter, include_warmup, original_space, r, flatten) for r in ('samples', 'logp')]
        if since_iter is None:
            since_iter = 0 if include_warmup else self.n_warmup
        since_iter = int(since_iter)
        if since_iter >= self.i_iter - 1:
            raise ValueError('since_iter is too large. Nothing to return.')
        if return_type == 'samples':
            s = (self._samples_original if original_space else self._samples)[:, since_iter:]
            return s.reshape((-1, self.input_size)) if flatten else s
        if return_type == 'logp':
            l = (self._logp_original if original_space else self.logp)[:, since_iter:]
            return l.flatten() if flatten else l
        raise ValueError('invalid value for return_type.')

    __call__ = get

    def __getitem__(self, key):
        return self.sample_traces[key]

    def __len__(self):
        return self.n_chain

    def __iter__(self):
        return iter(self.sample_traces)


def _get_step_size(sample_trace):
    """Warm-start step size for the next round (samplers/sample_trace.py:804-817): exp(log_bar) * d^(1/4), averaged
    over ALL chains (summed on this rank's device, all-reduced over the ranks)."""
    if not isinstance(sample_trace, TraceTuple) or sample_trace._chains is None:
        raise ValueError('invalid value for sample_trace.')
    from .. import parallel
    lb = sample_trace._chains.field('log_bar')
    tot = parallel.all_reduce_sum((lb.exp() * sample_trace.input_size**0.25).sum().reshape(1))
    return float(tot.item()) / sample_trace.n_chain


def _get_metric(sample_trace, target, from_samples=True):
    """Warm-start metric (samplers/sample_trace.py:820-847): the covariance of the post-warm-up samples in the sampler's
    space (two passes on the device: mean, then the centred Gram matrix, each all-reduced over the ranks), or the mean
    of the chains' adapted covariances."""
    if not isinstance(sample_trace, TraceTuple):
        raise ValueError('invalid value for sample_trace.')
    if target not in ('diag', 'full'):
        raise ValueError('unexpected value for target.')
    from .. import parallel
    import torch
    if from_samples:
        x = sample_trace._local_tensor('samples')[:, sample_trace.n_warmup:]
        x = x.reshape(-1, x.shape[-1]).to(torch.float64)
        n = sample_trace.n_refit_rows()
        mean = parallel.all_reduce_sum(x.sum(0)) / n
        xc = x - mean
        cov = parallel.all_reduce_sum(xc.T @ xc) / (n - 1)  # np.cov(..., rowvar=False)
    else:
        if sample_trace._chains is None:
            raise ValueError('invalid value for sample_trace.')
        cov = parallel.all_reduce_sum(sample_trace._chains.covariance().sum(0)) / sample_trace.n_chain
    cov = cov.cpu().numpy()
    return np.diag(cov) if target == 'diag' else cov
