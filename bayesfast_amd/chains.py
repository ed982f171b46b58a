"""Batched chain state on one GPU and the fused NUTS/HMC launch (bfhip_sampler_run)."""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceDensity, _ptr

__all__ = ['DeviceChains']

# what 'auto' runs while the trees of a group are in step: 'split' (eight waves per 16 chains, two per SIMD with disjoint work:
# integrator and bookkeeper waves; NUTS on the plain surrogate at 33 <= d <= 64 -- the library runs everything else as 'group')
# or 'group'.  Same results either way, bit for bit; 'split' is 3 % (7-leaf trees) to 13 % (15-leaf trees) faster
IN_STEP_LAYOUT = __import__('os').environ.get('BFHIP_IN_STEP_LAYOUT', 'split')
if IN_STEP_LAYOUT not in ('group', 'split'):
    raise ValueError("BFHIP_IN_STEP_LAYOUT should be 'group' or 'split', not {!r}.".format(IN_STEP_LAYOUT))


def _torch():
    import torch
    return torch


class DeviceChains:
    """All chains of one GPU shard: positions, per-chain step-size and metric adaptation state, RNG streams.

    The per-chain arithmetic is the reference's (``BaseHMC.astep``, samplers/hmc_utils/base_hmc.py:62-85);
    the state that the reference keeps in one ``NTrace``/``HTrace`` object per chain
    (samplers/sample_trace.py:157-455) lives here in three device tensors, so a later ``run`` continues the
    chains exactly where they stopped.

    Parameters
    ----------
    density : DeviceDensity
    x_0 : (n_chain, d) array_like, starting points in the sampler's (transformed) space
    seed, first_stream : the xoshiro256++ stream of chain i is (seed, first_stream + i); with
        first_stream = global index of this shard's first chain, results do not depend on the sharding.
    """

    def __init__(self, density, x_0, seed=0, first_stream=0, step_size=1., metric=None, initial_mean=None,
                 initial_weight=10., adapt_window=60):
        torch = _torch()
        if not isinstance(density, DeviceDensity):
            raise ValueError('density should be a DeviceDensity.')
        self.density = density
        self.ctx = density.ctx
        x_0 = self.ctx.tensor(x_0, torch.float64)
        if x_0.dim() != 2 or x_0.shape[1] != density.d:
            raise ValueError('x_0 should have shape (n_chain, {}).'.format(density.d))
        self.n_chain, self.d = x_0.shape
        self._n_cu = None
        lib, h = self.ctx._lib, self.ctx.handle
        self.rng = torch.empty((self.n_chain, 4), dtype=torch.int64, device=self.ctx.device)
        self.sc = self.ctx.empty((self.n_chain, _lib.SC_N))
        self.vec = self.ctx.empty((self.n_chain, _lib.VEC_N, self.d))
        self.n_leapfrog = torch.zeros((1,), dtype=torch.int64, device=self.ctx.device)
        # metric: None / 1-d variances -> QuadMetricDiag(Adapt); 'full' / 2-d covariance -> QuadMetricFull(Adapt)
        cov0 = None
        self.full_metric = isinstance(metric, str) and metric == 'full'
        if isinstance(metric, str):
            if metric not in ('diag', 'full'):
                raise ValueError('invalid value for metric.')
            metric = None
        elif metric is not None and np.ndim(metric) == 2:
            cov0 = np.asarray(metric, dtype=np.float64)
            if cov0.shape != (self.d, self.d):
                raise ValueError('invalid value for metric.')
            metric, self.full_metric = None, True
        mv = None if metric is None else self.ctx.tensor(np.asarray(metric, dtype=np.float64).reshape(self.d))
        im = None if initial_mean is None else self.ctx.tensor(np.asarray(initial_mean, dtype=np.float64).reshape(self.d))
        _lib.check(lib.bfhip_rng_seed(h, self.n_chain, int(seed) & (2**64 - 1), int(first_stream), _ptr(self.rng)))
        _lib.check(lib.bfhip_chain_init(h, self.n_chain, self.d, _ptr(x_0), float(step_size), _ptr(mv), _ptr(im),
                                        float(initial_weight), int(adapt_window), _ptr(self.sc), _ptr(self.vec)))
        self.mat = None
        if self.full_metric:
            self.mat = torch.zeros((self.n_chain, _lib.MAT_N, self.d, self.d), dtype=torch.float64, device=self.ctx.device)
            c0 = None if cov0 is None else self.ctx.tensor(cov0, torch.float64)
            _lib.check(lib.bfhip_metric_init_full(h, self.n_chain, self.d, _ptr(c0), float(initial_weight), _ptr(self.sc),
                                                  _ptr(self.mat)))
            self.raise_on_error()
        self.i_iter = 0

    def run(self, n_run, sampler='NUTS', n_warmup=500, max_treedepth=10, n_int_step=32, max_change=1000.,
            target_accept=0.8, gamma=0.05, k=0.75, t_0=10., adapt_step_size=True, adapt_metric=True,
            update_window=1, doubling=True, samples=None, stats=None, check=True, launch_iters='auto', layout='auto'):
        """Advance every chain by ``n_run`` iterations, in kernel launches of at most ``launch_iters`` iterations
        (None: one launch; a sequence: these lengths, the last one repeating; 'auto': launches of 100 while the chains adapt,
        then of 250 -- measured on the default 1500-iteration run of 4096 chains: 64.5 ms in launches of 250, 61.6 ms with the
        warm-up in launches of 100, 72.8 ms with the warm-up in one launch) queued back to back on the context's stream.

        The chains of a workgroup share the gradient tiles of every trip, so they run fastest in step; chains whose
        trees differ drift apart inside a launch and every launch boundary lines them up again (measured on the
        default 1500-iteration run: 83 ms in one launch, 75 ms in launches of 250).  The cut does not change any
        chain's results.

        ``layout`` chooses how a workgroup's 16 chains are laid out (``bfhip_sampler_config.chain_layout``): 'group' (lane
        per chain: fastest while the chains of a workgroup stay in step), 'wave' (wave per chain: insensitive to chains
        out of step) or 'auto', decided per launch: 'group' when at least 98 % of the NUTS trees of an earlier launch's
        last 32 iterations had one and the same size (static HMC: always), 'wave' otherwise and for the first launches.
        "Earlier" is the launch just before inside a run, and the one before that for the first launch of a run: a pure
        function of the sequence of launches, never of host timing.  With ``hist_reduce`` set (``sample()`` does it when the
        chains are sharded over ranks) the trees of all ranks decide, so the choice does not depend on the sharding.
        Both layouts follow the same per-chain arithmetic and random streams; their floating-point sums are ordered
        differently, so results are bit-reproducible (and independent of sharding and launch cuts) for a fixed layout,
        and agree to rounding between layouts.

        Returns (samples (n_chain, n_run, d), stats (n_chain, n_run, 11)) device tensors; the stats columns
        follow ``_lib.NSTATS`` / ``_lib.HSTATS`` (samplers/hmc_utils/stats.py:7-14)."""
        torch = _torch()
        self.density.upload_if_needed()
        cfg = _lib.SamplerConfig()
        cfg.sampler = {'NUTS': 0, 'HMC': 1}[sampler]
        cfg.n_warmup = int(n_warmup)
        cfg.max_treedepth = int(max_treedepth)
        cfg.n_int_step = int(n_int_step)
        cfg.max_change = float(max_change)
        cfg.target_accept, cfg.gamma, cfg.k, cfg.t_0 = float(target_accept), float(gamma), float(k), float(t_0)
        cfg.adapt_step_size, cfg.adapt_metric = int(bool(adapt_step_size)), int(bool(adapt_metric))
        cfg.update_window, cfg.doubling = int(update_window), int(bool(doubling))
        cfg.full_metric = int(self.full_metric)
        cfg.metric_mat = self.mat.data_ptr() if self.full_metric else None
        if layout not in ('auto', 'group', 'wave', 'split'):
            raise ValueError("layout should be 'auto', 'group', 'split' or 'wave'.")
        if layout == 'auto':   # (tuning: one layout for every launch the dispatch would have chosen; an explicit layout wins)
            layout = __import__('os').environ.get('BFHIP_FORCE_LAYOUT') or layout
            if layout not in ('auto', 'group', 'wave', 'split'):
                raise ValueError("BFHIP_FORCE_LAYOUT should be 'group', 'split' or 'wave'.")
        judged = layout == 'auto'
        n_run = int(n_run)
        if samples is None:
            samples = self.ctx.empty((self.n_chain, n_run, self.d))
        if stats is None:
            stats = self.ctx.empty((self.n_chain, n_run, _lib.STAT_STRIDE))
        # caller-supplied output buffers: the kernel writes n_chain * n_run rows with these strides, so anything else
        # would be a silent out-of-bounds or garbled write
        for name, t, shape in (('samples', samples, (self.n_chain, n_run, self.d)),
                               ('stats', stats, (self.n_chain, n_run, _lib.STAT_STRIDE))):
            if (tuple(t.shape) != shape or t.dtype != torch.float64 or t.device != self.ctx.device or
                    not t.is_contiguous()):
                raise ValueError('{} should be a contiguous float64 tensor of shape {} on {}.'.format(name, shape, self.ctx.device))
        # launch lengths: one number, or a sequence whose last entry repeats (sample(): the warm-up in one launch)
        if isinstance(launch_iters, str):
            if launch_iters != 'auto':
                raise ValueError("launch_iters should be a number, a sequence of numbers, None or 'auto'.")
            n_adapting = max(0, min(int(n_warmup) - self.i_iter, n_run))   # a function of the arguments only
            launch_iters = [100] * (-(-n_adapting // 100)) + [250]
        lens = [max(1, int(v)) for v in launch_iters] if isinstance(launch_iters, (list, tuple)) else [max(1, int(launch_iters) if launch_iters else n_run)]
        ends, steps = [], []
        while (ends[-1] if ends else 0) < n_run:
            steps.append(lens[min(len(ends), len(lens) - 1)])
            ends.append((ends[-1] if ends else 0) + steps[-1])
        for i_launch, (done, step) in enumerate(zip(ends, steps)):  # iter_end of each launch; output rows are relative to i_iter
            # the layout is chosen per launch from the trees of an EARLIER launch, as a pure function of the sequence of
            # launches (never of host timing): inside a run, the launch just before (the host waits for its answer: it has
            # nothing else to queue, and the gap is a launch latency); the first launch of a run, the launch before the
            # last one, so that runs issued back to back keep one launch queued behind the running one and still follow
            # the chains' behaviour, one launch late
            lay = layout
            if lay == 'auto':
                ans = self._trees_in_step(lag=1 if i_launch > 0 else 2)   # (asked for at every launch: the answers are consumed in order)
                tree, laggard = ans & 4095, ans >= 4096
                in_step = sampler == 'HMC' or tree > 0
                lay = IN_STEP_LAYOUT if in_step else 'wave'
                deep = self._deep_trees_prefer_waves()
                if sampler == 'NUTS' and deep and tree >= deep:
                    lay = 'wave'
                if sampler == 'NUTS' and self._small_problem():
                    lay = 'wave'
                elif sampler == 'NUTS' and self._lanes_whatever_the_trees():
                    lay = IN_STEP_LAYOUT
                # Some chain builds trees many times the common size (outside the bound, say, where the surrogate is its linear
                # extrapolation; reported from four times the mean over the window, in step or not): a launch lasts as long as its
                # busiest chain, and only the wave layout's launches have a second part for such chains (bfhip_sampler.hip:
                # launch_nuts_pipe).  64-d x 4096 chains, ONE chain of them outside the bound (16 x the others' leapfrogs): split
                # 2.6 x 10^8 (from 11.8), group 1.7, wave 4.9 (tools/leak_probe.py).
                if sampler == 'NUTS' and laggard:
                    lay = 'wave'
                if lay == 'split' and in_step and self._two_groups_fit_a_cu():
                    lay = 'group'
            cfg.chain_layout = {'group': 1, 'wave': 2, 'split': 3}[lay]
            self.last_layout = lay
            _lib.check(self.ctx._lib.bfhip_sampler_run(
                self.ctx.handle, C.byref(cfg), self.n_chain, self.i_iter + min(done, n_run), _ptr(self.rng), _ptr(self.sc),
                _ptr(self.vec), self.i_iter, n_run, _ptr(samples), _ptr(stats), _ptr(self.n_leapfrog)))
            if not judged:
                self._answers = getattr(self, '_answers', []) + [None]   # (a launch that was not judged: no stale answer later)
                del self._answers[:-4]
            if judged:
                # a launch that ends the warm-up is judged more leniently: its last iterations still adapt the step size
                # (a few trees of another size), the launch after it runs with the frozen, averaged one
                i0, i1 = self.i_iter + done - step, self.i_iter + min(done, n_run)
                # (launches inside the warm-up: 7-leaf trees with one 15-leaf tree in ten already run faster in step -- the late
                # warm-up launches of the default run 5.5 against 6.0 ms, tools/launch_times.py)
                self._note_trees(stats, done - step, min(done, n_run), sampler,
                                 share=0.85 if i0 < n_warmup <= i1 else (0.8 if i1 < n_warmup else 0.98))
        self.i_iter += n_run
        if check:
            self.raise_on_error()
        return samples, stats

    def _shape_facts(self):
        """(plain, featured, n): the common surrogate (linear + quadratic configs with the bound) with nothing else / with the decay
        term OR the constraint transform (the feature sets the pipelined wave-per-chain kernel has instantiations for) / chains per
        rank (sharded: the ranks' average, equal on all of them).  A function of the shapes only."""
        if self._n_cu is None:
            self._n_cu = int(_torch().cuda.get_device_properties(self.ctx.device).multi_processor_count)
        sp = self.density.spec
        # (input scaling of a linear + quadratic surrogate is folded into its coefficients at upload: device.density_desc_from_spec)
        from .device import folds_input_scales
        common = ((sp.get('su_lo') is None or folds_input_scales(sp)) and sp.get('link') is None and sp.get('chi2') is None and
                  bool(sp['poly'].get('use_bound')) and
                  sorted(c['order'] for c in sp['poly']['configs']) == ['linear', 'quadratic'] and not self.full_metric)
        dec, tr = bool(sp.get('use_decay')), sp.get('ranges') is not None
        n = self.n_chain if self.n_chain_rule is None else self.n_chain_rule
        return common and not dec and not tr, common and (dec != tr), n

    def _small_problem(self):
        """NUTS where the wave-per-chain kernel beats the lane-per-chain layouts although the trees are in step: the latter have
        d / 16 (group) or 2 d / 16 (split) waves per workgroup of 16 chains, so few chains leave most of a CU idle, while the
        wave-per-chain kernel spreads fewer chains per workgroup over more CUs (bfhip_sampler.hip: wave_layout_cpg).  Measured,
        in-step 7-leaf trees (tools/layout_ab.py, profiles/r03s_layout_ab.log), wave against the best lane-per-chain layout: d = 32
        (split, two + two waves): 1024 chains 3.2 against 3.0 x 10^8, 2048 5.5 against 6.0, 4096 8.2 against 12.1; d = 16 (split,
        one + one wave): 1024 2.75 against 2.80, 4096 7.5 against 11.0; d = 64: the split layout ahead from 2048 chains.  With the
        decay term or behind the constraint transform the lane-per-chain layout is the group kernel (no split instantiation) and the
        pipelined kernel stays ahead up to eight chains per CU (tools/dispatch_sweep.py, profiles/r04e_dispatch_sweep.log: d = 32
        x 1024 chains 3.3 against 2.2 x 10^8 with the decay term, 2.4 against 1.5 bounded; d = 64 x 1024 3.0 against 2.5 and 2.3
        against 1.4; at sixteen chains per CU the group kernel wins everywhere).  A function of the shapes only (never of timing)."""
        plain, featured, n = self._shape_facts()
        if featured:
            return self.d <= 64 and n <= 8 * self._n_cu
        if not plain:   # (everything else runs the sliced kernel in the wave layout and the group kernel in step)
            return False
        # (32 < d <= 64: with at most four chains per workgroup -- n <= 4 x CUs, wave_layout_cpg -- the pipelined kernel's jobs run
        # on 4 x 4 x 4 MFMA tiles: 1024 chains 3.7 against the split layout's 2.9 x 10^8, 512 chains 1.9 against 1.5)
        # (round 5: up to four chains per CU at d <= 32 the wave layout is the latency kernel, csrc/bfhip_lone.h -- d = 16 x 1024 chains
        # 4.8 against the split layout's 2.8 x 10^8, profiles/r05_lone_sweep.log -- so four chains per CU are "small" at d <= 16 too)
        return ((self.d <= 16 and n <= 4 * self._n_cu) or (16 < self.d <= 32 and n < 6 * self._n_cu) or
                (32 < self.d <= 64 and n <= 4 * self._n_cu))

    def _lanes_whatever_the_trees(self):
        """NUTS on the plain surrogate at d <= 32 with at least sixteen chains per CU: the split layout's trip is short there (one or
        two integrator waves per 16 chains), and it stays ahead of the wave layout when the trees of a group differ -- 7- and 15-leaf
        trees side by side: d = 16 x 4096 chains 12.2 against 8.9 x 10^8, d = 32 14.3 against 9.8; trees of 7 to 63 leaves: 10.6 / 10.5
        against 9.8 (tools/dispatch_sweep.py, profiles/r04e_dispatch_sweep.log).  At eight chains per CU it depends on how different
        the trees are, and the judgement of the last launch decides as everywhere else."""
        plain, _, n = self._shape_facts()
        return plain and self.d <= 32 and n >= 16 * self._n_cu

    def _deep_trees_prefer_waves(self):
        """The common surrogate WITH the decay term at 33 <= d <= 64: the group kernel's rate falls with the tree size (every trip runs
        the bound's and the decay's tiles, and a chain outside the decay ellipsoid makes its whole group's trips 60 % longer, which the
        launch then waits for), the pipelined wave-per-chain kernel's does not -- 4096 chains x 64-d, trees in step: 7 leaves 9.7 against
        6.2 x 10^8, 15: 7.7 against 6.6, 31: 6.3 against 7.0, 1022 (config 3's second round): 5.8 -- 3.7 with ONE such chain -- against
        7.4 (tools/layout_ab.py, bench.py --workload banana_decay; docs/EXPERIMENTS.md).  From 24 leaves up 'auto' takes the wave
        layout there.  Behind the constraint transform the group kernel stays ahead (31 leaves: 6.6 against 5.0), and the plain
        surrogate's split kernel too (10.6 against 9.2).  Round 6: where the decay term's matrix is the bound's the wave layout runs two
        matrices (bfhip_nuts_pipe.h, DEC = 2) and wins earlier -- 4096 chains, 15-leaf trees: d = 64 8.1 against 7.8 x 10^8, d = 32 9.0
        against 5.7; 7-leaf trees stay with the group kernel (9.8 against 7.8, 8.2 against 7.9): profiles/r06_dispatch_sweep.log.
        Returns the tree size from which 'auto' takes the wave layout (0: never).  A function of the shapes and the uploaded arrays only."""
        plain, featured, n = self._shape_facts()
        sp = self.density.spec
        if plain and 32 < self.d <= 64 and 4 * self._n_cu < n <= 8 * self._n_cu:
            # (round 6 sweep, the one cell under 0.9: plain surrogate, 64-d x 2048 chains -- eight chains per CU, where the pipelined
            # kernel's jobs run on 4 x 4 x 4 tiles -- 15-leaf trees: wave 6.1 against split 5.3 x 10^8; 7-leaf trees: 5.9 against 6.1)
            return 12
        if not (featured and bool(sp.get('use_decay'))):
            return 0
        from .workloads import decay_shares_bound
        shared = decay_shares_bound(sp)
        if 32 < self.d <= 64:
            return 12 if shared else 24
        if 16 < self.d <= 32 and shared:
            return 12
        return 0

    def _two_groups_fit_a_cu(self):
        """Trees in step at 17 <= d <= 32 with at least two 16-chain groups per CU: the group kernel's two waves and 75 KB of LDS
        let two groups share a CU, a wave per SIMD, where the split kernel's 83 KB admit one -- 8192 chains x 32-d, 7-leaf trees:
        group 1.78 against split 1.27 x 10^9 leapfrog steps/s; at 4096 chains 0.88 against 1.26, and at d <= 16 the split kernel fits
        three groups and stays ahead (2.42 against 1.41; profiles/r05_groups_per_cu.log).  A function of the shapes only."""
        plain, _, n = self._shape_facts()
        return plain and 16 < self.d <= 32 and n >= 32 * self._n_cu

    def run_tempered(self, n_run, base_mean, base_cov, logxi=0., u_0=None, n_warmup=500, max_treedepth=10, max_change=1000.,
                     target_accept=0.8, gamma=0.05, k=0.75, t_0=10., adapt_step_size=True, adapt_metric=True,
                     update_window=1, doubling=True, check=True):
        """TNUTS (samplers/tnuts.py; ``bfhip_tnuts_run``) with a Gaussian base density N(base_mean, base_cov) and
        ``logxi`` (``TNTrace(density_base=..., logxi=...)``, samplers/sample_trace.py:540-567).  ``u_0`` (n_chain,): the
        tempering coordinate at the start of a fresh run (default: standard normal draws, as the reference takes them
        from NumPy's global generator, base_hmc.py:241); later calls continue from the chains' own u.

        Returns (samples (n_chain, n_run, d), stats (n_chain, n_run, 11), stats_t (n_chain, n_run, 2) = u and weight)."""
        self.density.upload_if_needed()
        d = self.d
        mean = np.asarray(base_mean, dtype=np.float64).reshape(d)
        cov = np.asarray(base_cov, dtype=np.float64).reshape(d, d)
        prec = np.linalg.inv(cov)
        tp = _lib.Tempering()
        S = self.ctx.tensor(-prec)                     # log N = c0 + lin.x + x.S x / 2
        lin = self.ctx.tensor(prec @ mean)
        tp.base_S, tp.base_lin = S.data_ptr(), lin.data_ptr()
        tp.base_c0 = float(-0.5 * mean @ prec @ mean - 0.5 * (d * np.log(2 * np.pi) + np.linalg.slogdet(cov)[1]))
        tp.logxi = float(logxi)
        if getattr(self, 'tu', None) is None:
            if u_0 is None:
                u_0 = np.random.normal(0, 1, size=self.n_chain)
            self.tu = self.ctx.tensor(np.asarray(u_0, dtype=np.float64).reshape(self.n_chain))
        cfg = _lib.SamplerConfig()
        cfg.sampler, cfg.n_warmup, cfg.max_treedepth, cfg.n_int_step = 0, int(n_warmup), int(max_treedepth), 1
        cfg.max_change = float(max_change)
        cfg.target_accept, cfg.gamma, cfg.k, cfg.t_0 = float(target_accept), float(gamma), float(k), float(t_0)
        cfg.adapt_step_size, cfg.adapt_metric = int(bool(adapt_step_size)), int(bool(adapt_metric))
        cfg.update_window, cfg.doubling = int(update_window), int(bool(doubling))
        cfg.full_metric = int(self.full_metric)   # (the full-rank metric, cubic configs, d = 128, the pipeline density: bfhip_tnuts_gen.hip)
        cfg.metric_mat = self.mat.data_ptr() if self.full_metric else None
        n_run = int(n_run)
        samples = self.ctx.empty((self.n_chain, n_run, d))
        stats = self.ctx.empty((self.n_chain, n_run, _lib.STAT_STRIDE))
        stats_t = self.ctx.empty((self.n_chain, n_run, 2))
        _lib.check(self.ctx._lib.bfhip_tnuts_run(
            self.ctx.handle, C.byref(cfg), C.byref(tp), self.n_chain, self.i_iter + n_run, _ptr(self.rng), _ptr(self.sc),
            _ptr(self.vec), _ptr(self.tu), self.i_iter, n_run, _ptr(samples), _ptr(stats), _ptr(stats_t), _ptr(self.n_leapfrog)))
        self.i_iter += n_run
        self._answers = []
        if check:
            self.raise_on_error()
        return samples, stats, stats_t

    # ``hist_reduce``: None, or a callable summing an int64 device tensor over the ranks in place (``parallel.all_reduce_sum``;
    # ``sample()`` sets it when the chains are sharded).  With it the 'auto' layout is decided from the tree sizes of ALL
    # ranks' chains -- a collective per launch, the same launches on every rank -- so that every rank picks the same
    # layout and results do not depend on the number of ranks.  Without it each DeviceChains decides from its own chains.
    hist_reduce = None
    n_chain_rule = None   # set by sample() under torch.distributed: chains per rank on average (``_small_problem``)

    def _note_trees(self, stats, row0, row1, sampler, n_last=32, share=0.98):
        """Queue, behind the launch that wrote rows [row0, row1) of ``stats``, the answer to "did the chains run in step?":
        at least ``share`` of the NUTS trees of its last ``n_last`` iterations (all chains) had the most common size
        (``bfhip_tree_size_mode_share``: one small kernel).  The flag travels to a slot of a small pinned ring
        asynchronously; nothing here synchronises (except with ``hist_reduce``, which is a collective)."""
        torch = _torch()
        if not hasattr(self, '_answers'):
            self._answers = []   # one entry per launch, oldest first: bool, None (no answer: not NUTS) or (event, slot)
        if sampler != 'NUTS' or row1 <= row0:
            self._answers.append(None)
            return
        r0 = max(row0, row1 - n_last)
        if self.hist_reduce is not None:
            with torch.cuda.stream(self.ctx.stream):
                ts = stats[:, r0:row1, _lib.NSTATS.index('tree_size')].reshape(-1)
                # (binned as bf_tree_mode_kernel bins them: negative and NaN sizes go to bucket 4095)
                ts = torch.where((ts >= 0.) & (ts < 4095.), ts, torch.full_like(ts, 4095.)).to(torch.int64)
                hist = torch.zeros(4096 + 64, dtype=torch.int64, device=self.ctx.device)
                hist.scatter_add_(0, ts, torch.ones_like(ts))
                # (the chains by the size class of their leapfrogs in the window, as bf_tree_mode_kernel counts them)
                edges = torch.tensor(_lib.LAG_EDGES, dtype=torch.int64, device=self.ctx.device)
                cls = (torch.searchsorted(edges, ts.view(self.n_chain, -1).sum(1), right=True) - 1).clamp_(0, 63)
                hist.scatter_add_(0, 4096 + cls, torch.ones_like(cls))
            self.ctx.stream.synchronize()
            self.hist_reduce(hist)
            h = [int(v) for v in hist.cpu()]
            sizes, classes = h[:4096], h[4096:]
            mode = max(1, sizes.index(max(sizes)))
            n_all, tot = sum(classes), sum(i * v for i, v in enumerate(sizes))
            top = max(j for j in range(64) if classes[j]) if n_all else 0
            lag = 4096 if (tot > 0 and _lib.LAG_EDGES[top] * n_all >= 4 * tot) else 0   # (some chain lags far behind the rest: bf_tree_mode_kernel)
            self._answers.append((mode if max(sizes) >= share * sum(sizes) else 0) + lag)
            del self._answers[:-4]
            return
        if getattr(self, '_step_host', None) is None:
            self._step_host = torch.zeros(8, dtype=torch.int32, pin_memory=True)
            self._step_dev = torch.zeros(_lib.TREE_MODE_WORK, dtype=torch.int32, device=self.ctx.device)
            self._n_flag = 0
        _lib.check(self.ctx._lib.bfhip_tree_size_mode_share(self.ctx.handle, self.n_chain, stats.shape[1], _ptr(stats), r0,
                                                            row1 - r0, float(share), _ptr(self._step_dev)))
        slot = self._n_flag % 8
        self._n_flag += 1
        with torch.cuda.stream(self.ctx.stream):
            self._step_host[slot:slot + 1].copy_(self._step_dev[:1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.ctx.stream)
        self._answers.append((ev, slot))
        del self._answers[:-4]  # (at most the last two are ever read: the ring's 8 slots stay unambiguous)

    def _trees_in_step(self, lag=1):
        """The answer of ``_note_trees`` for the launch ``lag`` launches back (1 = the last one) -- the common tree size when the
        chains ran in step (+ 4096 when some chain builds far larger trees than the rest), else 0 --, waited for if it is still on its way; 0 when there is none (the first launches of a chain set, launches that were not NUTS).  A pure
        function of the launches so far: an answer that happens to have arrived early is not used before its turn."""
        ans = getattr(self, '_answers', [])
        if len(ans) < lag:
            return 0
        a = ans[-lag]
        if isinstance(a, tuple):
            ev, slot = a
            ev.synchronize()
            a = ans[-lag] = int(self._step_host[slot])
        return int(a or 0)

    def raise_on_error(self):
        """Synchronises; raises like the reference does for a chain that hit a fatal condition."""
        err = self.sc[:, _lib.SC_FIELDS.index('error')]
        bad = (err != 0).nonzero()
        if bad.numel():
            i = int(bad[0, 0])
            code = int(err[i])
            if code == 3:  # metrics.py:107-108
                raise ValueError('the input covariance is not positive definite.')
            if code == 1:  # base_hmc.py:72-76
                raise RuntimeError('Bad initial energy for chain #{}, please check the Hamiltonian.'.format(i))
            raise FloatingPointError("logp can't be nan (chain #{}).".format(i))  # nuts.py:201-202

    def covariance(self):
        """Per-chain metric covariance (n_chain, d, d): QuadMetricFull._cov, or diag(var) for the diagonal metric."""
        torch = _torch()
        if self.full_metric:
            return self.mat[:, 0].transpose(1, 2).contiguous()
        return torch.diag_embed(self.field('var'))

    def field(self, name):
        """One per-chain quantity by its reference name (device tensor view)."""
        if name in _lib.SC_FIELDS:
            return self.sc[:, _lib.SC_FIELDS.index(name)]
        return self.vec[:, _lib.VEC_FIELDS.index(name)]

    @property
    def total_leapfrog(self):
        return int(self.n_leapfrog.item())
