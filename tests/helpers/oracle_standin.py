"""TEST-ONLY stand-ins that let the host side of the package run without a GPU: the CPU oracle behind the interfaces of
``bayesfast_amd.chains.DeviceChains`` / ``bayesfast_amd.device.DeviceDensity``, and the oracle's NumPy restatement of the
fit behind ``bayesfast_amd.integrate._device_fit``.  ``install(monkeypatch)`` swaps them in for one test.  Nothing in the
package can reach this module; it exists so that the reference's own ``Recipe`` can be driven through
``bayesfast_amd.integrate`` in the build container (tests/test_integrate_reference.py), where there is no GPU."""
import numpy as np
import torch

from oracle import oracle as orc


class _CpuCtx:
    device = torch.device('cpu')
    stream = None

    def tensor(self, a, dtype=None):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a))
        return (t if dtype is None else t.to(dtype)).contiguous()

    def empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or torch.float64)

    zeros = empty


class OracleDensity:
    def __init__(self, spec, ctx=None):
        self.spec, self.d, self.ctx = spec, int(spec['d']), _CpuCtx()

    def upload_if_needed(self):
        pass

    def logp_and_grad(self, x, original_space=False):
        x = np.asarray(x, dtype=np.float64)
        lp, g = orc.logp_and_grad(self.spec, x.reshape(-1, self.d), original_space)
        return torch.from_numpy(np.atleast_1d(lp)), torch.from_numpy(np.atleast_2d(g))


class OracleTransform:
    """``DeviceDensity.constraint`` for a density with input scales, answered by the oracle's transforms (row by row)."""
    _FN = {'from_original': 'bfo_from_original_f', 'from_original_grad': 'bfo_from_original_j', 'from_original_grad2': 'bfo_from_original_jj',
           'to_original': 'bfo_to_original_f', 'to_original_grad': 'bfo_to_original_j', 'to_original_grad2': 'bfo_to_original_jj'}

    def __init__(self, ranges, hard_bounds):
        self.ranges = np.ascontiguousarray(ranges, dtype=np.float64)
        self.hb = np.ascontiguousarray(hard_bounds, dtype=np.uint8)

    def constraint(self, which, x):
        import ctypes as C
        dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        xt = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64)
        shape = xt.shape
        rows = np.ascontiguousarray(xt.reshape(-1, shape[-1]), dtype=np.float64)
        out = np.empty_like(rows)
        f = getattr(orc.lib(), self._FN[which])
        for i in range(rows.shape[0]):
            rc = f(rows[i].ctypes.data_as(dp), self.ranges.ctypes.data_as(dp), out[i].ctypes.data_as(dp), self.hb.ctypes.data_as(up),
                   rows.shape[1])
            if which.startswith('from') and rc:
                raise ValueError('variable #{} out of bound.'.format(rc - 1))
        return torch.from_numpy(out.reshape(shape))


class OracleChains:
    """The constructor and ``run`` arguments of ``DeviceChains``; chains are ``oracle.ChainSet`` on the same xoshiro streams
    (seed, first_stream + i), so what runs here is what the device tests compare the kernels with."""
    hist_reduce = None

    def __init__(self, density, x_0, seed=0, first_stream=0, step_size=1., metric=None, initial_mean=None,
                 initial_weight=10., adapt_window=60):
        self.density, self.ctx = density, density.ctx
        self.x_0 = np.ascontiguousarray(np.asarray(x_0, dtype=np.float64))
        self.n_chain, self.d = self.x_0.shape
        self._init = dict(seed=seed, first_stream=first_stream, step_size=step_size, metric=metric, initial_mean=initial_mean,
                          initial_weight=initial_weight, adapt_window=adapt_window)
        self.full_metric = (isinstance(metric, str) and metric == 'full') or (metric is not None and np.ndim(metric) == 2)
        self.cs, self.i_iter, self.total_leapfrog = None, 0, 0

    def run(self, n_run, sampler='NUTS', n_warmup=500, max_treedepth=10, n_int_step=32, max_change=1000., target_accept=0.8,
            gamma=0.05, k=0.75, t_0=10., adapt_step_size=True, adapt_metric=True, update_window=1, doubling=True,
            samples=None, stats=None, check=True, launch_iters=250, layout='auto'):
        if sampler != 'NUTS':
            raise NotImplementedError('the stand-in runs NUTS.')
        if self.cs is None:
            i = self._init
            metric = None if isinstance(i['metric'], str) and i['metric'] == 'diag' else i['metric']
            self.cs = orc.ChainSet(self.density.spec, self.x_0, i['seed'], first_stream=i['first_stream'],
                                   step_size=i['step_size'], adapt_step_size=adapt_step_size, target_accept=target_accept,
                                   gamma=gamma, k=k, t_0=t_0, metric=metric, adapt_metric=adapt_metric,
                                   initial_mean=i['initial_mean'], initial_weight=i['initial_weight'],
                                   adapt_window=i['adapt_window'], update_window=update_window, doubling=doubling)
        s, st, n = self.cs.run(int(n_run), int(n_warmup), max_treedepth=max_treedepth, max_change=max_change)
        self.i_iter += int(n_run)
        self.total_leapfrog += n
        self.last_layout = 'oracle'
        return torch.from_numpy(s), torch.from_numpy(np.stack([st[k] for k in orc.NSTATS], -1))

    def run_tempered(self, n_run, base_mean, base_cov, logxi=0., u_0=None, n_warmup=500, max_treedepth=10, max_change=1000.,
                     target_accept=0.8, gamma=0.05, k=0.75, t_0=10., adapt_step_size=True, adapt_metric=True, update_window=1,
                     doubling=True, check=True):
        """``DeviceChains.run_tempered`` answered by the oracle's TNUTS (one chain after the other, same xoshiro streams)."""
        i = self._init
        if getattr(self, 'tchains', None) is None:
            self.tchains = [orc.Chain(self.x_0[c], step_size=i['step_size'], adapt_step_size=adapt_step_size, target_accept=target_accept,
                                      gamma=gamma, k=k, t_0=t_0, metric=None, adapt_metric=adapt_metric, initial_mean=i['initial_mean'],
                                      initial_weight=i['initial_weight'], adapt_window=i['adapt_window'], update_window=update_window,
                                      doubling=doubling) for c in range(self.n_chain)]
            self.trngs = [orc.make_rng('xoshiro', seed=i['seed'], stream=i['first_stream'] + c) for c in range(self.n_chain)]
            self.tu = np.random.default_rng(i['seed']).normal(size=self.n_chain) if u_0 is None else np.asarray(u_0, dtype=np.float64)
        base = orc.gaussian_base_spec(np.asarray(base_mean, dtype=np.float64), np.asarray(base_cov, dtype=np.float64))
        ss, sts, stts = [], [], []
        for c in range(self.n_chain):
            s, st, u = orc.tnuts_run(self.density.spec, base, float(logxi), self.tchains[c], self.trngs[c], float(self.tu[c]), int(n_run),
                                     int(n_warmup), max_treedepth=max_treedepth, max_change=max_change)
            self.tu[c] = u
            ss.append(s)
            sts.append(np.stack([st[k] for k in orc.NSTATS], -1))
            stts.append(np.stack([st['u'], st['weight']], -1))
            self.total_leapfrog += int(st['tree_size'].sum())
        self.i_iter += int(n_run)
        self.cs = type('CS', (), {'chains': self.tchains})()
        return torch.from_numpy(np.stack(ss)), torch.from_numpy(np.stack(sts)), torch.from_numpy(np.stack(stts))

    def raise_on_error(self):
        pass

    def field(self, name):
        if name in ('log_step', 'log_bar', 'hbar', 'count'):
            return torch.tensor([float(c.scalar(name)) for c in self.cs.chains], dtype=torch.float64)
        return torch.from_numpy(np.stack([c.vec(name) for c in self.cs.chains]))

    def covariance(self):
        if self.full_metric:
            return torch.from_numpy(np.stack([c.mat('cov') for c in self.cs.chains]))
        return torch.diag_embed(self.field('var'))


def oracle_fit(mirror, x, y, logp, w):
    """``bayesfast_amd.PolyModel.fit`` answered by the oracle's restatement of modules/poly.py:505-589 (scipy lstsq)."""
    poly = dict(input_size=mirror.input_size, output_size=mirror.output_size,
                configs=[dict(order=c.order, input_mask=np.array(c._input_mask), output_mask=np.array(c._output_mask))
                         for c in mirror.configs])
    out = orc.poly_fit(poly, x, y, logp, w, bound_options=dict(use_bound=False))
    for c, f in zip(mirror.configs, out['configs']):
        c._coef = np.array(f['coef'], dtype=np.float64)


def _cpu_sort(a):
    k = torch.where(a == 0., torch.zeros_like(a), a).contiguous().view(torch.int64)  # -0 == +0, as numpy sorts them
    neg = k < 0
    key = torch.where(neg, ~k, k | (-2**63)) ^ (-2**63)  # the order-preserving key of bfhip_sort_keys, as signed values
    order = torch.sort(key, stable=True).indices
    return key[order], order


def install(monkeypatch):
    import bayesfast_amd.chains as chains
    import bayesfast_amd.core.refit as refit
    monkeypatch.setattr(refit, 'device_sort', _cpu_sort)
    monkeypatch.setattr(refit, '_device_count', lambda keys, q, upper: torch.searchsorted(keys, q, right=bool(upper)))
    import bayesfast_amd.integrate as integrate
    from bayesfast_amd.core.density import SurrogateDensity
    monkeypatch.setattr(chains, 'DeviceChains', OracleChains)
    monkeypatch.setattr(SurrogateDensity, 'device', lambda self, ctx=None: OracleDensity(self.spec()))
    monkeypatch.setattr(SurrogateDensity, '_transform_device', lambda self: OracleTransform(self._input_scales, self._hard_bounds))
    monkeypatch.setattr(integrate, '_device_fit', oracle_fit)
