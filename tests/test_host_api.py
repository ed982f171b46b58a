"""CPU-only tests of the host side that mirrors the reference interface (no compute calls: no GPU here)."""
import copy
import os
import pickle

import numpy as np
import pytest

from bayesfast_amd import PolyConfig, PolyModel, NTrace, HTrace, TraceTuple, SurrogateDensity
from bayesfast_amd import _lib, parallel
from oracle import oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def test_polyconfig_shapes_and_errors():
    with pytest.raises(ValueError):
        PolyConfig('quartic')
    c = PolyConfig('cubic-3', input_mask=[4, 1, 1, 2], output_mask=[0])
    assert list(c.input_mask) == [1, 2, 4] and c.input_size == 3 and c.output_size == 1
    assert c._A_shape == (1, 3, 3, 3) and c._a_shape == (1,)
    assert PolyConfig('quadratic', [0, 1, 2], [0, 1])._a_shape == (6,)
    with pytest.raises(RuntimeError):
        PolyConfig('linear')._A_shape


@pytest.mark.parametrize('order', ['linear', 'quadratic', 'cubic-2', 'cubic-3'])
def test_polyconfig_set_matches_reference_scatter(order):
    """PolyConfig._set packs like modules/_poly.pyx:183-214 (checked through the oracle's restatement)."""
    n = 5
    c = PolyConfig(order, input_mask=np.arange(n), output_mask=[0, 1])
    a = np.random.default_rng(0).normal(size=c._a_shape)
    c._set(a, 1)
    assert np.array_equal(c._coef[1], orc.dense_coef(order, a, n))
    assert not c._coef[0].any()
    with pytest.raises(ValueError):
        c._set(a[:-1], 0)
    with pytest.raises(ValueError):
        c._set(a, 2)


def test_polymodel_construction_recipe_and_nparam():
    m = PolyModel('quadratic', input_size=64, output_size=1)
    assert [c.order for c in m.configs] == ['linear', 'quadratic'] and m.n_param == 2145
    assert m.bound_options.use_bound and m.bound_options.alpha_p == 100.
    assert PolyModel('cubic-3', input_size=4, output_size=1).n_param == 5 + 10 + 16 + 4
    with pytest.raises(ValueError):  # two quadratic configs on the same output (modules/poly.py:308-313)
        PolyModel([PolyConfig('quadratic'), PolyConfig('quadratic')], input_size=3, output_size=1)
    with pytest.raises(ValueError):  # output 1 has no config (:334-337)
        PolyModel([PolyConfig('linear', output_mask=[0])], input_size=3, output_size=2)
    with pytest.raises(ValueError):
        PolyModel('quintic', input_size=3, output_size=1)
    with pytest.raises(ValueError):
        m.set_bound_options(alpha=None, alpha_p=None)
    with pytest.raises(ValueError):
        m.fit(np.zeros((10, 64)), np.zeros((10, 1)))  # fewer points than parameters (:521-523)
    with pytest.raises(ValueError):
        m.fit(np.zeros((3000, 63)), np.zeros((3000, 1)))
    with pytest.raises(RuntimeError):
        m.poly_spec()  # not fitted


def test_polymodel_survives_deepcopy_and_pickle():
    """Recipe deep-copies and pickles surrogates (core/recipe.py:822,1163): no device handles in the state."""
    m = PolyModel('quadratic', input_size=3, output_size=1, input_scales=np.array([[0., 2.]] * 3))
    for c in m.configs:
        c._set(np.arange(c._a_shape[0], dtype=float), 0)
    m._mu, m._hess, m._alpha, m._f_mu = np.zeros(3), np.eye(3), 2., np.array([1.])
    for m2 in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert np.array_equal(m2.configs[1]._coef, m.configs[1]._coef)
        assert m2.poly_spec()['alpha'] == 2. and m2.scope == (0, 1)
        assert np.array_equal(m2._input_scales_diff, [2., 2., 2.])


def test_trace_options_validation_and_defaults():
    t = NTrace()
    assert (t.n_chain, t.n_iter, t.n_warmup, t.max_treedepth, t.max_change) == (4, 1500, 500, 10, 1000.)
    assert t.run_kwargs()['target_accept'] == 0.8 and HTrace().n_int_step == 32
    for bad in (dict(n_chain=0), dict(n_iter=10, n_warmup=10), dict(max_treedepth=0), dict(max_treedepth=99),
                dict(step_size=-1.), dict(target_accept=1.5), dict(max_change=0.), dict(metric='banana')):
        with pytest.raises(ValueError):
            NTrace(**bad)
    assert NTrace(metric='full')._metric == 'full' and NTrace(metric=np.eye(3))._metric.shape == (3, 3)
    with pytest.raises(ValueError):
        NTrace(metric=np.ones((2, 3)))
    assert NTrace(random_generator=7).seed() == 7


def test_tracetuple_get_semantics():
    """get() drops the warm-up by default and flattens chains (samplers/sample_trace.py:277-303,753-785)."""
    tr = NTrace(n_chain=3, n_iter=10, n_warmup=4)
    C, n, d = 3, 10, 2
    s = np.arange(C * n * d, dtype=float).reshape(C, n, d)
    st = np.zeros((C, n, _lib.STAT_STRIDE))
    st[:, :, _lib.NSTATS.index('tree_size')] = 3
    st[:, :, 0] = np.arange(n)
    tt = TraceTuple(tr, s, st, s * 10, st[:, :, 0] - 1.)
    assert tt.get().shape == (C * 6, d) and np.array_equal(tt.get()[0], s[0, 4] * 10)
    assert tt.get(original_space=False, flatten=False).shape == (C, 6, d)
    assert tt.get(include_warmup=True, return_type='logp').shape == (C * n,)
    assert len(tt) == 3 and tt[1].samples.shape == (n, d) and [t.chain_id for t in tt] == [0, 1, 2]
    assert tt[0].n_call == 3 * (n - 1) + n + 1 and tt.n_call == 3 * tt[0].n_call
    assert tt[2].stats.get()['tree_size'] == [3] * 6 and tt[0].stats.n_iter == n
    with pytest.raises(ValueError):
        tt.get(since_iter=9)
    with pytest.raises(ValueError):
        tt.get(return_type='weights')


def test_surrogate_density_argument_checks():
    with pytest.raises(ValueError):
        SurrogateDensity(PolyModel('quadratic', input_size=4, output_size=2))
    with pytest.raises(ValueError):
        SurrogateDensity(PolyModel('quadratic', input_size=4, output_size=1), input_scales=np.ones((3, 2)))
    den = SurrogateDensity(PolyModel('quadratic', input_size=4, output_size=1), input_scales=np.array([[0., 1.]] * 4),
                           hard_bounds=True)
    assert den._hard_bounds.shape == (4, 2) and den._hard_bounds.all() and den.input_size == 4


def test_shard_range_partitions_chains():
    for n, ws in ((4096, 8), (10, 3), (5, 8), (1, 1)):
        r = [parallel.shard_range(n, k, ws) for k in range(ws)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
        assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def _gather_worker(rank, ws, port, n_chain, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        b, e = parallel.shard_range(n_chain, rank, ws)
        full = torch.arange(n_chain * 3 * 2, dtype=torch.float64).reshape(n_chain, 3, 2)
        out = parallel.all_gather_chains(full[b:e].clone(), n_chain)
        dev_ok = parallel.local_device(8, env={'LOCAL_RANK': str(rank)}) == rank  # one process per GPU
        q.put((rank, bool(torch.equal(out, full)) and dev_ok, parallel.world()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_chain', [8, 7])
def test_all_gather_chains_gloo_world2(n_chain):
    """The refit exchange step on 2 CPU processes (gloo): even and ragged shards."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_chain) % 2000
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, n_chain, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in ps]
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [(0, 2), (1, 2)]


def test_local_device_selection():
    """One process per GPU: the rank's device comes from LOCAL_RANK when a process group exists (here: none -> None)."""
    from bayesfast_amd import parallel
    assert parallel.local_device(8, env={'LOCAL_RANK': '3'}) is None  # no process group: nothing to decide
    assert parallel.shard_range(10, 3, 4) == (8, 10)


def test_systematic_resampler_matches_reference_pattern():
    """SystematicResampler (utils/misc.py:21-108): ranks evenly spaced between the node percentiles of the sorted
    array, weights per interval, uniqueness check."""
    from bayesfast_amd import SystematicResampler
    a = np.random.default_rng(0).normal(size=1000)
    i = SystematicResampler()(a, 10)
    order = np.argsort(a)
    expect = order[np.linspace(1. * 999 / 100, 100. * 999 / 100, 10, True).astype(int)]
    assert np.array_equal(i, expect)
    i2 = SystematicResampler(nodes=(0., 50., 100.), weights=(1., 3.))(a, 8)   # 2 from the lower half, 6 from the upper
    ranks = np.argsort(order)[i2]
    assert np.array_equal(ranks, [0, 249, 499, 599, 699, 799, 899, 999])
    with pytest.raises(RuntimeError):
        SystematicResampler()(a[:5], 10)          # repeated indices
    with pytest.warns(RuntimeWarning):
        SystematicResampler(require_unique=False)(a[:5], 10)
    for bad in (dict(nodes=(5.,)), dict(nodes=(50., 10.)), dict(nodes=(0., 101.)), dict(weights=(1., 2.))):
        with pytest.raises(ValueError):
            SystematicResampler(**bad)


def test_select_fit_points_logp_cutoff_and_importance_weights():
    """Refit glue of Recipe._sam_step (core/recipe.py:1060-1155) and the truncated weights of PostStep (:1289-1296)."""
    from bayesfast_amd import select_fit_points, importance_weights
    rng = np.random.default_rng(1)
    x = rng.normal(size=(5000, 3))
    logq = -0.5 * np.sum(x**2, 1)
    calls = []

    def logp_true(z):  # the true model agrees with the surrogate, except that a slab of points is "bad"
        calls.append(len(z))
        return np.where(z[:, 0] > 1., -1e3, -0.5 * np.sum(z**2, 1))

    xf, lf, n_calls = select_fit_points(x, logq, logp_true, 400, logp_cutoff=False)
    assert xf.shape == (400, 3) and n_calls == 400 and np.allclose(lf, logp_true(xf))
    calls.clear()
    import warnings
    with warnings.catch_warnings():  # supplementary rounds until alpha_min * n_eval good points remain
        warnings.simplefilter('ignore')
        xf, lf, n_calls = select_fit_points(x, logq, logp_true, 400, alpha_min=0.95)
    assert xf.shape[0] >= int(0.95 * 400) and np.all(lf > -1e3) and np.all(xf[:, 0] <= 1.)
    assert n_calls == sum(calls) and len(calls) >= 2
    with pytest.raises(RuntimeError):
        select_fit_points(x, logq, lambda z: np.full(len(z), -np.inf), 100)   # f_good == 0
    with pytest.raises(RuntimeError):
        select_fit_points(x[:50], logq[:50], logp_true, 100)                  # not enough points
    ratio = np.ones(logq.size)
    ratio[7] = 1e6                                   # one wild weight is clipped at mean(w) n^k_trunc
    w, wt = importance_weights(logq + np.log(ratio), logq, k_trunc=0.25)
    assert np.allclose(w, ratio) and wt.max() == pytest.approx(np.mean(w) * logq.size**0.25) and wt[0] == pytest.approx(1.)
    assert np.array_equal(*importance_weights(logq, logq - 1., k_trunc=-1))


def test_trace_fields_recipe_pokes_keep_the_reference_meaning():
    """core/recipe.py:967-975,1028-1045 reads and writes sample_trace.x_0, ._x_0_transformed, ._metric ('diag' / 'full' /
    array) and ._step_size (None until filled from the previous round) directly."""
    from bayesfast_amd import NTrace
    t = NTrace(n_chain=4, n_iter=20, n_warmup=5, step_size=None)
    assert t.x_0 is None and t._x_0_transformed is False
    assert t._metric == 'diag' and t._step_size is None
    t._step_size = 0.3
    t._metric = np.array([1., 2., 3.])
    t.x_0 = np.zeros((4, 3))
    t._x_0_transformed = True
    assert NTrace(metric='full')._metric == 'full'
    with pytest.raises(ValueError):
        NTrace(step_size=-1.)


class _RefConfig:
    def __init__(self, c):
        self.order, self.input_mask, self.output_mask, self._coef = c.order, c._input_mask, c._output_mask, c._coef


class _RefPoly:
    """The attributes adapters.py reads from a reference PolyModel, taken from this package's own object."""

    def __init__(self, pm):
        self.configs = [_RefConfig(c) for c in pm.configs]
        for k in ('_input_size', '_output_size', '_use_bound', '_all_linear', '_mu', '_hess', '_alpha', '_f_mu', '_input_scales',
                  '_input_scales_diff'):
            setattr(self, k, getattr(pm, k, None))


class _RefDensity:
    def __init__(self, den):
        self._surrogate_list = [_RefPoly(den.surrogate)]
        self.input_size = den._d
        for k in ('_input_scales', '_hard_bounds', '_use_decay', '_mu', '_hess', '_alpha_2', '_gamma'):
            setattr(self, k, getattr(den, k, None))


def _fitted_density_on_host():
    """A SurrogateDensity with hand-set (not fitted: no GPU here) coefficients, bound, scales and decay."""
    from bayesfast_amd import PolyModel, SurrogateDensity
    rng = np.random.default_rng(4)
    d = 4
    su = PolyModel('quadratic', input_size=d, output_size=1)
    for c in su.configs:  # PolyConfig._set: the packed coefficients of output 0 (modules/poly.py:131-158)
        a = rng.normal(size=c._a_shape) * 0.1
        c._set(a, 0)
    su._mu, su._hess, su._alpha, su._f_mu = np.zeros(d), np.eye(d), 5., np.array([-1.])
    den = SurrogateDensity(su, input_scales=np.stack([-8. * np.ones(d), 9. * np.ones(d)], 1), hard_bounds=np.array([[1, 1], [0, 1], [1, 0], [0, 0]]),
                           decay_options=dict(use_decay=True))
    den._mu, den._hess, den._alpha_2, den._alpha = np.zeros(d), np.eye(d), 30., 30.**0.5
    return den


def test_adapters_read_a_reference_density_by_duck_typing():
    """adapters.density_spec_from_reference / surrogate_density_from_reference on an object that exposes the reference's
    attribute names give the same device description as the package's own density."""
    from bayesfast_amd import adapters
    den = _fitted_density_on_host()
    ref = _RefDensity(den)
    a, b = adapters.density_spec_from_reference(ref), den.spec()
    c = adapters.surrogate_density_from_reference(ref).spec()

    def same(u, v):
        if isinstance(u, dict):
            assert set(u) == set(v)
            [same(u[k], v[k]) for k in u]
        elif isinstance(u, (list, tuple)):
            assert len(u) == len(v)
            [same(x, y) for x, y in zip(u, v)]
        elif u is None or isinstance(u, str):
            assert u == v
        else:
            assert np.array_equal(np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64))

    same(a, b)
    same(c, b)


def test_gaussian_link_in_the_oracle_and_own_refit_loop_on_the_standin(monkeypatch):
    """The link (Gaussian likelihood of the surrogate's single output, core/density.py:527-560) in the CPU oracle against
    the composition written out, and the package's own config-1 loop (tests/helpers/donut.py) with the oracle stand-in
    behind the device entry points: host logic of fit(x, logp, y=...), sample(), gather(), select_fit_points, warm starts.
    The same loop runs on the GPU in tests/test_gpu_recipe.py."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'helpers'))
    import donut
    import oracle_standin
    from oracle import oracle as orc
    d = 3
    rng = np.random.default_rng(0)
    poly = dict(input_size=d, output_size=1, use_bound=False,
                configs=[dict(order='linear', input_mask=np.arange(d), output_mask=np.array([0]), coef=rng.normal(size=(1, d + 1))),
                         dict(order='quadratic', input_mask=np.arange(d), output_mask=np.array([0]), coef=rng.normal(size=(1, d, d)))])
    spec = dict(d=d, ranges=np.array([[-3., 4.]] * d), hard_bounds=np.array([[1, 1], [0, 0], [1, 0]], np.uint8), su_lo=None,
                su_diff=None, poly=poly, use_decay=False)
    x = rng.normal(size=(5, d))
    m, gm = orc.logp_and_grad(spec, x, original_space=True)
    lp, g = orc.logp_and_grad(dict(spec, link=dict(kind='gaussian', y=5., prec=4., logp0=0.)), x, original_space=True)
    np.testing.assert_allclose(lp, -(m - 5.)**2 / 0.5, rtol=1e-14)
    np.testing.assert_allclose(g, (-2. * (m - 5.) / 0.5)[:, None] * gm, rtol=1e-14)
    # transformed space: the log-Jacobian is added after the link, as density.py:747-750 does
    mt, _ = orc.logp_and_grad(spec, x, original_space=False)
    lpt, _ = orc.logp_and_grad(dict(spec, link=dict(kind='gaussian', y=5., prec=4., logp0=0.)), x, original_space=False)
    spec_o = dict(spec, ranges=None, hard_bounds=None)
    assert not np.allclose(lpt, -(mt - 5.)**2 / 0.5)
    oracle_standin.install(monkeypatch)
    import bayesfast_amd.modules.poly as mp
    monkeypatch.setattr(mp.PolyModel, 'fit', lambda self, x, y, logp=None, w=None: oracle_standin.oracle_fit(self, x, y, logp, w))
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'recipe.npz'))
    ring, tt = donut.own_refit_loop(z['step0.x_fit'], float(z['step0.step_size']), n_steps=8)
    assert abs(ring[-2:, 0].mean() - 5.02) < 0.1 and abs(ring[-2:, 1].mean() - 0.49) < 0.08 and ring[-1, 2] < 0.3, ring


def test_reference_timing_fixture_is_present_and_sane():
    """tests/golden/reference_timing.json (tools/time_reference.py: the as-shipped reference timed in the build container,
    BASELINE.md section 3): ~1e4 leapfrog steps/s/core, the Python pipeline overhead dominating the Cython kernels."""
    import json
    import os
    rt = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_timing.json')))
    assert 2e3 < rt['leapfrog_steps_per_sec_per_core'] < 1e5
    assert rt['quadratic_kernels_us'] < rt['polymodel_fun_and_jac_us'] < rt['logp_and_grad_us'] <= rt['leapfrog_step_us'] * 1.2
    assert rt['host']['logical_cpus'] >= 1 and rt['polymodel_fit_shape'] == [4290, 2145]


def test_sobol_normal_points_equal_the_references():
    """utils/sobol.py against fixtures of the reference's own generator (utils/sobol.py:12-61, utils/_sobol.pyx): the default
    starting points of sample() (core/sample.py:106-113) are the reference's, bit for bit."""
    import os
    from bayesfast_amd.utils.sobol import multivariate_normal, uniform
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sobol.npz'))
    for k in z.files:
        if k.startswith('normal_') and 'cov' not in k:
            d, n = (int(v) for v in k.split('_')[1:])
            assert np.array_equal(multivariate_normal(np.zeros(d), np.eye(d), n), z[k]), k
    assert np.array_equal(uniform(np.zeros(3), np.ones(3), 10), z['uniform_3_10'])
    np.testing.assert_allclose(multivariate_normal(z['mean'], z['cov'], 12), z['normal_cov_4_12'], rtol=1e-13, atol=1e-13)
    assert np.array_equal(uniform(np.zeros(3), np.ones(3), 4, skip=7), z['uniform_3_10'][6:])
    with pytest.raises(ValueError):
        uniform(np.zeros(3), np.ones(2), 4)
    with pytest.raises(ValueError):
        multivariate_normal(np.zeros(3), np.eye(2), 4)


def test_blas_single_thread_context_limits_and_restores():
    """utils/threads.blas_single_thread: BLAS calls inside run on one thread, the pool's size comes back after (the host
    linear algebra of the fit must not leave spinning workers beside the GPU runtime's threads)."""
    from bayesfast_amd.utils.threads import blas_single_thread
    threadpoolctl = pytest.importorskip('threadpoolctl')
    before = [p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas']
    with blas_single_thread():
        inside = [p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas']
        a = np.random.default_rng(0).normal(size=(300, 40))
        assert np.allclose(a.T @ a, np.einsum('ij,ik->jk', a, a))
    after = [p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas']
    assert inside and all(n == 1 for n in inside) and after == before


def test_layout_rules_are_functions_of_the_shapes_only():
    """DeviceChains' shape rules for the automatic layout (tools/dispatch_sweep.py measured them): the wave layout for few chains,
    up to eight per CU with the decay term or the constraint transform (the pipelined kernel's feature sets), never for both
    together or for other densities; the lane layouts whatever the trees at d <= 32 from sixteen chains per CU."""
    import types
    from bayesfast_amd.chains import DeviceChains

    def stub(d, n, spec_extra=None, configs=('linear', 'quadratic'), n_rule=None, full=False):
        sp = dict(poly=dict(use_bound=True, configs=[dict(order=o) for o in configs]), **(spec_extra or {}))
        o = types.SimpleNamespace(density=types.SimpleNamespace(spec=sp), d=d, n_chain=n, n_chain_rule=n_rule, full_metric=full, _n_cu=256, ctx=None)
        for name in ('_shape_facts', '_small_problem', '_lanes_whatever_the_trees'):
            setattr(o, name, types.MethodType(getattr(DeviceChains, name), o))
        return o

    dec = dict(use_decay=True)
    tr = dict(ranges=np.zeros((64, 2)))
    assert stub(64, 1024)._small_problem() and not stub(64, 2048)._small_problem()
    assert stub(32, 1024)._small_problem() and not stub(32, 2048)._small_problem()
    assert stub(64, 2048, dec)._small_problem() and not stub(64, 4096, dec)._small_problem()
    assert stub(32, 2048, tr)._small_problem() and not stub(32, 4096, tr)._small_problem()
    assert not stub(64, 512, dict(dec, **tr))._small_problem()                      # no pipelined instantiation with both
    assert not stub(64, 512, configs=('linear', 'quadratic', 'cubic-2'))._small_problem()
    assert not stub(64, 512, dec, full=True)._small_problem()
    assert stub(64, 16384, dec, n_rule=2048)._small_problem()                       # sharded: chains per rank
    assert stub(16, 4096)._lanes_whatever_the_trees() and stub(32, 8192)._lanes_whatever_the_trees()
    assert not stub(32, 2048)._lanes_whatever_the_trees() and not stub(64, 4096)._lanes_whatever_the_trees()
    assert not stub(16, 4096, dec)._lanes_whatever_the_trees()


def test_input_scales_fold_into_a_quadratic_surrogate_and_its_bound():
    """device.density_desc_from_spec folds Surrogate.input_scales (module.py:190-226) into a linear + quadratic surrogate's
    coefficients and its bound's centre and Hessian: the folded polynomial at x equals the scaled one at (x - lo) / diff, the
    bound's radius is the same number, gradients differ by the factor 1 / diff -- and cubic configs keep the scaling."""
    from bayesfast_amd.device import density_desc_from_spec
    rng = np.random.default_rng(4)
    d = 7
    lo, diff = rng.normal(size=d), rng.uniform(0.3, 4., size=d)
    im = np.arange(d)
    cl, cq = rng.normal(size=(1, d + 1)), rng.normal(size=(1, d, d))
    mu, hh = rng.normal(size=d) * 0.2, rng.normal(size=(d, d))
    hess = hh @ hh.T + d * np.eye(d)
    poly = dict(input_size=d, output_size=1, use_bound=True, mu=mu, hess=hess, alpha=3., f_mu=np.array([0.7]),
                configs=[dict(order='linear', input_mask=im, output_mask=np.arange(1), coef=cl),
                         dict(order='quadratic', input_mask=im, output_mask=np.arange(1), coef=cq)])
    spec = dict(d=d, ranges=None, hard_bounds=None, su_lo=lo, su_diff=diff, poly=poly, use_decay=False)
    ds, keep = density_desc_from_spec(spec)
    assert not ds.su_lo and not ds.su_diff
    arr = lambda p, n: np.ctypeslib.as_array(p, shape=(n,)).copy()
    lin, quad = arr(ds.lin, d), arr(ds.quad, d * d).reshape(d, d)
    mu2, h2 = arr(ds.mu, d), arr(ds.hess, d * d).reshape(d, d)
    iu = np.triu_indices(d)
    A = np.zeros((d, d))
    A[iu] = cq[0][iu]
    for x in rng.normal(size=(5, d)) * 3.:
        xs = (x - lo) / diff
        f_ref = cl[0, 0] + cl[0, 1:] @ xs + xs @ A @ xs
        f_fold = ds.c0 + lin @ x + x @ np.triu(quad) @ x
        assert abs(f_fold - f_ref) < 1e-11 * (1. + abs(f_ref))
        g_ref = (cl[0, 1:] + (A + A.T) @ xs) / diff
        g_fold = lin + (np.triu(quad) + np.triu(quad).T) @ x
        np.testing.assert_allclose(g_fold, g_ref, rtol=1e-11, atol=1e-11)
        b_ref = (xs - mu) @ hess @ (xs - mu)
        b_fold = (x - mu2) @ h2 @ (x - mu2)
        assert abs(b_fold - b_ref) < 1e-11 * (1. + abs(b_ref))
    assert ds.alpha == 3. and ds.f_mu == 0.7
    # cubic configs: the third-order expansion around x = 0
    c2, c3 = rng.normal(size=(1, d, d)), rng.normal(size=(1, d, d, d))
    poly3 = dict(poly, configs=poly['configs'] + [dict(order='cubic-2', input_mask=im, output_mask=np.arange(1), coef=c2),
                                                  dict(order='cubic-3', input_mask=im, output_mask=np.arange(1), coef=c3)])
    ds3, keep3 = density_desc_from_spec(dict(spec, poly=poly3))
    assert not ds3.su_lo and not ds3.su_diff
    lin3, quad3 = arr(ds3.lin, d), arr(ds3.quad, d * d).reshape(d, d)
    q2, q3 = arr(ds3.cubic2, d * d).reshape(d, d), arr(ds3.cubic3, d**3).reshape(d, d, d)
    j, k, l = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), indexing='ij')
    t3 = np.where((j < k) & (k < l), c3[0], 0.)

    def cubic_poly(c0_, l_, q_, c2_, c3_, z):   # the reference's conventions (modules/_poly.pyx:13-137)
        return c0_ + l_ @ z + z @ np.triu(q_) @ z + (z * z) @ (c2_ @ z) + np.einsum('jkl,j,k,l->', c3_, z, z, z)

    for x in rng.normal(size=(5, d)) * 2.:
        xs = (x - lo) / diff
        f_ref = cubic_poly(cl[0, 0], cl[0, 1:], A, c2[0], t3, xs)
        f_fold = cubic_poly(ds3.c0, lin3, quad3, q2, q3, x)
        assert abs(f_fold - f_ref) < 1e-10 * (1. + abs(f_ref)), (f_fold, f_ref)
        h = 1e-6   # (the gradient by differences of the folded form against differences of the scaled one)
        for i in (0, d - 1):
            e = np.zeros(d)
            e[i] = h
            gf = (cubic_poly(ds3.c0, lin3, quad3, q2, q3, x + e) - cubic_poly(ds3.c0, lin3, quad3, q2, q3, x - e)) / (2 * h)
            gr = (cubic_poly(cl[0, 0], cl[0, 1:], A, c2[0], t3, (x + e - lo) / diff) - cubic_poly(cl[0, 0], cl[0, 1:], A, c2[0], t3, (x - e - lo) / diff)) / (2 * h)
            assert abs(gf - gr) < 1e-5 * (1. + abs(gr))
    np.testing.assert_allclose(arr(ds3.mu, d), lo + diff * mu)
    # ... and so does a range far from the origin in units of its width (cancellation in the folded form)
    ds4, keep4 = density_desc_from_spec(dict(spec, su_lo=lo + 1000.))
    assert bool(ds4.su_lo) and bool(ds4.su_diff)


def test_decay_shares_bound_and_the_flop_count_follow_the_arrays():
    """The decay term's statistics are the bound's when both come from the same points (core/density.py:796-811 and
    modules/poly.py:262-276 are the same statements): `decay_shares_bound` is the comparison the upload makes, and the algorithmic
    flops of a leapfrog step count the shared product once."""
    from bayesfast_amd.workloads import decay_shares_bound, flops_per_leapfrog_spec
    rng = np.random.default_rng(1)
    d = 6
    xs = rng.normal(size=(200, d))
    cfgs = [dict(order='linear', input_mask=np.arange(d), output_mask=np.arange(1), coef=rng.normal(size=(1, d + 1))),
            dict(order='quadratic', input_mask=np.arange(d), output_mask=np.arange(1), coef=np.triu(rng.normal(size=(d, d)))[None] * -0.3)]
    poly = dict(input_size=d, output_size=1, configs=cfgs, use_bound=False)
    poly.update(orc.set_bound(poly, xs, rng.normal(size=xs.shape[0]), dict(alpha_p=80.)))
    spec = dict(d=d, poly=poly)
    assert not decay_shares_bound(spec) and flops_per_leapfrog_spec(spec) == 4 * d * d
    same = dict(spec, **orc.set_decay(xs, alpha_p=150.))
    assert np.array_equal(same['decay_hess'], poly['hess']) and np.array_equal(same['decay_mu'], poly['mu'])
    assert decay_shares_bound(same) and flops_per_leapfrog_spec(same) == 4 * d * d
    other = dict(spec, **orc.set_decay(xs * 0.9, alpha_p=150.))
    assert not decay_shares_bound(other) and flops_per_leapfrog_spec(other) == 6 * d * d
    moved = dict(same, decay_mu=same['decay_mu'] + 1e-300)     # (bit for bit: the smallest change counts)
    assert decay_shares_bound(moved) == bool(np.array_equal(moved['decay_mu'], poly['mu']))
