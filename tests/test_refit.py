"""Refit glue: the resampler against fixtures recorded from the reference
(tests/golden/refit.npz, make_golden.py:gen_refit), the sharded selection on 2 CPU ranks (gloo), and -- marked gpu --
the device kernels behind them through the C ABI."""
import os
import warnings

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(G, 'refit.npz'))


def _cases(z):
    for i in range(int(z['n_case'])):
        w = z['c%d.weights' % i]
        yield i, z['c%d.a' % i], int(z['c%d.n' % i]), z['c%d.nodes' % i], (None if w.size == 0 else w), z['c%d.idx' % i]


def test_resampler_indices_equal_the_reference(fx):
    """SystematicResampler.run (utils/misc.py:61-108): the index pattern and the selection, exactly."""
    from bayesfast_amd import SystematicResampler
    from bayesfast_amd.utils.resample import systematic_ranks
    for i, a, n, nodes, w, idx in _cases(fx):
        r = SystematicResampler(nodes=nodes, weights=w)
        assert np.array_equal(r.run(a, n), idx), i
        assert np.array_equal(np.argsort(a, kind='stable')[systematic_ranks(a.size, n, nodes, w)], idx)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        j = SystematicResampler(require_unique=False).run(fx['ties.a'], int(fx['ties.n']))
    assert np.array_equal(fx['ties.a'][j], fx['ties.values'])  # with ties the VALUES are pinned, the order among equals is not


def test_importance_weights_host_equal_the_reference(fx):
    from bayesfast_amd.core.refit import importance_weights
    for k in (0.25, -1.):
        w, wt = importance_weights(fx['iw.logp'], fx['iw.logq'], k)
        assert np.array_equal(w, fx['iw.k%g.w' % k]) and np.array_equal(wt, fx['iw.k%g.wt' % k])


# ---- the sharded selection on CPU ranks: the device primitives are replaced by torch stand-ins, the collectives run ----
def _cpu_sort(a):
    import torch
    k = torch.where(a == 0., torch.zeros_like(a), a).contiguous().view(torch.int64)  # -0 == +0, as numpy sorts them
    neg = k < 0
    key = torch.where(neg, ~k, k | (-2**63)) ^ (-2**63)  # same order-preserving key as bf_order_key, as signed values
    order = torch.sort(key, stable=True).indices
    return key[order], order


def _cpu_count(keys, q, upper):
    import torch
    return torch.searchsorted(keys, q, right=bool(upper))


def _shard_worker(rank, ws, port, q):
    import torch
    import torch.distributed as dist
    from bayesfast_amd import parallel
    from bayesfast_amd.core.refit import select_rows_sharded
    from bayesfast_amd.utils.resample import SystematicResampler
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        rng = np.random.default_rng(5)
        n_chain, n_keep, d = 7, 300, 6           # ragged shards: 4 + 3 chains
        x = rng.normal(size=(n_chain, n_keep, d))
        lq = np.round(rng.normal(size=(n_chain, n_keep)), 2)  # rounded: plenty of ties across the ranks
        b, e = parallel.shard_range(n_chain, rank, ws)
        ranks = SystematicResampler(require_unique=False).ranks(n_chain * n_keep, 257)
        st = {}
        rows, vals = select_rows_sharded(torch.as_tensor(lq[b:e].reshape(-1)), torch.as_tensor(x[b:e].reshape(-1, d)), ranks,
                                         sort_fn=_cpu_sort, count_fn=_cpu_count, stats=st)
        # single-rank answer: stable argsort of the chain-major flattened array
        idx = np.argsort(lq.reshape(-1), kind='stable')[ranks]
        ok = np.array_equal(rows.numpy(), x.reshape(-1, d)[idx]) and np.array_equal(vals.numpy(), lq.reshape(-1)[idx])
        S, g = st['n_splitter'], st['candidates_per_rank']
        want = (S + 1) * 8 * ws + 2 * ws * S * 8 + 257 * (g + 2) * 8 * ws + 257 * (d + 1) * 8
        # a second selection with a tiny splitter set (wide candidate intervals) and one with more splitters than rows
        for ns in (3, 5000):
            r2, v2 = select_rows_sharded(torch.as_tensor(lq[b:e].reshape(-1)), torch.as_tensor(x[b:e].reshape(-1, d)), ranks,
                                         sort_fn=_cpu_sort, count_fn=_cpu_count, n_splitter=ns, n_loc_max=4 * n_keep)
            ok = ok and np.array_equal(r2.numpy(), x.reshape(-1, d)[idx]) and np.array_equal(v2.numpy(), lq.reshape(-1)[idx])
        q.put((rank, bool(ok), st['wire_bytes'], want, st['collectives']))
    finally:
        dist.destroy_process_group()


def test_sharded_selection_gloo_world2_is_exact_and_moves_only_selected_rows():
    """2 ranks, ragged shards, many ties: both ranks end with the rows a single rank would select, bit for bit, in FOUR
    collectives (splitters, counts, candidates, rows), and the bytes on the wire are those four messages -- not the samples."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    assert [r[1] for r in res] == [True, True]
    for _, _, wire, want, n_coll in res:
        assert n_coll == 4 and wire == want
        assert wire < 7 * 300 * 7 * 8 * 2                                   # (all samples would be 117 600 B per gather)


def _warmstart_worker(rank, ws, port, q):
    import torch
    import torch.distributed as dist
    from bayesfast_amd import parallel
    from bayesfast_amd.samplers.sample_trace import NTrace, TraceTuple, _get_metric
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        rng = np.random.default_rng(6)
        n_chain, n_iter, d = 5, 40, 3
        x = rng.normal(size=(n_chain, n_iter, d)) * np.array([1., 2., 0.5])
        st = rng.normal(size=(n_chain, n_iter, 11))
        b, e = parallel.shard_range(n_chain, rank, ws)
        tr = NTrace(n_chain=n_chain, n_iter=n_iter, n_warmup=10, random_generator=None)
        seed = tr.seed()
        tt = TraceTuple(tr, torch.as_tensor(x[b:e]), torch.as_tensor(st[b:e]), torch.as_tensor(x[b:e]), torch.as_tensor(st[b:e, :, 0]))
        cov = _get_metric(tt, 'full')
        ref = np.cov(x[:, 10:].reshape(-1, d), rowvar=False)
        ok = np.allclose(cov, ref, rtol=1e-12, atol=1e-14)
        try:   # a host view before gather() would be a collective behind the caller's back: it raises instead
            tt.get(flatten=True)
            ok = False
        except RuntimeError:
            pass
        tt.gather()
        ok = ok and np.array_equal(tt.get(flatten=True), x[:, 10:].reshape(-1, d))  # host view = all chains, on every rank
        q.put((rank, bool(ok), seed))
    finally:
        dist.destroy_process_group()


def test_seed_and_warm_start_are_rank_invariant_gloo_world2():
    """the trace's seed is resolved once and taken from rank 0; the warm-start metric reduces over ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    ps = [ctx.Process(target=_warmstart_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == res[1][2]


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_resampler_and_weights_equal_the_reference(fx):
    """bfhip_sort_keys behind SystematicResampler.run on a device tensor: the reference's indices exactly (no ties) /
    values exactly (ties); bfhip_importance_weights against the recorded weights."""
    import torch
    from bayesfast_amd import SystematicResampler
    from bayesfast_amd.core.refit import importance_weights, device_sort
    for i, a, n, nodes, w, idx in _cases(fx):
        r = SystematicResampler(nodes=nodes, weights=w)
        got = r.run(torch.as_tensor(a, device='cuda'), n)
        assert np.array_equal(got.cpu().numpy(), idx), i
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        j = SystematicResampler(require_unique=False).run(torch.as_tensor(fx['ties.a'], device='cuda'), int(fx['ties.n']))
    assert np.array_equal(fx['ties.a'][j.cpu().numpy()], fx['ties.values'])
    # stable order with NaN last, -0 == +0 (ties in index order), infinities at the ends
    a = np.array([3., np.nan, -np.inf, 0., -0., 3., np.inf, -1e-300, 3.])
    keys, order = device_sort(torch.as_tensor(a, device='cuda'))
    assert order.cpu().tolist() == [2, 7, 3, 4, 0, 5, 8, 6, 1]
    assert bool((keys[1:] >= keys[:-1]).all())
    for k in (0.25, -1.):
        w_, wt_ = importance_weights(torch.as_tensor(fx['iw.logp'], device='cuda'), torch.as_tensor(fx['iw.logq'], device='cuda'), k)
        np.testing.assert_allclose(w_.cpu().numpy(), fx['iw.k%g.w' % k], rtol=2e-15)
        np.testing.assert_allclose(wt_.cpu().numpy(), fx['iw.k%g.wt' % k], rtol=1e-13)


@pytest.mark.gpu
def test_select_fit_points_from_device_resident_trace_matches_host_path():
    """sample() leaves the rows on the GPU; select_fit_points(TraceTuple, ...) picks the same points as the host path on
    the materialised arrays, including the logp_cutoff supplements (recipe.py:1097-1155)."""
    import bayesfast_amd as bfa
    from bayesfast_amd.core.refit import select_fit_points
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 8
    _, cov = correlated_gaussian_spec(d)
    prec = np.linalg.inv(cov)
    rng = np.random.default_rng(1)
    su = bfa.PolyModel('quadratic', input_size=d, output_size=1)
    dens = bfa.SurrogateDensity(su)
    xf = rng.multivariate_normal(np.zeros(d), cov * 2.25, size=4 * su.n_param)
    dens.fit(xf, -0.5 * np.einsum('ij,jk,ik->i', xf, prec, xf))
    tt = bfa.sample(dens, dict(n_chain=24, n_iter=120, n_warmup=60, random_generator=3), verbose=False)

    def logp_true(x):
        return -0.5 * np.einsum('ij,jk,ik->i', x, prec, x) - 0.3 * (x[:, 0] > 0.5)

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        xa, la, na = select_fit_points(tt, None, logp_true, 200)
        xb, lb, nb = select_fit_points(tt.get(flatten=True), tt.get(return_type='logp', flatten=True), logp_true, 200)
    assert na == nb and np.array_equal(xa, xb) and np.array_equal(la, lb)
