"""CPU checks of the group sampler kernel (bayesfast_amd/csrc/bfhip_group.h) through its host emulation.

The same header that hipcc compiles for gfx950 is compiled for the host with one fibre per lane (tests/emu): the
per-lane chain state machines, the cross-wave reductions through the emulated LDS, the barrier placement and the
MFMA tile ownership run exactly as on the device (a collective called from divergent control flow deadlocks the
emulation, a cross-wave race shows up as a wrong result), only libm and the instruction timing differ.  The oracle
shares the xoshiro256++ streams, so tree depth / size / divergence must match exactly and positions to 1e-9 on the
head of a run (float64 summation order is the only difference).  The GPU parity tests proper are tests/test_gpu_*.py."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'emu'))
G = os.path.join(HERE, 'golden')
SEED = 20240917


@pytest.fixture(scope='module')
def samp():
    return np.load(os.path.join(G, 'sampler.npz'))


def _spec(z, prefix):
    from specio import rebuild_spec
    return rebuild_spec(z, prefix)


def _emu(spec, x0, n_iter, n_warmup, sampler='NUTS', first_stream=0, split=None, seed=SEED, launch_iters=None, **kw):
    import emu
    ec = emu.EmuChains(spec, x0, seed=seed, first_stream=first_stream, step_size=kw.get('step_size', 1.))
    run_kw = {k: v for k, v in kw.items() if k in ('max_treedepth', 'max_change', 'n_int_step', 'target_accept', 'layout')}
    if split is None:
        s, st = ec.run(n_iter, sampler, n_warmup=n_warmup, launch_iters=launch_iters, **run_kw)
    else:
        s1, st1 = ec.run(split, sampler, n_warmup=n_warmup, **run_kw)
        s2, st2 = ec.run(n_iter - split, sampler, n_warmup=n_warmup, **run_kw)
        s = np.concatenate([s1, s2], 1)
        st = {k: np.concatenate([st1[k], st2[k]], 1) for k in st1}
    return s, st, ec


def _oracle(spec, x0, n_iter, n_warmup, sampler='NUTS', first_stream=0, seed=SEED, **kw):
    from oracle import oracle as orc
    out = []
    for i in range(x0.shape[0]):
        ch = orc.Chain(x0[i], **{k: v for k, v in kw.items() if k in ('step_size', 'target_accept')})
        rng = orc.make_rng('xoshiro', seed=seed, stream=first_stream + i)
        if sampler == 'NUTS':
            out.append(orc.nuts_run(spec, ch, rng, n_iter, n_warmup, max_treedepth=kw.get('max_treedepth', 10),
                                    max_change=kw.get('max_change', 1000.)) + (ch,))
        else:
            out.append(orc.hmc_run(spec, ch, rng, n_iter, n_warmup, n_int_step=kw.get('n_int_step', 32),
                                   max_change=kw.get('max_change', 1000.)) + (ch,))
    return out


def _compare_nuts(dev, orc_runs, n_exact, rtol_q=1e-5, n_head=8, tol_head=1e-9):
    s, st, ec = dev
    for i, (so, sto, ch) in enumerate(orc_runs):
        for f in ('tree_depth', 'tree_size', 'diverging', 'warmup'):
            assert np.array_equal(st[f][i][:n_exact], sto[f][:n_exact]), (i, f, st[f][i][:n_exact], sto[f][:n_exact])
        np.testing.assert_allclose(s[i][:n_head], so[:n_head], rtol=tol_head, atol=tol_head, err_msg='chain %d' % i)
        np.testing.assert_allclose(s[i][:n_exact], so[:n_exact], rtol=rtol_q, atol=rtol_q, err_msg='chain %d' % i)
        for f in ('logp', 'energy', 'mean_tree_accept', 'step_size', 'step_size_bar', 'energy_change', 'max_energy_change'):
            np.testing.assert_allclose(st[f][i][:n_head], sto[f][:n_head], rtol=1e-8, atol=1e-8, err_msg=f)
            np.testing.assert_allclose(st[f][i][:n_exact], sto[f][:n_exact], rtol=1e-4, atol=1e-4, err_msg=f)


@pytest.mark.parametrize('name,n_chain,n_iter,n_warmup', [('plain16', 5, 40, 25), ('d64', 18, 14, 9)])
def test_group_kernel_nuts_trajectories_match_oracle(samp, name, n_chain, n_iter, n_warmup):
    """W = 1 (one wave per group) and W = 4 (four waves, two groups, the second one ragged)."""
    spec = _spec(samp, name + '.')
    x0 = np.random.default_rng(3).normal(size=(n_chain, spec['d'])) * 0.5
    dev = _emu(spec, x0, n_iter, n_warmup)
    orc_runs = _oracle(spec, x0, n_iter, n_warmup)
    _compare_nuts(dev, orc_runs, n_iter)
    s, st, ec = dev
    for i, (so, sto, ch) in enumerate(orc_runs):
        np.testing.assert_allclose(ec.field('var')[i], ch.vec('var'), rtol=1e-5)
    assert int(ec.n_leapfrog[0]) == int(sum(r[1]['tree_size'].sum() for r in orc_runs))
    assert (ec.field('error') == 0).all()


def test_group_kernel_resume_shard_and_launch_cut_invariance(samp):
    """Two launches == one launch == launches of 7 iterations; a chain's result depends on its GLOBAL stream index only,
    not on its group or lane (bitwise)."""
    spec = _spec(samp, 'plain16.')
    x0 = np.random.default_rng(4).normal(size=(20, 16)) * 0.5
    s_all, st_all, ec_all = _emu(spec, x0, 30, 20)
    s_split, st_split, ec_split = _emu(spec, x0, 30, 20, split=13)
    assert np.array_equal(s_all, s_split)
    assert np.array_equal(st_all['tree_size'], st_split['tree_size'])
    s_cut, st_cut, ec_cut = _emu(spec, x0, 30, 20, launch_iters=7)
    assert np.array_equal(s_all, s_cut) and np.array_equal(ec_all.sc, ec_cut.sc) and np.array_equal(ec_all.rng, ec_cut.rng)
    for k in st_all:
        assert np.array_equal(st_all[k], st_cut[k])
    s_tail, st_tail, _ = _emu(spec, x0[7:], 30, 20, first_stream=7)
    assert np.array_equal(s_all[7:], s_tail)


def test_group_kernel_divergences_depth_cap_and_far_start(samp):
    """Divergent leaves and immediate U-turns (huge step), the depth cap, and chains that start 6 sigma out, i.e.
    outside the alpha-ellipsoid (two-pass evaluations, poly.py:480-503; weights over a huge dynamic range)."""
    spec = _spec(samp, 'div5.')
    x0 = np.repeat(samp['div5.x0'], 4, 0) + np.arange(4)[:, None] * 0.1
    kw = dict(step_size=40., max_change=50.)
    dev = _emu(spec, x0, 30, 10, **kw)
    orc_runs = _oracle(spec, x0, 30, 10, **kw)
    assert sum(r[1]['diverging'].sum() for r in orc_runs) >= 1
    _compare_nuts(dev, orc_runs, 30)
    spec = _spec(samp, 'plain16.')
    x0 = np.random.default_rng(5).normal(size=(4, 16))
    kw = dict(step_size=0.05, max_treedepth=3)
    dev = _emu(spec, x0, 12, 0, **kw)
    orc_runs = _oracle(spec, x0, 12, 0, **kw)
    assert dev[1]['tree_depth'].max() == 3 and dev[1]['tree_size'].max() == 7
    _compare_nuts(dev, orc_runs, 12)
    spec = _spec(samp, 'd64.')
    x0 = np.random.default_rng(6).normal(size=(3, 64)) * 6.
    dev = _emu(spec, x0, 8, 6)
    orc_runs = _oracle(spec, x0, 8, 6)
    _compare_nuts(dev, orc_runs, 8, tol_head=1e-8)


def _same(a, b):
    sa, sta, ea = a
    sb, stb, eb = b
    assert np.array_equal(sa, sb, equal_nan=True)
    for k in sta:
        assert np.array_equal(sta[k], stb[k], equal_nan=True), k
    assert np.array_equal(ea.sc, eb.sc) and np.array_equal(ea.vec, eb.vec) and np.array_equal(ea.rng, eb.rng)
    assert int(ea.n_leapfrog[0]) == int(eb.n_leapfrog[0])


def test_split_layout_is_bit_identical_to_the_group_layout(samp):
    """bfhip_split.h (chain_layout 3: integrator and bookkeeper waves, two per SIMD, the bookkeepers one leaf behind and the
    integrators running ahead on the assumption that the tree goes on) against bfhip_group.h on the emulator: every leaf that
    is used is computed from the same state with the same arithmetic, so samples, all statistics, the adapted state and the
    random streams are EQUAL -- with trees out of step in a ragged second group, through warm-up with metric updates, across
    launch cuts and a resumed run, with starts outside the bound (second and third passes, the extra barrier), divergent
    leaves, the depth cap, a padded dimension, and a chain whose initial energy is not finite."""
    spec = _spec(samp, 'd64.')
    x0 = np.random.default_rng(12).normal(size=(18, 64)) * 0.7           # two groups, the second one ragged
    g = _emu(spec, x0, 9, 6, layout='group')
    _same(g, _emu(spec, x0, 9, 6, layout='split'))
    _compare_nuts((g[0][:2], {k: v[:2] for k, v in g[1].items()}, g[2]), _oracle(spec, x0[:2], 9, 6), 9)   # (= the oracle's)
    far = np.random.default_rng(6).normal(size=(3, 64)) * 6.             # outside the alpha-ellipsoid; the split run resumed
    _same(_emu(spec, far, 5, 4, layout='group'), _emu(spec, far, 5, 4, layout='split', split=3))   # (two launches, cut in warm-up)
    kw = dict(step_size=30., max_change=20.)                             # divergent leaves, immediate U-turns
    a = _emu(spec, x0[:4], 6, 2, layout='group', **kw)
    assert a[1]['diverging'].sum() >= 1
    _same(a, _emu(spec, x0[:4], 6, 2, layout='split', **kw))
    kw = dict(step_size=0.02, max_treedepth=3)                           # the depth cap
    a = _emu(spec, x0[:3], 4, 0, layout='group', **kw)
    assert a[1]['tree_depth'].max() == 3
    _same(a, _emu(spec, x0[:3], 4, 0, layout='split', **kw))
    from bayesfast_amd.workloads import correlated_gaussian_spec
    spec40 = correlated_gaussian_spec(40)[0]                              # d = 40 in the 64-wide layout
    x40 = np.random.default_rng(9).normal(size=(5, 40))
    _same(_emu(spec40, x40, 6, 4, layout='group'), _emu(spec40, x40, 6, 4, layout='split'))
    spec32 = correlated_gaussian_spec(32)[0]                              # W = 2: two integrator and two bookkeeper waves
    x32 = np.random.default_rng(10).normal(size=(18, 32))
    _same(_emu(spec32, x32, 8, 5, layout='group'), _emu(spec32, x32, 8, 5, layout='split'))
    spec20 = correlated_gaussian_spec(20)[0]                              # d = 20 in the 32-wide layout, a resumed run
    x20 = np.random.default_rng(11).normal(size=(5, 20)) * 2.
    _same(_emu(spec20, x20, 6, 4, layout='group'), _emu(spec20, x20, 6, 4, layout='split', split=3))
    spec10 = correlated_gaussian_spec(10)[0]                              # W = 1: one integrator and one bookkeeper wave, d = 10 of 16
    x10 = np.random.default_rng(12).normal(size=(19, 10))
    _same(_emu(spec10, x10, 10, 6, layout='group'), _emu(spec10, x10, 10, 6, layout='split'))
    bad = x0[:3].copy()
    bad[1, 0] = np.inf                                                   # bad initial energy: error flag, chain stops
    a, b = _emu(spec, bad, 3, 2, layout='group'), _emu(spec, bad, 3, 2, layout='split')
    assert a[2].field('error')[1] == 1 and np.array_equal(a[2].field('error'), b[2].field('error'))
    assert np.array_equal(a[0][[0, 2]], b[0][[0, 2]])


def test_group_kernel_hmc_matches_oracle(samp):
    spec = _spec(samp, 'plain16.')
    x0 = np.random.default_rng(7).normal(size=(6, 16)) * 0.5
    s, st, ec = _emu(spec, x0, 30, 20, sampler='HMC', n_int_step=8)
    orc_runs = _oracle(spec, x0, 30, 20, sampler='HMC', n_int_step=8)
    for i, (so, sto, ch) in enumerate(orc_runs):
        for f in ('accepted', 'diverging', 'n_int_step'):
            assert np.array_equal(st[f][i], sto[f]), (i, f)
        np.testing.assert_allclose(s[i][:8], so[:8], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(s[i], so, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(st['accept_stat'][i], sto['accept_stat'], rtol=1e-3, atol=1e-3)
        for f in ('logp', 'energy', 'energy_change', 'step_size'):
            np.testing.assert_allclose(st[f][i][:8], sto[f][:8], rtol=1e-8, atol=1e-8, err_msg=f)
    assert int(ec.n_leapfrog[0]) == 6 * 30 * 8


@pytest.mark.parametrize('decay,bounds,d', [(True, False, 64), (False, True, 30), (True, True, 48)])
def test_group_kernel_feature_sets_match_oracle(samp, decay, bounds, d):
    """Decay penalty (density.py:740-746) and / or constraint transform with all four kinds of bounds
    (density.py:92-140,747-750): template parameter FS = 3, 5, 7."""
    from bayesfast_amd.workloads import correlated_gaussian_spec
    spec = dict(_spec(samp, 'd64.')) if d == 64 else dict(correlated_gaussian_spec(d)[0])
    if decay:
        spec.update(use_decay=True, decay_mu=spec['poly']['mu'], decay_hess=spec['poly']['hess'],
                    decay_alpha2=float(spec['poly']['alpha'])**2 * 0.6, decay_gamma=0.1)
    if bounds:
        lo = np.full(d, -9.) + np.arange(d) * 0.01
        hb = np.array(([[1, 1], [1, 0], [0, 1], [0, 0]] * d)[:d], dtype=np.uint8)
        spec.update(ranges=np.stack([lo, lo + 18.], 1), hard_bounds=hb)
    x0 = np.random.default_rng(8).normal(size=(5, d)) * 0.3
    dev = _emu(spec, x0, 12, 8)
    orc_runs = _oracle(spec, x0, 12, 8)
    _compare_nuts(dev, orc_runs, 12, n_head=6, tol_head=1e-8)
    # HMC through the same feature set
    s, st, ec = _emu(spec, x0[:3], 8, 5, sampler='HMC', n_int_step=5)
    for i, (so, sto, ch) in enumerate(_oracle(spec, x0[:3], 8, 5, sampler='HMC', n_int_step=5)):
        assert np.array_equal(st['accepted'][i], sto['accepted'])
        np.testing.assert_allclose(s[i][:5], so[:5], rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize('case', ['scales_1e3', 'scales_small_alpha', 'scales_decay', 'scales_1e5_split'])
def test_group_kernel_folded_input_scales_weighted_proof_and_linearity_match_oracle(case):
    """The upload paths of round 4 on the CPU (device.py: density_desc_from_spec), through the group and split kernels' host
    emulation: Surrogate.input_scales folded into the coefficients, the bound's centre and Hessian (a Hessian seen through
    per-dimension scales spread over three to five decades: the weighted bound proof), the bound shrunk so that leaves sit outside
    it (S x_0 by linearity, bfhip_oob.h), and the decay term on top.  The oracle evaluates the unfolded density."""
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 24
    spec = dict(correlated_gaussian_spec(d, fit_scale=1.5)[0])
    rng = np.random.default_rng(31)
    spread = 5. if case == 'scales_1e5_split' else 3.
    diff = 10.**rng.uniform(-spread / 2., spread / 2., size=d)
    lo = rng.normal(size=d) * diff * 0.5
    spec.update(su_lo=lo, su_diff=diff)
    po = dict(spec['poly'])
    if case == 'scales_small_alpha':
        po['alpha'] = 0.35 * float(po['alpha'])   # most leaves outside the bound
    spec['poly'] = po
    if case == 'scales_decay':
        # (the decay term lives in the original space: its ellipsoid is the bound's seen through the scales)
        mu_o = lo + diff * np.asarray(po['mu'])
        hess_o = np.asarray(po['hess']) / np.outer(diff, diff)
        spec.update(use_decay=True, decay_mu=mu_o + 0.05 * diff, decay_hess=hess_o, decay_alpha2=(0.8 * float(po['alpha']))**2, decay_gamma=0.1)
    x0 = lo + diff * (rng.normal(size=(4, d)) * 0.6)
    kw = dict(layout='split') if case == 'scales_1e5_split' else {}
    step = 0.5 * float(diff.min())   # (identity metric at the start: the step the narrowest direction takes)
    dev = _emu(spec, x0, 9, 6, step_size=step, **kw)
    orc_runs = _oracle(spec, x0, 9, 6, step_size=step)
    s, st, ec = dev
    for i, (so, sto, ch) in enumerate(orc_runs):
        for f in ('tree_depth', 'tree_size', 'diverging'):
            assert np.array_equal(st[f][i], sto[f]), (case, i, f, st[f][i], sto[f])
        np.testing.assert_allclose((s[i][:6] - lo) / diff, (so[:6] - lo) / diff, rtol=1e-7, atol=1e-7, err_msg='chain %d' % i)
        np.testing.assert_allclose(st['logp'][i][:6], sto['logp'][:6], rtol=1e-8, atol=1e-7)
    if case == 'scales_small_alpha':
        from oracle import oracle as orc
        assert dev[1]['tree_size'].sum() > 0


def test_group_kernel_bad_initial_energy_sets_error_flag(samp):
    """Non-finite initial energy is an error, not a divergence (base_hmc.py:72-76): error code 1, nothing sampled."""
    spec = _spec(samp, 'plain16.')
    x0 = np.zeros((3, 16))
    x0[1, 0] = np.inf
    s, st, ec = _emu(spec, x0, 3, 1)
    assert list(ec.field('error')) == [0., 1., 0.]
    assert list(ec.field('i_iter')) == [3., 0., 3.]


@pytest.mark.parametrize('d', [1, 3, 17, 33])
def test_group_kernel_odd_shapes(d):
    """Dimensions that are not multiples of the 16-wide tiles; 1 and 17 chains (ragged groups)."""
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from oracle import oracle as orc
    spec, _ = correlated_gaussian_spec(d)
    for C in (1, 17):
        x0 = np.random.default_rng(d * 1000 + C).normal(size=(C, d)) * 0.7
        s, st, ec = _emu(spec, x0, 10, 7, seed=9)
        for i in sorted(set((0, C - 1))):
            so, sto = orc.nuts_run(spec, orc.Chain(x0[i]), orc.make_rng('xoshiro', seed=9, stream=i), 10, 7)
            assert np.array_equal(st['tree_size'][i], sto['tree_size']), (d, C, i)
            np.testing.assert_allclose(s[i, :5], so[:5], rtol=1e-8, atol=1e-8)


def test_bound_proof_eigenvalue_bound_is_proven_and_tight():
    """bf_bound_lam_max (bfhip_pack.h): the upper bound of lam_max((H + H^T) / 2) that the kernels' bound proof multiplies
    |x - mu|^2 with.  It must never be below the true largest eigenvalue (the proof would be wrong) and should not be far
    above it (the proof would rarely hold); NaN input must not produce a usable bound."""
    import ctypes as C
    import emu
    f = emu.lib().bfemu_bound_lam_max
    f.restype = C.c_double
    rng = np.random.default_rng(3)
    for d in (1, 2, 7, 64, 128):
        for kind in ('spd', 'ill', 'nonsym', 'indefinite', 'diag'):
            a = rng.normal(size=(d, d))
            if kind == 'spd':
                h = a @ a.T / d + 0.1 * np.eye(d)
            elif kind == 'ill':
                q, _ = np.linalg.qr(a)
                h = (q * np.logspace(-8, 2, d)) @ q.T
            elif kind == 'nonsym':
                h = a @ a.T / d + 0.3 * (a - a.T)
            elif kind == 'indefinite':
                h = 0.5 * (a + a.T)
            else:
                h = np.diag(rng.uniform(0.1, 5., d))
            h = np.ascontiguousarray(h)
            lam = float(np.linalg.eigvalsh(0.5 * (h + h.T))[-1])
            got = f(h.ctypes.data_as(C.c_void_p), C.c_int(d))
            assert got >= lam, (d, kind, got, lam)
            scale = max(abs(lam), float(np.abs(np.linalg.eigvalsh(0.5 * (h + h.T))).max()))
            assert got <= 1.2 * scale + 1e-300, (d, kind, got, lam)
    # zero row sums: the all-ones vector is in the null space (a symmetric start vector would stall the power iteration and,
    # before the search was floored and capped, hang the upload); also the zero and a negative definite matrix
    lap = np.diag(np.full(8, 2.)) - np.eye(8, k=1) - np.eye(8, k=-1) - np.eye(8, k=7) - np.eye(8, k=-7)
    for h in (np.array([[1., -1.], [-1., 1.]]), lap, np.zeros((3, 3)), -np.eye(5)):
        h = np.ascontiguousarray(h)
        lam = float(np.linalg.eigvalsh(h)[-1])
        got = f(h.ctypes.data_as(C.c_void_p), C.c_int(h.shape[0]))
        assert got >= lam and got <= 1.2 * np.abs(np.linalg.eigvalsh(h)).max() + 1e-300, (h, got, lam)
    bad = np.full((4, 4), np.nan)
    got = f(bad.ctypes.data_as(C.c_void_p), C.c_int(4))
    assert not (got < np.inf)  # NaN or inf: `lam_max * r2 < alpha^2` is then false and the proof never claims anything
