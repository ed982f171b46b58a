"""Tempered samplers: the oracle's restatement of TCpuLeapfrogIntegrator and TNUTS against fixtures
recorded from the reference with logged random draws (tests/golden/tempered.npz, make_golden.py:gen_tempered), and --
marked gpu -- the device kernel against the oracle on shared xoshiro streams."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def fx():
    return np.load(os.path.join(G, 'tempered.npz'))


def _specs(fx):
    from specio import rebuild_spec
    from oracle import oracle as orc
    spec = rebuild_spec(fx, 't6.')
    base = orc.gaussian_base_spec(fx['t6.base_mean'], fx['t6.base_cov'])
    return spec, base, float(fx['t6.logxi'])


def test_oracle_tempered_integrator_matches_reference(fx):
    """compute_state and two steps of TCpuLeapfrogIntegrator (integration.py:132-222)."""
    from oracle import oracle as orc
    spec, base, logxi = _specs(fx)
    u0, v0 = fx['t6.lf.u0v0']
    st = orc.tempered_states(spec, base, logxi, fx['t6.lf.var'], fx['t6.lf.q0'], fx['t6.lf.p0'], u0, v0, fx['t6.lf.eps'])
    for k, tag in enumerate(('s0', 's1', 's2')):
        for f in ('q', 'p', 'u', 'v', 'weight', 'energy', 'logp'):
            np.testing.assert_allclose(st[f][k], fx['t6.lf.%s.%s' % (tag, f)], rtol=1e-11, atol=1e-11, err_msg='%s %s' % (tag, f))


@pytest.mark.parametrize('c', [0, 1])
def test_oracle_tnuts_replays_reference_trajectories(fx, c):
    """TNUTS with the reference's logged draws: tree depth / size / divergence exactly, samples, u and weights closely."""
    from oracle import oracle as orc
    spec, base, logxi = _specs(fx)
    k = 't6.tnuts%d.' % c
    n_iter, n_warmup = int(fx['t6.n_iter']), int(fx['t6.n_warmup'])
    ch = orc.Chain(fx['t6.x0'][c])
    rng = orc.make_rng('replay', normals=fx[k + 'normals'], uniforms=fx[k + 'uniforms'])
    s, st, u_last = orc.tnuts_run(spec, base, logxi, ch, rng, float(fx[k + 'u_first']), n_iter, n_warmup)
    for f in ('tree_depth', 'tree_size', 'diverging', 'warmup'):
        assert np.array_equal(st[f], fx[k + f]), f
    np.testing.assert_allclose(s, fx[k + 'samples'], rtol=1e-8, atol=1e-8)
    for f in ('u', 'weight', 'logp', 'energy', 'mean_tree_accept', 'step_size', 'step_size_bar', 'energy_change', 'max_energy_change'):
        np.testing.assert_allclose(st[f], fx[k + f], rtol=1e-7, atol=1e-8, err_msg=f)
    assert rng[0].i_normal == fx[k + 'normals'].size and rng[0].i_uniform == fx[k + 'uniforms'].size
