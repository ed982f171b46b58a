"""GPU parity (through the C ABI): batched surrogate logp+grad and batched leapfrog vs the CPU oracle and
vs the golden vectors the reference produced.  Tolerance: float64, relative 1e-11 (summation order only)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def ctx():
    from bayesfast_amd.device import get_context
    return get_context(0)


def _spec(z, prefix):
    from specio import rebuild_spec
    return rebuild_spec(z, prefix)


@pytest.mark.parametrize('case', ['plain', 'decay', 'scales', 'su', 'full', 'd64'])
def test_logp_grad_golden(ctx, case):
    """Density.logp_and_grad fixtures of the reference (core/density.py:724-754), both spaces."""
    from bayesfast_amd.device import DeviceDensity
    z = np.load(os.path.join(G, 'density.npz'))
    dd = DeviceDensity(_spec(z, case + '.'), ctx)
    lp, g = dd.logp_and_grad(z[case + '.x_trans'])
    np.testing.assert_allclose(lp.cpu().numpy(), z[case + '.logp_trans'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(g.cpu().numpy(), z[case + '.grad_trans'], rtol=1e-10, atol=1e-10)
    if case + '.x_orig' in z.files:
        lp, g = dd.logp_and_grad(z[case + '.x_orig'], original_space=True)
        np.testing.assert_allclose(lp.cpu().numpy(), z[case + '.logp_orig'], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(g.cpu().numpy(), z[case + '.grad_orig'], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize('case,n,scale', [('full', 1000, 1.0), ('d64', 4096, 1.0), ('d64', 333, 2.5), ('plain', 17, 3.0)])
def test_logp_grad_vs_oracle(ctx, case, n, scale):
    """Seeded random batches, ragged sizes, points inside and outside the bound, vs the C oracle."""
    from bayesfast_amd.device import DeviceDensity
    from oracle import oracle as orc
    z = np.load(os.path.join(G, 'density.npz'))
    spec = _spec(z, case + '.')
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n, spec['d'])) * scale
    lp0, g0 = orc.logp_and_grad(spec, x)
    lp, g = DeviceDensity(spec, ctx).logp_and_grad(x)
    lp, g = lp.cpu().numpy(), g.cpu().numpy()
    if spec['poly']['use_bound']:
        beta = np.sqrt(np.einsum('ij,jk,ik->i', x - spec['poly']['mu'], spec['poly']['hess'], x - spec['poly']['mu'])) \
            if spec.get('su_lo') is None and spec.get('ranges') is None else None
        if beta is not None and scale > 1.:
            assert (beta > spec['poly']['alpha']).any(), 'test should exercise the extrapolation branch'
    np.testing.assert_allclose(lp, lp0, rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(g, g0, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize('d,n,scale', [(128, 4099, 1.0), (128, 300, 1.75), (100, 77, 1.8), (128, 5, 1.0)])
def test_logp_grad_d128_cooperative_kernel_vs_oracle(ctx, d, n, scale):
    """d > 64 on the common surrogate: bf_logp_grad_coop128_kernel (eight waves per 16 points, one row tile each) against
    the C oracle -- points inside and outside the bound in the same tile, a padded dimension, ragged and tiny batches."""
    from bayesfast_amd.device import DeviceDensity
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from oracle import oracle as orc
    spec, _ = correlated_gaussian_spec(d)
    x = np.random.default_rng(8).normal(size=(n, d)) * scale
    lp0, g0 = orc.logp_and_grad(spec, x)
    po = spec['poly']
    beta = np.sqrt(np.einsum('ij,jk,ik->i', x - po['mu'], po['hess'], x - po['mu']))
    if scale > 1.:
        assert (beta > po['alpha']).any() and (beta < po['alpha']).any(), 'both branches in the batch'
    lp, g = DeviceDensity(spec, ctx).logp_and_grad(x)
    np.testing.assert_allclose(lp.cpu().numpy(), lp0, rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(g.cpu().numpy(), g0, rtol=1e-10, atol=1e-9)


def test_logp_grad_empty_and_errors(ctx):
    import torch
    from bayesfast_amd.device import DeviceDensity
    z = np.load(os.path.join(G, 'density.npz'))
    dd = DeviceDensity(_spec(z, 'plain.'), ctx)
    lp, g = dd.logp_and_grad(np.zeros((0, 5)))
    assert lp.shape == (0,) and g.shape == (0, 5)
    lp1, g1 = dd.logp_and_grad(np.zeros(5))
    assert lp1.dim() == 0 and g1.shape == (5,)
    assert torch.isfinite(lp1)


@pytest.mark.parametrize('name', ['full5', 'plain16', 'd64'])
def test_leapfrog_golden_and_oracle(ctx, name):
    """CpuLeapfrogIntegrator._step fixtures (integration.py:68-95), then a random batch vs the oracle."""
    import torch
    from bayesfast_amd.device import DeviceDensity
    from oracle import oracle as orc
    samp = np.load(os.path.join(G, 'sampler.npz'))
    spec = _spec(samp, name + '.')
    dd = DeviceDensity(spec, ctx)
    var, eps = samp[name + '.lf.var'], samp[name + '.lf.eps']
    T = lambda a: ctx.tensor(np.atleast_2d(a).copy(), torch.float64)
    for a, b, e in (('s0', 's1', eps[0]), ('s1', 's2', eps[1])):
        q, p, g = (T(samp['%s.lf.%s.%s' % (name, a, f)]) for f in ('q', 'p', 'q_grad'))
        v = ctx.empty(q.shape)
        logp, energy = dd.leapfrog(ctx.tensor(np.array([e])), T(var), q, p, g, velocity=v)
        for f, t in (('q', q), ('p', p), ('velocity', v), ('q_grad', g)):
            np.testing.assert_allclose(t.cpu().numpy()[0], samp['%s.lf.%s.%s' % (name, b, f)], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(logp.item(), samp['%s.lf.%s.logp' % (name, b)], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(energy.item(), samp['%s.lf.%s.energy' % (name, b)], rtol=1e-11, atol=1e-11)
    # batch of 77 chains with per-chain eps and var
    rng = np.random.default_rng(8)
    n, d = 77, spec['d']
    q, p = rng.normal(size=(n, d)) * 0.4, rng.normal(size=(n, d))
    var_b, eps_b = rng.uniform(0.5, 2., size=(n, d)), rng.uniform(-0.2, 0.2, size=n)
    _, g = orc.logp_and_grad(spec, q)
    tq, tp, tg = T(q), T(p), T(g)
    logp, energy = dd.leapfrog(ctx.tensor(eps_b), T(var_b), tq, tp, tg)
    for i in range(0, n, 7):
        r = orc.leapfrog(spec, var_b[i], eps_b[i], q[i], p[i], g[i])
        np.testing.assert_allclose(tq[i].cpu().numpy(), r['q'], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(tp[i].cpu().numpy(), r['p'], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(tg[i].cpu().numpy(), r['grad'], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(energy[i].item(), r['energy'], rtol=1e-11, atol=1e-10)


@pytest.mark.parametrize('d,n,scale', [(128, 333, 1.0), (100, 50, 1.9)])
def test_leapfrog_d128_cooperative_kernel_vs_oracle(ctx, d, n, scale):
    """CpuLeapfrogIntegrator._step at d > 64 on the common surrogate (bf_leapfrog_coop128_kernel) against the C oracle, end
    points inside and outside the bound."""
    import torch
    from bayesfast_amd.device import DeviceDensity, _ptr
    from bayesfast_amd.workloads import correlated_gaussian_spec
    from bayesfast_amd import _lib
    from oracle import oracle as orc
    spec, _ = correlated_gaussian_spec(d)
    rng = np.random.default_rng(6)
    q = rng.normal(size=(n, d)) * scale
    p = rng.normal(size=(n, d))
    var = rng.uniform(0.5, 2., size=(n, d))
    eps = rng.uniform(0.05, 0.15, size=n)
    _, g_start = orc.logp_and_grad(spec, q)
    ref = [orc.leapfrog(spec, var[i], eps[i], q[i], p[i], g_start[i]) for i in range(n)]
    dd = DeviceDensity(spec, ctx)
    qt, pt, gt = ctx.tensor(q, torch.float64), ctx.tensor(p, torch.float64), ctx.tensor(g_start, torch.float64)
    lpt, et, vt = ctx.empty((n,)), ctx.empty((n,)), ctx.empty((n, d))
    epst, vart = ctx.tensor(eps, torch.float64), ctx.tensor(var, torch.float64)
    _lib.check(ctx._lib.bfhip_leapfrog(ctx.handle, n, _ptr(epst), _ptr(vart), _ptr(qt), _ptr(pt), _ptr(gt), _ptr(lpt), _ptr(et), _ptr(vt)))
    for name, dev in (('q', qt), ('p', pt), ('grad', gt)):
        np.testing.assert_allclose(dev.cpu().numpy(), np.stack([r[name] for r in ref]), rtol=1e-10, atol=1e-9, err_msg=name)
    np.testing.assert_allclose(vt.cpu().numpy(), var * pt.cpu().numpy(), rtol=1e-14, atol=0)   # velocity = var p (metrics.py:88-91)
    np.testing.assert_allclose(et.cpu().numpy(), np.array([r['energy'] for r in ref]), rtol=1e-11, atol=1e-9)
    po = spec['poly']
    beta = np.sqrt(np.einsum('ij,jk,ik->i', qt.cpu().numpy() - po['mu'], po['hess'], qt.cpu().numpy() - po['mu']))
    if scale > 1.:
        assert (beta > po['alpha']).any() and (beta < po['alpha']).any(), 'both branches in the batch'


def test_constraint_transforms_golden(ctx):
    """from_original / to_original and their first and second derivatives vs the reference fixture
    (transforms/_constraint.pyx:19-215), through SurrogateDensity; out-of-bound inputs raise ValueError."""
    from bayesfast_amd import PolyModel, SurrogateDensity
    z = np.load(os.path.join(G, 'constraint.npz'))
    n = z['ranges'].shape[0]
    den = SurrogateDensity(PolyModel('quadratic', input_size=n, output_size=1), input_scales=z['ranges'],
                           hard_bounds=z['hard_bounds'])
    for nm, f in (('f', den.to_original), ('j', den.to_original_grad), ('jj', den.to_original_grad2)):
        np.testing.assert_allclose(f(z['x_trans']), z['to_' + nm], rtol=1e-14, atol=0)
    xo = z['to_f']
    for nm, f in (('f', den.from_original), ('j', den.from_original_grad), ('jj', den.from_original_grad2)):
        np.testing.assert_allclose(f(xo), z['from_' + nm], rtol=1e-12, atol=0)
    bad = xo[0].copy()
    bad[0] = z['ranges'][0, 1] + 1.
    with pytest.raises(ValueError):
        den.from_original(bad)
    plain = SurrogateDensity(PolyModel('quadratic', input_size=n, output_size=1))
    assert np.array_equal(plain.to_original(z['x_trans']), z['x_trans'])
    assert np.array_equal(plain.to_original_grad(z['x_trans']), np.ones_like(z['x_trans']))


def test_default_context_follows_torchs_current_stream():
    """get_context() re-binds the process-wide context to torch's current stream (ordered after the work queued on the
    previous one), so host-side torch ops and the library's launches share a stream inside ``torch.cuda.stream`` blocks."""
    import torch
    from bayesfast_amd.device import get_context, DeviceDensity
    from bayesfast_amd.workloads import correlated_gaussian_spec
    ctx = get_context(0)
    s0 = ctx.stream
    side = torch.cuda.Stream()
    spec, _ = correlated_gaussian_spec(16)
    x = np.random.default_rng(0).normal(size=(4096, 16))
    lp0, g0 = DeviceDensity(spec, ctx).logp_and_grad(ctx.tensor(x))
    with torch.cuda.stream(side):
        c2 = get_context(0)
        assert c2 is ctx and ctx.stream.cuda_stream == side.cuda_stream
        xt = ctx.tensor(x)  # an H2D copy on the side stream, then a launch that reads it
        lp1, g1 = DeviceDensity(spec, ctx).logp_and_grad(xt)
    side.synchronize()
    assert get_context(0).stream.cuda_stream == torch.cuda.current_stream().cuda_stream == s0.cuda_stream
    assert torch.equal(lp0, lp1) and torch.equal(g0, g1)
