"""Helper of tests/test_gpu_fit_sample.py::test_sample_two_ranks_equals_one_rank: runs bfa.sample() twice on a fixed density
(under torch.distributed.run with gloo when WORLD_SIZE > 1; all ranks share the box's GPU) and lets rank 0 save the result."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bayesfast_amd as bfa  # noqa: E402
from bayesfast_amd.samplers import _get_step_size, _get_metric  # noqa: E402


def main(out):
    ws = int(os.environ.get('WORLD_SIZE', 1))
    if ws > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo')
    d = 6
    rng = np.random.default_rng(3)
    Pm = np.eye(d) + 0.3 * rng.normal(size=(d, d)) / np.sqrt(d)
    Pm = Pm @ Pm.T
    if os.environ.get('BF_TEST_PIPELINE'):   # the pipeline density (multi-output surrogate + chi-square + prior behind the box transform)
        from bayesfast_amd.workloads import des_like_pipeline
        w = des_like_pipeline(d=d, m=24, n_nonlinear=3, n_prior=3, seed=5)
        su = bfa.PolyModel([bfa.PolyConfig('linear'), bfa.PolyConfig('quadratic', input_mask=w['nonlinear'])], input_size=d, output_size=24,
                           input_scales=w['para_range'])
        den = bfa.Chi2PipelineDensity(su, w['data'], prec_diag=np.ones(24), logp0=w['norm'], prior_mu=w['prior_mu'], prior_prec=w['prior_prec'],
                                      prior_c0=w['prior_c0'], input_scales=w['para_range'], hard_bounds=True)
        lo, hi = w['para_range'][:, 0], w['para_range'][:, 1]
        u0 = (w['x_true'] - lo) / (hi - lo)
        xf = lo + (hi - lo) * np.clip(u0 + 0.1 * rng.normal(size=(5 * su.n_param, d)), 0.02, 0.98)
        den.fit(xf, w['logp'](xf), y=w['model'](xf))
    else:
        den = bfa.SurrogateDensity(bfa.PolyModel('quadratic', input_size=d, output_size=1))
        xf = rng.normal(size=(120, d)) * 1.5
        den.fit(xf, -0.5 * np.einsum('ij,jk,ik->i', xf, Pm, xf))
    # default trace: no seed given by the caller (rank 0's entropy is what every rank must use), then a second round
    # warm-started from the first (_get_step_size / _get_metric reduce over all ranks)
    seed = int(os.environ['BF_TEST_SEED'])
    # BF_TEST_IPL: iterations per launch (several launches, each choosing its layout under 'auto' from the launches before)
    ipl = int(os.environ.get('BF_TEST_IPL', 0)) or None
    tt = bfa.sample(den, {'n_chain': 22, 'n_iter': 60, 'n_warmup': 40, 'random_generator': seed}, verbose=False,
                    iters_per_launch=ipl)
    step = _get_step_size(tt)
    metric = _get_metric(tt, 'diag')
    tt2 = bfa.sample(den, {'n_chain': 22, 'n_iter': 30, 'n_warmup': 10, 'random_generator': seed + 1, 'step_size': step,
                           'metric': metric}, verbose=False, iters_per_launch=ipl)
    tt.gather()   # collectives, on every rank (a host view before them raises when there is more than one rank)
    tt2.gather()
    res = dict(s=np.asarray(tt.samples), ts=np.stack([np.asarray(tt[i].stats._tree_size) for i in range(22)]),
               step=np.asarray(step), metric=np.asarray(metric), s2=np.asarray(tt2.samples),
               logp=np.asarray(tt.get(include_warmup=True, return_type='logp')))
    if int(os.environ.get('RANK', 0)) == 0:
        np.savez(out, **res)
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
