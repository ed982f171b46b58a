#!/usr/bin/env python3
"""Regenerates BASELINE.md section 3 -- the AS-SHIPPED reference's rates on this host -- into
tests/golden/reference_timing.json.

Build-container only: imports the reference from /root/reference through tests/golden/make_golden.py's recipe (copy,
cythonize, two shims).  What is timed is the reference's own code: one leapfrog step of CpuLeapfrogIntegrator on
Density.logp_and_grad of a fitted PolyModel('quadratic') at d = 64 (the per-step cost behind every NUTS leaf), its parts, and
PolyModel.fit at the headline size.  The JSON records host model and core count; the GPU box never sees the reference, so
this figure is quoted as "as-shipped reference (build container)" and extrapolations from it are labelled as such.

usage: python tools/time_reference.py [--ref /root/reference] [--work /tmp/bfref] [--no-fit]"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, ROOT)


def best(f, n, rep=5):
    ts = []
    for _ in range(rep):
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        ts.append((time.perf_counter() - t0) / n)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--work', default='/tmp/bfref')
    ap.add_argument('--no-fit', action='store_true')
    a = ap.parse_args()
    import make_golden
    bf = make_golden.prepare_reference(a.ref, a.work)
    from threadpoolctl import threadpool_limits
    from bayesfast_amd.workloads import correlated_gaussian_spec
    d = 64
    _, cov = correlated_gaussian_spec(d)
    prec = np.linalg.inv(cov)
    su = bf.modules.PolyModel('quadratic', input_size=d, output_size=1, input_vars='x', output_vars='logp')
    den = bf.Density(module_list=[bf.Module(fun=lambda x: -0.5 * x @ prec @ x, input_vars='x', output_vars='logp')],
                     input_shapes=[d], input_vars='x', density_name='logp', surrogate_list=su)
    rng = np.random.default_rng(7)
    x = 1.5 * rng.normal(size=(2 * su.n_param, d))
    y = -0.5 * np.einsum('ij,jk,ik->i', x, prec, x)
    out = {'host': {'cpu': [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0],
                    'logical_cpus': os.cpu_count(), 'python': platform.python_version(), 'numpy': np.__version__},
           'workload': "d = 64, PolyModel('quadratic') = linear + quadratic, bound on, diagonal metric, one thread"}
    with threadpool_limits(1):
        if not a.no_fit:
            t0 = time.perf_counter()
            su.fit(x, y[:, None], y)
            out['polymodel_fit_s'] = time.perf_counter() - t0
            out['polymodel_fit_shape'] = [int(x.shape[0]), int(su.n_param)]
        else:
            xs = x[:su.n_param + 50]
            su.fit(xs, y[:xs.shape[0], None], y[:xs.shape[0]])
        den.use_surrogate = True
        from bayesfast.samplers.hmc_utils.integration import CpuLeapfrogIntegrator
        from bayesfast.samplers.hmc_utils.metrics import QuadMetricDiag
        integ = CpuLeapfrogIntegrator(QuadMetricDiag(np.ones(d)), lambda q: den.logp_and_grad(q, original_space=False))
        q0 = 0.3 * rng.normal(size=d)
        st = integ.compute_state(q0, rng.normal(size=d))
        out['leapfrog_step_us'] = best(lambda: integ.step(0.05, st), 300) * 1e6
        out['logp_and_grad_us'] = best(lambda: den.logp_and_grad(q0, original_space=False), 300) * 1e6
        out['polymodel_fun_and_jac_us'] = best(lambda: su._fun_and_jac(q0), 1000) * 1e6
        from bayesfast.modules._poly import _quadratic_f, _quadratic_j
        A = su.configs[1]._coef
        f, j = np.empty(1), np.empty((1, d))
        out['quadratic_kernels_us'] = best(lambda: (_quadratic_f(q0, A, f, 1, d), _quadratic_j(q0, A, j, 1, d)), 2000) * 1e6
    out['leapfrog_steps_per_sec_per_core'] = 1e6 / out['leapfrog_step_us']
    out['extrapolated_all_cores'] = {'value': out['leapfrog_steps_per_sec_per_core'] * os.cpu_count(),
                                     'note': 'per-core rate x logical CPUs of this host: an upper bound (ignores pool / dill overhead)'}
    path = os.path.join(ROOT, 'tests', 'golden', 'reference_timing.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
