#!/usr/bin/env python3
"""What the host of the GPU box gives this process, and what the CPU port makes of it: topology, affinity, cgroup
quota, then the tuned port's leapfrog rate at 1 .. all threads with and without thread pinning (each setting in a child
process, because OMP_* are read when libgomp loads).  Writes one JSON line per measurement.

  python3 tools/cpu_probe.py            # the sweep
  python3 tools/cpu_probe.py --child N  # one measurement with N threads (env decides the pinning)
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host_facts():
    from bayesfast_amd.utils.hostinfo import host_cpu_facts
    return host_cpu_facts()


def child(n_thr, chains_per_thread=4, seconds=4.):
    from oracle import oracle as orc
    from bayesfast_amd.workloads import correlated_gaussian_spec
    spec, _ = correlated_gaussian_spec(64)
    n_chain = chains_per_thread * n_thr
    x0 = np.random.default_rng(1).normal(size=(n_chain, 64))
    cs = orc.ChainSet(spec, x0, 1, tuned=True, step_size=0.35, adapt_step_size=False, adapt_metric=False)
    cs.run(20, 0, n_threads=n_thr)   # page in
    it = 50
    nl = dt = 0.
    while dt < seconds:
        t0 = time.perf_counter()
        _, _, k = cs.run(it, 0, n_threads=n_thr)
        t1 = time.perf_counter() - t0
        dt += t1
        nl += k
        if t1 < 1.:
            it = min(it * 2, 3200)
    cs.close()
    print(json.dumps({'threads': n_thr, 'chains': n_chain, 'rate': nl / dt, 'rate_per_thread': nl / dt / n_thr,
                      'us_per_leapfrog_per_thread': 1e6 * dt * n_thr / nl, 'slice_iters': it,
                      'OMP_PROC_BIND': os.environ.get('OMP_PROC_BIND'), 'OMP_PLACES': os.environ.get('OMP_PLACES')}))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(int(sys.argv[2]))
        return
    facts = host_facts()
    print(json.dumps({'host': facts}))
    sys.stdout.flush()
    n_max = facts['usable_threads']
    ns = sorted({1, 2, 4, 8, 16, 32, 64, facts['usable_cores'], n_max} & set(range(1, n_max + 1)))
    for bind in (None, ('close', 'cores'), ('spread', 'threads')):
        for n in ns:
            env = dict(os.environ)
            env.pop('OMP_PROC_BIND', None)
            env.pop('OMP_PLACES', None)
            if bind:
                env['OMP_PROC_BIND'], env['OMP_PLACES'] = bind
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(n)], env=env, capture_output=True, text=True)
            print(r.stdout.strip() or json.dumps({'threads': n, 'error': r.stderr[-400:]}))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
