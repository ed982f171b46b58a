"""bench.py's CPU baselines: the repository's C restatement of the reference path (oracle/, kind "port") with its TUNED density
evaluation (oracle/bf_cpu_tuned.c), timed on the host cores the process may use, in a child process that never touches the GPU.

The oracle is test / measurement infrastructure: it is loaded here only to be timed beside the device path, never to compute
anything the bench reports as the device's."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


_HOST = None


def host_facts():
    """What the host gives this process, taken ONCE: libgomp binds the initial thread to its place when the first parallel region
    runs (OMP_PROC_BIND=close, OMP_PLACES=cores), after which the process's own affinity mask reads as one core -- the second
    round of the headline's baseline ran on 2 threads that way."""
    global _HOST
    if _HOST is None:
        from bayesfast_amd.utils.hostinfo import host_cpu_facts
        h = host_cpu_facts()
        h.pop('one_cpu_per_core', None)
        _HOST = h
    return dict(_HOST)


def _timed_slices(cs, n_warm_iter, n_thr, target_seconds, first_slice=50):
    """Leapfrogs and seconds of post-warm-up slices of the chain set, each slice long enough (>= ~1 s) for the parallel
    region's start-up and the output arrays' page faults not to count."""
    nl, dt, n_it, it = 0, 0., 0, first_slice
    while dt < target_seconds:
        t0 = time.perf_counter()
        _, _, k = cs.run(it, n_warm_iter, n_threads=n_thr)
        t1 = time.perf_counter() - t0
        dt += t1
        nl += k
        n_it += it
        if t1 < 1.:
            it = min(2 * it, 4000)
    return nl, dt, n_it


def _omp_env():
    return {k: os.environ.get(k) for k in ('OMP_PROC_BIND', 'OMP_PLACES', 'OMP_NUM_THREADS')}


def adapted_rate_here(spec, d, n_warm_iter, seed, target_seconds=12.):
    """Leapfrog steps/sec of the CPU port on the host cores THIS PROCESS MAY USE, post-warm-up (its own adaptation, untimed), on a
    bounded sample: 4 chains per thread.  /proc/cpuinfo lists every CPU of the machine; the affinity mask and the cgroup's cpu.max
    say what the process gets.  `threads` = min(CPUs in the affinity mask, cgroup quota), one chain per thread, threads bound to
    cores; the 1-thread rate of the same code is measured beside it."""
    from oracle import oracle as orc  # the checker, timed as a baseline only
    host = host_facts()
    n_thr = max(1, min(host['usable_threads'], orc.max_threads()))
    x1 = np.random.default_rng(seed).normal(size=(4, d))
    c1 = orc.ChainSet(spec, x1, seed, tuned=True)
    c1.run(n_warm_iter, n_warm_iter, n_threads=1)
    nl1, dt1, _ = _timed_slices(c1, n_warm_iter, 1, min(3., target_seconds / 4))
    c1.close()
    n_chain = 4 * n_thr
    x0 = np.random.default_rng(seed).normal(size=(n_chain, d))
    cs = orc.ChainSet(spec, x0, seed, tuned=True)
    cs.run(n_warm_iter, n_warm_iter, n_threads=n_thr)  # untimed adaptation, same as the GPU path
    nl, dt, n_it = _timed_slices(cs, n_warm_iter, n_thr, target_seconds)
    tuned = cs.tuned
    cs.close()
    one = nl1 / dt1
    return {'value': nl / dt, 'unit': 'leapfrog steps/sec', 'cores': host['usable_cores'], 'threads': n_thr, 'kind': 'port',
            'one_thread_value': one, 'per_thread_efficiency': (nl / dt) / (n_thr * one), 'host': host, 'omp': _omp_env(),
            'tuned_evaluation': bool(tuned),
            'sample': '%d chains x %d post-warm-up NUTS iterations (%d leapfrogs in %.1f s) of the same %d-d workload, '
                      'one chain per OpenMP thread on the %d CPUs the process may use (affinity %d, cgroup quota %s), %s density '
                      'evaluation, %s' % (n_chain, n_it, nl, dt, d, n_thr, host['affinity_cpus'], host['cgroup_cpu_quota'],
                                          'tuned (oracle/bf_cpu_tuned.c: dense symmetric matvecs, AVX2+FMA)' if tuned else 'statement-by-statement',
                                          host['model'])}


def fixed_rate_here(spec, x_start, step_size, var, seed, target_accept, target_seconds):
    """The CPU port on the same density from where the device chains are: the oracle's NUTS driver with the TUNED evaluation, one
    chain per OpenMP thread, bounded.  The chains start at the device chains' post-adaptation positions, with the device's adapted
    step size and diagonal metric (means over the chains) held fixed -- the CPU pays for sampling, not for a second adaptation
    (minutes on the deep-tree configs)."""
    from oracle import oracle as orc
    host = host_facts()
    n_thr = max(1, min(host['usable_threads'], orc.max_threads()))
    # (eight chains per thread, dealt dynamically: chains whose trees differ -- config 3's first round has chains at the depth limit
    # beside 7-leaf ones -- would leave threads idle at one chain each, which the 4096-chain workload on the same cores would not)
    n_chain = min(8 * n_thr, x_start.shape[0])
    cs = orc.ChainSet(spec, x_start[:n_chain], seed, tuned=True, step_size=step_size, metric=var, adapt_step_size=False,
                      adapt_metric=False, target_accept=target_accept)
    nl, dt, n_it, slice_it = 0, 0., 0, 2
    while dt < target_seconds:
        t0 = time.perf_counter()
        _, _, k = cs.run(slice_it, 0, n_threads=n_thr)
        t1 = time.perf_counter() - t0
        if n_it:   # (the first slice pays the page faults of the threads' stacks: untimed)
            dt += t1
            nl += k
        n_it += slice_it
        if t1 < 0.3:
            slice_it = min(4 * slice_it, 200)
    tuned = cs.tuned
    cs.close()
    return {'value': nl / dt, 'unit': 'leapfrog steps/sec', 'cores': host['usable_cores'], 'threads': n_thr, 'kind': 'port',
            'omp': _omp_env(), 'tuned_evaluation': bool(tuned), 'host': host,
            'sample': '%d chains x %d NUTS iterations (%d leapfrogs in %.1f s) of the same density from the device chains\' '
                      'post-adaptation positions, with their adapted step size and diagonal metric (chain means) held fixed; chains '
                      'dealt dynamically over the OpenMP threads; %s density evaluation' % (
                          n_chain, n_it, nl, dt, 'tuned (oracle/bf_cpu_tuned.c)' if tuned else 'statement-by-statement')}


def two_round_rate_here(rounds, seed, target_seconds):
    """The headline workload's CPU baseline: both rounds of config 3 (`rounds`: a list of dicts with spec, x_start, step_size, var,
    target_accept, leapfrogs -- the device's leapfrog count of that round), each timed as in fixed_rate_here, combined the way the
    device's value is: all leapfrogs over the time the CPU port would need for them at the rates it showed."""
    parts = [fixed_rate_here(r['spec'], r['x_start'], r['step_size'], r['var'], seed + i, r['target_accept'],
                             target_seconds / len(rounds)) for i, r in enumerate(rounds)]
    lf = [float(r['leapfrogs']) for r in rounds]
    value = sum(lf) / sum(l / p['value'] for l, p in zip(lf, parts))
    out = dict(parts[0])
    out.update({'value': value, 'rounds': [{'value': p['value'], 'sample': p['sample'], 'tuned_evaluation': p['tuned_evaluation'],
                                            'share_of_the_workloads_leapfrogs': l / sum(lf)} for l, p in zip(lf, parts)],
                'tuned_evaluation': all(p['tuned_evaluation'] for p in parts),
                'sample': 'both rounds of the workload (round 0: %s || round 1: %s), combined as the device value is: all leapfrogs of the '
                          'timed launches over the time the CPU port needs for them at these rates; tuned density evaluation '
                          '(oracle/bf_cpu_tuned.c: bound, decay and the extrapolation outside the bound by linearity)' % (
                              parts[0]['sample'], parts[-1]['sample'])})
    return out


def _child(job):
    """The CPU baseline runs in a CHILD process that never touches the GPU: libgomp reads its thread placement when it loads,
    so OMP_PROC_BIND=close / OMP_PLACES=cores / OMP_NUM_THREADS are set for the child only and the benchmark's own process
    (HIP runtime threads, launch path) keeps the scheduler's placement.  The job travels as a pickle, the answer as one JSON
    line."""
    import pickle
    import subprocess
    import tempfile
    host = host_facts()
    env = dict(os.environ)
    env.setdefault('OMP_PROC_BIND', 'close')
    env.setdefault('OMP_PLACES', 'cores')
    env.setdefault('OMP_NUM_THREADS', str(host['usable_threads']))
    with tempfile.NamedTemporaryFile(suffix='.pkl', delete=False) as f:
        pickle.dump(job, f)
        path = f.name
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-child', path], env=env, capture_output=True, text=True)
    finally:
        os.unlink(path)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not lines:
        return {'error': 'cpu baseline child failed (rc %d): %s' % (r.returncode, r.stderr[-400:])}
    return json.loads(lines[-1])


def adapted_rate(spec, d, n_warm_iter, seed, target_seconds=12.):
    return _child(dict(kind='adapted', spec=spec, d=d, n_warm_iter=n_warm_iter, seed=seed, target_seconds=target_seconds))


def fixed_rate(spec, x_start, step_size, var, seed, target_accept, target_seconds):
    return _child(dict(kind='fixed', spec=spec, x_start=x_start, step_size=step_size, var=var, seed=seed,
                       target_accept=target_accept, target_seconds=target_seconds))


def two_round_rate(rounds, seed, target_seconds=14.):
    return _child(dict(kind='two_round', rounds=rounds, seed=seed, target_seconds=target_seconds))


def child_main(path):
    import pickle
    with open(path, 'rb') as f:
        job = pickle.load(f)
    kind = job.pop('kind')
    out = {'adapted': adapted_rate_here, 'fixed': fixed_rate_here, 'two_round': two_round_rate_here}[kind](**job)
    print(json.dumps(out))
