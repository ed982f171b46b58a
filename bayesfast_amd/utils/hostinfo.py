"""What the host gives THIS process: the CPUs it may run on, not the ones /proc/cpuinfo lists.

bench.py's CPU baseline states a core count; a container sees every CPU of the machine in
/proc/cpuinfo while its affinity mask and its cgroup's cpu.max say what it can really use."""
import os

__all__ = ['host_cpu_facts']


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None


def _cgroup_quota():
    """CPUs' worth of quota of the process's cgroup (v2 cpu.max, v1 cfs_quota/period), None when unlimited or unknown."""
    v2 = _read('/sys/fs/cgroup/cpu.max')
    if v2:
        q = v2.split()
        if q[0] != 'max':
            return float(q[0]) / float(q[1])
        return None
    quota, period = _read('/sys/fs/cgroup/cpu/cpu.cfs_quota_us'), _read('/sys/fs/cgroup/cpu/cpu.cfs_period_us')
    if quota and period and int(quota) > 0:
        return int(quota) / int(period)
    return None


def _topology(cpus):
    """(package, core) pairs and thread-sibling structure of the CPUs in `cpus`, from sysfs."""
    cores = {}
    for c in cpus:
        base = '/sys/devices/system/cpu/cpu%d/topology/' % c
        pkg, core = _read(base + 'physical_package_id'), _read(base + 'core_id')
        if pkg is None or core is None:
            return None
        cores.setdefault((pkg, core), []).append(c)
    return cores


def host_cpu_facts():
    try:
        aff = sorted(os.sched_getaffinity(0))
    except Exception:
        aff = list(range(os.cpu_count() or 1))
    quota = _cgroup_quota()
    topo = _topology(aff)
    n_thr = len(aff)
    n_core = len(topo) if topo else n_thr
    if quota is not None:
        n_thr = max(1, min(n_thr, int(quota)))
        n_core = max(1, min(n_core, int(quota)))
    model = None
    try:
        model = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except Exception:
        pass
    # one CPU per physical core (the first sibling), for OMP_PLACES / explicit pinning
    one_per_core = sorted(min(v) for v in topo.values()) if topo else aff
    return {'model': model, 'cpuinfo_logical': os.cpu_count(), 'affinity_cpus': len(aff), 'cgroup_cpu_quota': quota,
            'physical_cores_in_affinity': len(topo) if topo else None,
            'smt': (None if not topo else max(len(v) for v in topo.values())),
            'usable_threads': n_thr, 'usable_cores': n_core, 'one_cpu_per_core': one_per_core,
            'loadavg': _read('/proc/loadavg')}
