"""Turns the raw rocprofv3 output of tools/profile_round.sh into the small files committed under profiles/."""
import sys, os, glob, json, csv
import numpy as np
out, tag = sys.argv[1], sys.argv[2]
KERNEL = None  # the kernel the bench line names (roofline.kernel): set below


def rows(pattern):
    fs = sorted(glob.glob(os.path.join(out, pattern), recursive=True))
    r = []
    for f in fs:
        with open(f) as fh:
            r += list(csv.DictReader(fh))
    return r


def line(fn):
    try:
        return json.loads([l for l in open(os.path.join(out, fn)) if l.startswith('{')][-1])
    except Exception:
        return None


def round1_slice(l):
    """Dispatch indices (of the dominant sampler kernel, in launch order) of the headline's round-1 TIMED launches in the run that
    printed line `l`: per round n_adapt / iters adaptation launches, `warmup` untimed launches, then the round's timed ones."""
    c = l['config']
    na = c['nuts_adaptation_iterations_per_round'] // c['nuts_iterations_per_step']
    k0, k1 = c['timed_launches_round0'], c['timed_launches_round1']
    i1 = (na + l['warmup'] + k0) + na + l['warmup']
    return slice(i1, i1 + k1)


res = {}
b = line('bench_line.json')
FULL = (b or {}).get('roofline', {}).get('kernel') or 'bf_group_kernel'
KERNEL = FULL.split('<')[0]
# kernel trace: durations of the sampler dispatches
STEM = FULL.split("<")[0] + "<" + FULL.split("<")[1].split(",")[0].split(">")[0] if "<" in FULL else FULL   # (the library names the instantiation loosely: stem and first argument)
kt = [r for r in rows("trace/**/*kernel_trace.csv") if STEM in r.get("Kernel_Name", "")]
dur = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6 for r in kt])
tb = line('trace_bench_line.json')
if len(dur):
    res['kernel_trace'] = {'kernel': kt[0]['Kernel_Name'], 'calls': int(len(dur)), 'total_ms': float(dur.sum()), 'avg_ms': float(dur.mean()),
                           'min_ms': float(dur.min()), 'max_ms': float(dur.max())}
    if tb:
        # the roofline's kernel_ms_per_launch is the HIP-event time of round 1's timed launches: the same dispatches in the trace
        sl = round1_slice(tb)
        res['kernel_trace'].update({'round1_timed_dispatches': [sl.start, sl.stop], 'ms_round1_timed_dispatches': [float(v) for v in dur[sl]],
                                    'avg_ms_round1_timed_dispatches': float(dur[sl].mean()) if len(dur[sl]) else None,
                                    'kernel_ms_per_launch_hip_events_of_the_traced_run': tb['roofline']['kernel_ms_per_launch'],
                                    'kernel_ms_per_launch_hip_events_of_the_plain_run': (b or {}).get('roofline', {}).get('kernel_ms_per_launch')})
st = rows('trace/**/*kernel_stats.csv')
if st:
    with open(os.path.join(out, '%s_bench_kernel_stats.csv' % tag), 'w') as fh:
        w = csv.DictWriter(fh, fieldnames=list(st[0].keys()))
        w.writeheader()
        w.writerows(st)


def counter(pattern, name):
    v = [float(r['Counter_Value']) for r in rows(pattern) if r.get('Counter_Name') == name and FULL in r.get('Kernel_Name', '')]
    return v


res['bench_line'] = b
for nm, pat in (('FETCH_SIZE', 'pmc_fetch/**/*counter_collection.csv'), ('WRITE_SIZE', 'pmc_write/**/*counter_collection.csv')):
    v = counter(pat, nm)
    res[nm + '_per_dispatch_raw_KB'] = v
fl, wl = line('pmc_fetch_line.json'), line('pmc_write_line.json')
if res.get('FETCH_SIZE_per_dispatch_raw_KB') and res.get('WRITE_SIZE_per_dispatch_raw_KB') and fl and wl:
    # round 1's timed dispatches of the counter runs; counters are in KB
    f = float(np.mean(res['FETCH_SIZE_per_dispatch_raw_KB'][round1_slice(fl)])) * 1024.
    w = float(np.mean(res['WRITE_SIZE_per_dispatch_raw_KB'][round1_slice(wl)])) * 1024.
    lf_f = fl['config3_round1']['leapfrogs_timed'] / fl['config3_round1']['steps_timed']
    lf_w = wl['config3_round1']['leapfrogs_timed'] / wl['config3_round1']['steps_timed']
    tr = {'dim': 64, 'kernel': FULL, 'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 3, the timed dispatch of round 1',
          'fetch_bytes_per_launch_raw': f, 'write_bytes_per_launch': w, 'leapfrogs_per_launch': 0.5 * (lf_f + lf_w),
          'note': 'FETCH_SIZE is the raw counter (KB -> bytes); the gfx950 x2 correction of MI355X_MICROARCH.md is calibrated for 16-B/lane '
                  'streaming reads only and this kernel reads 8 B/lane, so the raw value is a lower bound and 2x it an upper bound',
          'hbm_bytes_per_leapfrog': f / lf_f + w / lf_w, 'hbm_bytes_per_leapfrog_upper': 2 * f / lf_f + w / lf_w}
    res['hbm_traffic'] = tr
    json.dump(tr, open(os.path.join(out, 'hbm_traffic.json'), 'w'), indent=1)
sq = {}
for nm in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_VALU_MFMA_BUSY_CYCLES'):
    v = counter('pmc_sq/**/*counter_collection.csv', nm)
    if v:
        sq[nm] = float(np.mean(v[round1_slice(line('pmc_sq_line.json'))])) if line('pmc_sq_line.json') else float(np.mean(v[-2:]))
res['sq_counters_per_timed_dispatch'] = sq
json.dump(res, open(os.path.join(out, '%s_summary.json' % tag), 'w'), indent=1)
print(json.dumps(res, indent=1)[:6000])
