"""Turns the raw rocprofv3 output of tools/profile_configs.sh into <tag>_config_counters.json: per workload and sampler kernel
the average duration of the timed dispatches (kernel trace), HBM-side bytes per dispatch (FETCH_SIZE, WRITE_SIZE: separate
passes, raw KB counters -> bytes) and the SQ counters, each over the last `n_timed` dispatches of that kernel in the run."""
import sys, os, glob, json, csv
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from benchlib.blocks import source_hash   # (the HIP sources the profiled library was built from: bench.py reads a stored traffic value only for the same sources)
out, tag = sys.argv[1], sys.argv[2]
N_TIMED = 2   # bench.py config_block: steps = 2 timed launches per block
SAMPLERS = ('bf_sampler_kernel', 'bf_nuts_pipe_kernel', 'bf_group_kernel', 'bf_split_kernel', 'bf_lone_kernel')


def rows(d, pattern):
    r = []
    for f in sorted(glob.glob(os.path.join(d, pattern), recursive=True)):
        with open(f) as fh:
            r += list(csv.DictReader(fh))
    return r


def line(fn):
    try:
        return json.loads([l for l in open(fn) if l.startswith('{')][-1])
    except Exception:
        return None


def by_kernel(rs, value):
    """dispatch-ordered values per sampler kernel name"""
    d = {}
    for r in rs:
        k = r.get('Kernel_Name', '')
        if k.startswith('void '):
            k = k[5:]
        if k.startswith(SAMPLERS):
            d.setdefault(k, []).append((int(r.get('Dispatch_Id', 0) or 0), value(r)))
    return {k: [v for _, v in sorted(vs)] for k, vs in d.items()}


res = {}
for wd in sorted(glob.glob(os.path.join(out, '*/'))):
    w = os.path.basename(wd.rstrip('/'))
    b = line(os.path.join(wd, 'line_trace.json'))
    if not b:
        continue
    dur = by_kernel(rows(wd, 'trace/**/*kernel_trace.csv'), lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    fetch = by_kernel([r for r in rows(wd, 'pmc_fetch/**/*counter_collection.csv') if r.get('Counter_Name') == 'FETCH_SIZE'], lambda r: float(r['Counter_Value']))
    write = by_kernel([r for r in rows(wd, 'pmc_write/**/*counter_collection.csv') if r.get('Counter_Name') == 'WRITE_SIZE'], lambda r: float(r['Counter_Value']))
    sqr = rows(wd, 'pmc_sq/**/*counter_collection.csv')
    blocks = {'round_0': b}
    if 'round_1' in b:
        blocks['round_1'] = b['round_1']
    entry = {}
    for name, blk in blocks.items():
        kern = blk['roofline']['kernel']
        stem = kern.split('<')[0]
        # the library names the instantiation loosely ("<4, ...>"): match the trace's kernels by stem and first argument
        first = kern.split('<')[1].split(',')[0].split('>')[0].strip() if '<' in kern else ''
        cand = [k for k in dur if k.startswith(stem + '<' + first)]
        if not cand:
            continue
        k = max(cand, key=lambda c: sum(dur[c][-N_TIMED:]))
        # a launch of the wave-per-chain kernel is two dispatches (the launch and its tail, in two instantiations that differ in
        # the last template argument): everything below is summed over them
        stem_of = lambda c: c.split('>(')[0].rsplit(',', 1)[0]
        parts = [c for c in cand if stem_of(c) == stem_of(k)] if k.startswith('bf_nuts_pipe_kernel') else [k]
        tot = lambda d_: float(sum(np.mean(d_[c][-N_TIMED:]) for c in parts if c in d_))
        lf = blk['value'] * blk['ms_per_launch'] * 1e-3
        e = {'kernel': k, 'kernel_named_by_library': kern, 'timed_dispatches': N_TIMED, 'avg_ms_kernel_trace': tot(dur),
             'ms_per_launch_hip_events': blk['ms_per_launch'], 'leapfrogs_per_launch': lf}
        if len(parts) > 1:
            e['dispatches_per_launch'] = {c: float(np.mean(dur[c][-N_TIMED:])) for c in parts}
        if k in fetch and k in write:
            f, wv = tot(fetch) * 1024., tot(write) * 1024.
            e.update(fetch_bytes_per_launch_raw=f, write_bytes_per_launch=wv, hbm_bytes_per_launch=f + wv, hbm_bytes_per_leapfrog=(f + wv) / lf,
                     note='FETCH_SIZE raw (KB -> bytes): these kernels read 8 B per lane, outside the 16-B/lane calibration of the x2 '
                          'correction of MI355X_MICROARCH.md, so the raw value is a lower bound and twice it an upper bound')
        sq = {}
        for nm in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS',
                   'SQ_VALU_MFMA_BUSY_CYCLES'):
            v = by_kernel([r for r in sqr if r.get('Counter_Name') == nm], lambda r: float(r['Counter_Value']))
            if v.get(k):
                sq[nm] = tot(v)
        if sq.get('SQ_WAVE_CYCLES'):
            wc = sq['SQ_WAVE_CYCLES']
            sq['share_executing'] = sq.get('SQ_ACTIVE_INST_ANY', 0.) / wc
            sq['share_waiting'] = sq.get('SQ_WAIT_ANY', 0.) / wc
            sq['mfma_busy_share_of_wave_cycles'] = sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.) / wc
        e['sq'] = sq
        e['source_hash'] = source_hash()
        for key in ('value', 'mean_tree_size', 'max_tree_depth', 'launch_tail', 'chains', 'dim'):
            if key in blk:
                e[key] = blk[key]
        entry[name] = e
    res[w] = entry
if 'banana_round0' in res and 'banana_decay' in res:   # (round 0's kernel also runs in round 1's adaptation: its own run is the clean one)
    res['banana_decay'].pop('round_0', None)
json.dump(res, open(os.path.join(out, '%s_config_counters.json' % tag), 'w'), indent=1)
json.dump(res, open(os.path.join(out, 'config_traffic.json'), 'w'), indent=1)   # (copied to profiles/config_traffic.json: what bench.py's roofline.traffic reads)
print(json.dumps(res, indent=1)[:5000])
