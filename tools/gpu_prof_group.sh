#!/bin/bash
# trip timeline + SQ counters of the group kernel on the headline workload
cd /root/repo
export TMPDIR=/tmp
out=$PWD/gpurun_out/gprof
mkdir -p $out
BFHIP_LIBRARY=$PWD/bayesfast_amd/variants/libbfhip_gtrace.so timeout 300 python3 tools/trace_group.py > $out/trace_group.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fit > $out/pmc_sq_line.json 2> $out/pmc_sq_stderr.log )
python3 - <<'PY'
import glob, csv, os
import numpy as np
out = '/root/repo/gpurun_out/gprof'
rows = []
for f in glob.glob(out + '/pmc_sq/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
names = sorted(set(r['Counter_Name'] for r in rows))
for nm in names:
    v = [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == nm and 'bf_group_kernel' in r['Kernel_Name']]
    print(nm, len(v), np.mean(v[-2:]) if v else None)
PY
find $out -name "*.csv" -size +1M -delete
head -50 $out/trace_group.log
