#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
out=$PWD/gpurun_out/gprof
mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -i "icache\|ifetch\|SQ_INST_LEVEL\|SQ_WAIT_INST\|SQ_INSTS_\|SQ_BUSY\|SQ_ACTIVE_INST" | head -60 > $out/counters_list.txt
( cd /tmp && timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $out/pmc_ic -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fit > $out/pmc_ic_line.json 2> $out/pmc_ic_stderr.log )
python3 - <<'PY'
import glob, csv
import numpy as np
out = '/root/repo/gpurun_out/gprof'
rows = []
for f in glob.glob(out + '/pmc_ic/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
for nm in sorted(set(r['Counter_Name'] for r in rows)):
    v = [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == nm and 'bf_group_kernel' in r['Kernel_Name']]
    print(nm, len(v), np.mean(v[-2:]) if v else None)
PY
tail -5 $out/pmc_ic_stderr.log
cat $out/counters_list.txt | cut -c1-150 | head -40
find $out -name "*.csv" -size +1M -delete
