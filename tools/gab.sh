#!/bin/bash
# A/B of group-kernel tuning builds on the headline bench: tools/gab.sh name1 name2 ...   (names of tools/gvariant.sh builds)
cd /root/repo
for n in "$@"; do
  for rep in 1 2; do
    BFHIP_LIBRARY=$PWD/bayesfast_amd/variants/libbfhip_$n.so timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fit --no-extras 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$n', '%.4g' % j['value'], '%.3f ms' % j['roofline']['kernel_ms_per_launch'], j['config']['mean_tree_size'])
"
  done
done
