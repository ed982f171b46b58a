#!/bin/bash
# A/B the bench across tuning builds: tools/ab.sh name1 name2 ...   (run on the GPU box)
# The launch time is set by the slowest chain of the launch, which depends on the seed and on rounding, so
# every build is run on several seeds (SEEDS) and the mean is what counts.
for v in "$@"; do
  lib=bayesfast_amd/variants/libbfhip_$v.so
  [ "$v" = base ] && lib=bayesfast_amd/libbfhip.so
  for seed in ${SEEDS:-2024 1 2 3}; do
    BFHIP_LIBRARY=$PWD/$lib python bench.py --steps ${STEPS:-4} --warmup 3 --no-cpu-baseline --seed $seed
  done | python -c "
import json,sys
v=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$v', 'mean %.4e' % (sum(j['value'] for j in v)/len(v)), ' '.join('%.3e' % j['value'] for j in v), '| ms', ' '.join('%.1f' % j['ms_per_step'] for j in v))"
done
