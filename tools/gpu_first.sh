#!/bin/bash
# first GPU contact of the group kernel: sampler parity tests, then a short A/B of the three NUTS kernels
set -x
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_sampler.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/first_pytest.log
for k in group pipe sliced; do
  BFHIP_NUTS_KERNEL=$k timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fit > gpurun_out/first_bench_$k.json 2> gpurun_out/first_bench_$k.err
done
tail -3 gpurun_out/first_pytest.log
cat gpurun_out/first_bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    try:
        j = json.loads(l); print(j['value'], j['ms_per_step'], j['config']['mean_tree_size'], j['roofline']['kernel_ms_per_launch'])
    except Exception as e: print('ERR', l[:200])
"
