for q in 3 2 1; do for st in 4 16; do
BFHIP_TAIL_Q=$q BFHIP_TAIL_STOP=$st BENCH_ROUND0_ONLY=1 python bench.py --workload banana_decay --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('tail_q $q tail_stop $st', '%.4g'%l['value'], l.get('mean_tree_size'), l.get('launch_tail'), l.get('ms_per_launch'))"
done; done
