#!/bin/bash
# Counters of the kernels behind bench.py's config blocks (the BASELINE configs' own targets and the DES-shaped pipeline):
# for each workload a kernel trace and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters) of
#   python3 bench.py --workload <name> --no-cpu-baseline
# into gpurun_out/cfgprof_<tag>/<name>/, then tools/summarise_config_profiles.py -> <tag>_config_counters.json.
# usage (through gpurun):  tools/profile_configs.sh r04a [workloads...]
tag=${1:-r04}; shift
wl=${@:-gauss32 banana_decay banana_round0 funnel cubic128 des_pipeline}
out=$PWD/gpurun_out/cfgprof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
for wn in $wl; do
  o=$out/$wn; mkdir -p "$o"
  w=$wn
  unset BENCH_ROUND0_ONLY
  # (the first round of config 3 on its own: its kernel also runs in the second round's adaptation launches)
  if [ "$wn" = banana_round0 ]; then w=banana_decay; export BENCH_ROUND0_ONLY=1; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$o/trace" -- python3 "$OLDPWD/bench.py" --workload $w --no-cpu-baseline > "$o/line_trace.json" 2> "$o/err_trace.log" )
  ( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$o/pmc_fetch" -- python3 "$OLDPWD/bench.py" --workload $w --no-cpu-baseline > "$o/line_fetch.json" 2> "$o/err_fetch.log" )
  ( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$o/pmc_write" -- python3 "$OLDPWD/bench.py" --workload $w --no-cpu-baseline > "$o/line_write.json" 2> "$o/err_write.log" )
  ( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$o/pmc_sq" -- python3 "$OLDPWD/bench.py" --workload $w --no-cpu-baseline > "$o/line_sq.json" 2> "$o/err_sq.log" )
done
python3 tools/summarise_config_profiles.py "$out" "$tag" > "$out/summary_stdout.log" 2>&1
find "$out" -name "*_kernel_trace.csv" -size +2M -delete
find "$out" -name "*counter_collection.csv" -size +2M -delete
find "$out" -name "*.db" -delete
tail -5 "$out/summary_stdout.log"
