#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/prof_<tag>/ :
#   bench line (plain run), rocprofv3 kernel trace + stats of the same command, FETCH_SIZE and WRITE_SIZE in
#   separate --pmc passes, the in-kernel phase stamps and the lone-chain latency sweep.
# usage (through gpurun):  tools/profile_round.sh r01b
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 > "$out/bench_line.json" 2> "$out/bench_stderr.log"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$out/trace_bench_line.json" 2> "$out/trace_stderr.log" )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$OLDPWD/bench.py" --steps 2 --warmup 3 --no-cpu-baseline --no-extras > "$out/pmc_fetch_line.json" 2> "$out/pmc_fetch_stderr.log" )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$OLDPWD/bench.py" --steps 2 --warmup 3 --no-cpu-baseline --no-extras > "$out/pmc_write_line.json" 2> "$out/pmc_write_stderr.log" )
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$out/pmc_sq" -- python3 "$OLDPWD/bench.py" --steps 2 --warmup 3 --no-cpu-baseline --no-extras > "$out/pmc_sq_line.json" 2> "$out/pmc_sq_stderr.log" )
python3 tools/lone.py > "$out/lone_latency.log" 2>&1
for l in auto group wave; do LAYOUT=$l python3 tools/hetero_ab.py >> "$out/hetero_ab.log" 2>&1; done
find "$out" -name "*.csv" | head -40 > "$out/files.txt"
# keep the merge-back small: the per-dispatch traces are summarised by tools/summarise_profiles.py
python3 tools/summarise_profiles.py "$out" "$tag" > "$out/summary_stdout.log" 2>&1
find "$out" -name "*_kernel_trace.csv" -size +2M -delete
find "$out" -name "*counter_collection.csv" -size +2M -delete
ls -la "$out"
