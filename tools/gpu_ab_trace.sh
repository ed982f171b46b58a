#!/bin/bash
# usage: tools/gpu_ab_trace.sh variant...   -> A/B bench of the variants, then the trip timeline of the gtrace build
cd /root/repo
bash tools/gab.sh "$@"
BFHIP_LIBRARY=$PWD/bayesfast_amd/variants/libbfhip_gtrace.so timeout 300 python3 tools/trace_group.py 2>&1 | sed -n 8,24p
