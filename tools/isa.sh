#!/bin/bash
# Dump the gfx950 ISA of one HIP source of the library and print per-kernel register/spill statistics.
# usage: tools/isa.sh bfhip_sampler.hip [/tmp/asm/out.s]
src=${1:-bfhip_sampler.hip}
out=${2:-/tmp/asm/${src%.hip}.s}
mkdir -p "$(dirname "$out")"
cd "$(dirname "$0")/../bayesfast_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o "$out" "$src" 2>/dev/null
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_count|sgpr_spill_count|private_segment_fixed_size):" "$out" | paste - - - - - - | sed 's/\s\+/ /g'
