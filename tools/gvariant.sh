#!/bin/bash
# Tuning build of the group kernel: compile bfhip_group.hip with extra -D flags (headline instantiation only) and link it
# with the other objects into bayesfast_amd/variants/libbfhip_<name>.so.  Select it with BFHIP_LIBRARY=<path>.
# usage: tools/gvariant.sh <name> [-DFLAG=..]...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/bayesfast_amd/csrc"
mkdir -p _obj ../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -ffp-contract=off -DBF_ONLY_HEADLINE "$@" -c bfhip_group.hip -o _obj/bfhip_group_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/libbfhip_$name.so _obj/bfhip_api.o _obj/bfhip_eval.o _obj/bfhip_sampler.o _obj/bfhip_group_$name.o _obj/bfhip_fit.o _obj/bfhip_poly.o _obj/bfhip_refit.o _obj/bfhip_sit.o _obj/bfhip_spline_build.o _obj/bfhip_tnuts.o _obj/bfhip_tnuts_gen.o _obj/bfhip_pld.o
echo built bayesfast_amd/variants/libbfhip_$name.so
