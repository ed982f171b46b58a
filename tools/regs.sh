#!/bin/bash
# register / spill summary of every instantiation of a kernel TU: tools/regs.sh bfhip_group.hip [extra flags]
cd "$(dirname "$0")/../bayesfast_amd/csrc"
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /tmp/regs_tmp.o 2>&1 | python3 -c "
import sys,re
name=None; info={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); info[name]=[]
    for k in ('VGPRs:','AGPRs:','ScratchSize \[bytes/lane\]:','SGPRs Spill:','VGPRs Spill:'):
        m=re.search(k+r' (\d+)',l)
        if m and name: info[name].append(int(m.group(1)))
print('kernel : VGPR AGPR scratch sgpr_spill vgpr_spill')
for n,v in info.items(): print(n[:70], v)
"
