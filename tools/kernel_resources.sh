#!/bin/bash
# usage: tools/kernel_resources.sh [game ...]   (CPU: hipcc cross-compiles)
# Registers, scratch and occupancy of every kernel of the default variant of a game's source, from the compiler's own
# remarks (-Rpass-analysis=kernel-resource-usage).  A render or logic kernel with a ScratchSize other than 0 has had an
# array pushed out of registers (a loop the unroller gave up on): bossfight's render kernel went 0.49 -> 0.72 ms that way.
R=$(cd "$(dirname "$0")/.." && pwd)
for G in ${@:-coinrun maze bossfight climber caveflyer chaser jumper}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-gpu-rdc -I$R/procgen2_amd/csrc -DPG_VARIANT=0 \
    -c $R/procgen2_amd/csrc/$G.hip -o /tmp/kr_$G.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
name=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: name=m.group(1); d={}
    for key in ('VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','VGPRs Spill','TotalSGPRs','LDS Size \[bytes/block\]'):
        m=re.search(r'remark:\s+'+key+r': (\d+)',line)
        if m: d[key]=int(m.group(1))
    if 'LDS Size' in line and name:
        short=re.sub(r'^_ZN2pg8variant0\d+[a-z]+\d+','',name)[:28]
        print('%-10s %-30s vgpr %3d sgpr %3d scratch %4d spill %3d occupancy %d lds %d' % ('$G',short,d.get('VGPRs',0),d.get('TotalSGPRs',0),d.get('ScratchSize \[bytes/lane\]',0),d.get('VGPRs Spill',0),d.get('Occupancy \[waves/SIMD\]',0),d.get('LDS Size \[bytes/block\]',0)))
"
done
