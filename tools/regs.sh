#!/bin/bash
# usage: tools/regs.sh <file.hip> [extra hipcc flags]   -> VGPR / spill counts per kernel
f=$1; shift
b=$(basename $f .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -I/root/repo/include -ffp-contract=fast -fno-slp-vectorize "$@" -c $f -o /tmp/$b.o -save-temps=obj 2>/dev/null
grep -E "^\s+\.(vgpr_count|vgpr_spill_count)|\.name:" /tmp/$b-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - | awk '{print $2,"vgpr",$4,"spill",$6}' | c++filt | sed 's/(anonymous namespace):://g; s/(.*)//'
