#!/bin/bash
# tools/pmc_any.sh <tag> <kernel substring> "<cmd>" "<counters>" ["<counters>" ...]
TAG=$1; KEY=$2; CMD=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- $CMD > /dev/null 2> $OUT/pmc$i.err || tail -3 $OUT/pmc$i.err
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if '$KEY' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta=(r['Grid_Size'],r['Workgroup_Size'],r['LDS_Block_Size'],r['VGPR_Count'],r['SGPR_Count'])
print('grid,wg,lds,vgpr,sgpr =',meta)
for k,v in sorted(acc.items()):
    print(f'{k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})')
PY
