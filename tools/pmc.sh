#!/bin/bash
# tools/pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]  -> gpurun_out/<tag>/pmcN + summary
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc$i.err || { tail -5 $OUT/pmc$i.err; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('$OUT/pmc*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rfft64k' in r['Kernel_Name'] or 'irfft64k' in r['Kernel_Name'] or 'filter64k' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[1][-30:] if False else r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(f'{k[0]:42s} {k[1]:28s} {sum(v)/len(v):16.1f}  (n={len(v)})')
PY
