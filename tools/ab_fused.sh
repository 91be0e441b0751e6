#!/bin/bash
# tools/ab_fused.sh "<cases>" "<pmc cases>" lib1 lib2 ...: interleaved timing (tools/run_op.py, 2 rounds) of several builds of the library on one box,
# then FETCH_SIZE / WRITE_SIZE per launch of the team kernel for the PMC cases (KiB; FETCH x2 on gfx950).  Output: gpurun_out/ab_fused/
CASES=$1; PMC=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/ab_fused; mkdir -p $OUT; : > $OUT/times.txt
for round in 1 2; do
  for c in $CASES; do
    for L in "$@"; do
      echo -n "$(basename $L) $c " >> $OUT/times.txt
      DSC_MI355X_LIB=$PWD/$L python3 tools/run_op.py $c 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('path'), '%.4f ms' % d['hip_event_ms'], '%.1f %%' % (100*d['frac_of_8TBps']))" >> $OUT/times.txt
    done
  done
done
cat $OUT/times.txt
for c in $PMC; do
  for L in "$@"; do
    n=$(basename $L .so)
    for ctr in FETCH_SIZE WRITE_SIZE; do
      DSC_MI355X_LIB=$PWD/$L timeout -k 10 200 rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${n}_${c}_$ctr -- python3 tools/run_op.py $c --iters 3 --ramp-ms 0 > /dev/null 2> $OUT/pmc_${n}_${c}_$ctr.err || tail -2 $OUT/pmc_${n}_${c}_$ctr.err
    done
  done
done
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/ab_fused/pmc_*/**/*counter_collection.csv', recursive=True):
    tag = re.search(r'pmc_(.*?)_(FETCH_SIZE|WRITE_SIZE)', f).group(1)
    for r in csv.DictReader(open(f)):
        if 'fused_l2_kernel' in r['Kernel_Name'] or 'two_pass' in r['Kernel_Name']:
            acc[(tag, r['Counter_Name'])].append(float(r['Counter_Value']))
tags = sorted({t for t, _ in acc})
for t in tags:
    f = sum(acc[(t, 'FETCH_SIZE')]) / max(1, len(acc[(t, 'FETCH_SIZE')])) * 2 * 1024 / 1e9
    w = sum(acc[(t, 'WRITE_SIZE')]) / max(1, len(acc[(t, 'WRITE_SIZE')])) * 1024 / 1e9
    print(f'{t:50s} fetched {f:6.2f} GB  written {w:6.2f} GB  total {f + w:6.2f} GB')
PY
