#!/bin/bash
# tools/l2probe_run.sh [L2P_ONLY tags]: timing run + two PMC passes of tools/bin/l2probe, merged by tools/l2probe_sum.py
# (run on the GPU box: gpurun -- 'bash tools/l2probe_run.sh E1,E2')
export TMPDIR=/tmp
OUT=gpurun_out/l2probe; mkdir -p $OUT
[ -n "$1" ] && export L2P_ONLY=$1
tools/bin/l2probe > $OUT/timing.txt 2>&1 || { tail -5 $OUT/timing.txt; exit 1; }
export L2P_PMC=1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- tools/bin/l2probe > $OUT/pmc_$c.txt 2> $OUT/pmc_$c.err || { tail -5 $OUT/pmc_$c.err; exit 1; }
done
python3 tools/l2probe_sum.py $OUT | tee $OUT/summary.txt
