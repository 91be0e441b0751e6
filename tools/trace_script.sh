#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of `python3 <script> [args]`, prints the per-kernel table.
# Usage: tools/trace_script.sh <tag> <script.py> [args...]   -> gpurun_out/<tag>/
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/stdout.txt 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
cat $OUT/stdout.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e6:8.4f} ms  {r['Percentage']:>6s}%")
PY
