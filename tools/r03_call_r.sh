#!/bin/bash
# the one-exchange post-pass / filter pairs on the two-pass lines (256 - 1024 points): liboncetwo against the library
mkdir -p gpurun_out/r3r
DSC_MI355X_LIB=$PWD/tools/bin/liboncetwo.so timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "filter or mid or small or padded" 2>&1 | tail -3 | tee gpurun_out/r3r/tests.txt || exit 1
for L in "" oncetwo; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 200 python tools/bench_filter_mid.py 512 1024 2048 2>/dev/null | grep -E "filter" | cut -c1-80
  timeout -k 10 200 python tools/bench_filter_mid.py 512 1024 2048 --f64 2>/dev/null | grep -E "filter" | cut -c1-80
  timeout -k 10 200 python tools/bench_mid.py 512 1024 2048 2>/dev/null | grep -E "^rfft" | cut -c1-100
  timeout -k 10 200 python tools/bench_mid.py 512 1024 2048 --f64 2>/dev/null | grep -E "^rfft" | cut -c1-100
done 2>&1 | tee gpurun_out/r3r/once_two.txt
