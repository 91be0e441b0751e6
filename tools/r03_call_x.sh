#!/bin/bash
# columns per tile of the short f32 column forms (64 / 128 / 256 points): library 128 / 64 / 32, libcwwide 256 / 128 / 64, libcwnarrow 64 / 32 / 16
mkdir -p gpurun_out/r3x
for L in "" cwwide cwnarrow; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 300 python tools/check_cols_4step.py --bench 2>&1 | grep "^fft axis 0"
  timeout -k 10 300 python tools/bench_cols.py 2>/dev/null | grep -E "fft" | cut -c1-130
done 2>&1 | tee gpurun_out/r3x/cw.txt
