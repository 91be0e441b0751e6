#!/bin/bash
# column kernel: one-exchange real passes (library: forward + f64 inverse) against the two-exchange forms (libcolsold); libcolsnopair:
# the f32 inverse with it too instead of the partner loads from memory
mkdir -p gpurun_out/r3aa
timeout -k 10 600 python -m pytest tests/test_gpu_headline.py -m gpu -x -q -k "column or strided or axis" 2>&1 | tail -3 | tee gpurun_out/r3aa/tests.txt || exit 1
DSC_MI355X_LIB=$PWD/tools/bin/libcolsnopair.so timeout -k 10 600 python -m pytest tests/test_gpu_headline.py -m gpu -x -q -k "column or strided or axis" 2>&1 | tail -3 | tee -a gpurun_out/r3aa/tests.txt || exit 1
for L in colsold "" colsnopair; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 300 python tools/bench_cols.py 2>/dev/null | grep -v "^dsc_ctx" | cut -c1-150
done 2>&1 | tee gpurun_out/r3aa/cols_once.txt
