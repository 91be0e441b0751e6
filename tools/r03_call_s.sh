#!/bin/bash
# inverse pre-pass that moves only the upper halves (library) against the two-exchange form (liboldpre); libnopair: the f32 lines of
# 16384 points with it instead of the partner loads from memory
mkdir -p gpurun_out/r3s
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "mid or small or padded or f64 or hazard or persistent or lines" 2>&1 | tail -3 | tee gpurun_out/r3s/tests.txt || exit 1
for L in oldpre "" nopair; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 300 python tools/bench_mid.py 1024 2048 4096 8192 16384 32768 2>/dev/null | grep -E "^irfft" | cut -c1-100
  [ "$L" = nopair ] || timeout -k 10 300 python tools/bench_mid.py 1024 2048 4096 8192 16384 32768 --f64 2>/dev/null | grep -E "^irfft" | cut -c1-100
done 2>&1 | tee gpurun_out/r3s/pre_once.txt
