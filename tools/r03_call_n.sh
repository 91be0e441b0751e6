#!/bin/bash
# f64 lines of 16384 points: persistent pipelined form (library) against the one-line-per-launch-group form (libmidnopipe)
mkdir -p gpurun_out/r3n
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r3n/tests.txt || exit 1
for L in midnopipe ""; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 200 python tools/bench_mid.py 32768 --f64 2>/dev/null | grep -E "fft" | cut -c1-100
done 2>&1 | tee gpurun_out/r3n/f64_pipe.txt
