#!/bin/bash
# forward real post-pass that moves only the upper halves (library) against the two-exchange form (liboldpost), f32 and f64
mkdir -p gpurun_out/r3o
timeout -k 10 600 python -m pytest tests/test_gpu_headline.py -m gpu -x -q -k "mid or f64 or hazard" 2>&1 | tail -3 | tee gpurun_out/r3o/tests.txt || exit 1
for L in oldpost ""; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 200 python tools/bench_mid.py 4096 8192 16384 32768 2>/dev/null | grep -E "^rfft" | cut -c1-100
  timeout -k 10 200 python tools/bench_mid.py 4096 8192 16384 32768 --f64 2>/dev/null | grep -E "^rfft" | cut -c1-100
done 2>&1 | tee gpurun_out/r3o/post_once.txt
