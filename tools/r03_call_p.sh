#!/bin/bash
# cache policies of the persistent f64 real forms (lines of 16384 points): library = cached loads and stores
mkdir -p gpurun_out/r3p
for L in "" pl_s ps_s pls_s; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 200 python tools/bench_mid.py 32768 --f64 2>/dev/null | grep -E "fft" | cut -c1-100
done 2>&1 | tee gpurun_out/r3p/policies.txt
