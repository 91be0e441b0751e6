#!/bin/bash
# fused mid-size filter: pairs through one exchange of the upper halves (library) against the five-barrier form (liboldfilt)
mkdir -p gpurun_out/r3q
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "filter" 2>&1 | tail -3 | tee gpurun_out/r3q/tests.txt || exit 1
for L in oldfilt ""; do
  echo "== ${L:-library}"
  if [ -n "$L" ]; then export DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so; else unset DSC_MI355X_LIB; fi
  timeout -k 10 200 python tools/bench_filter_mid.py 2>/dev/null | grep -E "filter" | cut -c1-110
  timeout -k 10 200 python tools/bench_filter_mid.py --f64 2>/dev/null | grep -E "filter" | cut -c1-110
done 2>&1 | tee gpurun_out/r3q/filter_once.txt
