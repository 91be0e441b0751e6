#!/bin/bash
mkdir -p gpurun_out/r3m
for L in midbase small128 small512; do
  echo "== $L"
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_mid.py 64 128 256 512 2>/dev/null | grep -E "fft" | cut -c1-90
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_mid.py 64 256 512 --f64 2>/dev/null | grep -E "fft" | cut -c1-90
done 2>&1 | tee gpurun_out/r3m/small_nt.txt
