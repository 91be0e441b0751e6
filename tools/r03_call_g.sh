#!/bin/bash
mkdir -p gpurun_out/r3g
for L in midbase two128 two512; do
  echo "== $L"
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_mid.py 512 1024 2048 2>/dev/null | grep -E "fft" | cut -c1-90
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_filter_mid.py 1024 2048 2>/dev/null | grep filter | cut -c1-90
done 2>&1 | tee gpurun_out/r3g/two_nt.txt
for L in midbase b4_512; do
  echo "== $L"
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_mid.py 8192 2>/dev/null | grep -E "fft" | cut -c1-90
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_filter_mid.py 8192 2>/dev/null | grep filter | cut -c1-90
done 2>&1 | tee gpurun_out/r3g/b4_nt.txt
