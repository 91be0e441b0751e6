#!/bin/bash
mkdir -p gpurun_out/r3e
for L in midbase mid256 mid128; do
  echo "== $L"
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_filter_mid.py 4096 8192 16384 2>/dev/null | grep filter | cut -c1-90
  DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python tools/bench_mid.py 4096 8192 16384 2>/dev/null | grep -E "rfft|irfft" | cut -c1-90
done 2>&1 | tee gpurun_out/r3e/mid_nt.txt
python - <<'PY' 2>/dev/null | tee gpurun_out/r3e/cols_f64_32.txt
import sys, os
sys.path.insert(0, '.')
import numpy as np
for lib in (None, 'tools/bin/liboldcols.so'):
    pass
PY
FUSED="fft_c32_131072 fft_c32_65536 fft_c64_131072 fft_c64_32768 fft_c64_65536 ifft_c32_65536 ifft_c64_131072 irfft_c5_f64_262144 irfft_f32_131072 irfft_f32_262144 irfft_f64_131072 irfft_f64_65536 rfft_c5_f64_262144 rfft_f32_131072 rfft_f32_262144 rfft_f64_131072 rfft_f64_65536"
bash tools/profile_families.sh r03fam "$FUSED" "rfft_c5_f64_262144 irfft_c5_f64_262144 rfft_f32_131072 fft_c32_65536" > gpurun_out/r3e/fam_fused.log 2>&1
tail -3 gpurun_out/r3e/fam_fused.log | cut -c1-200
