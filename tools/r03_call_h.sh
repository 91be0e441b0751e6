#!/bin/bash
mkdir -p gpurun_out/r3h
python -m pytest tests -m gpu -q -x > gpurun_out/r3h/pytest.log 2>&1; tail -3 gpurun_out/r3h/pytest.log
for L in "" tools/bin/libf64two128.so; do
  echo "== lib ${L:-current}"
  [ -n "$L" ] && export DSC_MI355X_LIB=$PWD/$L
  python tools/bench_mid.py 1024 2048 --f64 2>/dev/null | grep fft | cut -c1-90
  python tools/bench_filter_mid.py 1024 --f64 2>/dev/null | grep filter | cut -c1-90
done 2>&1 | tee gpurun_out/r3h/f64_two.txt
unset DSC_MI355X_LIB
python tools/bench_cols_small_f64.py 2>/dev/null | grep axis | tee gpurun_out/r3h/cols_small_f64_new.txt
DSC_MI355X_LIB=$PWD/tools/bin/liboldcols.so python tools/bench_cols_small_f64.py 2>/dev/null | grep axis | tee gpurun_out/r3h/cols_small_f64_old.txt
# where the team kernel's time goes (config 5): per-phase clocks of one workgroup
DSC_MI355X_LIB=$PWD/tools/bin/libfprof.so python tools/run_op.py rfft_c5_f64_262144 --iters 2 --ramp-ms 0 2>&1 | grep -E "fused_l2 profile" | tail -4 | tee gpurun_out/r3h/fused_profile.txt
DSC_MI355X_LIB=$PWD/tools/bin/libfprof.so python tools/run_op.py irfft_c5_f64_262144 --iters 2 --ramp-ms 0 2>&1 | grep -E "fused_l2 profile" | tail -4 | tee -a gpurun_out/r3h/fused_profile.txt
