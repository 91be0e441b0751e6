#!/bin/bash
mkdir -p gpurun_out/r3i
python -m pytest tests -m gpu -q -x > gpurun_out/r3i/pytest.log 2>&1; tail -3 gpurun_out/r3i/pytest.log
python tools/bench_cols_small_f64.py 2>/dev/null | grep axis | tee gpurun_out/r3i/cols_small_f64.txt
for round in 1 2; do
for c in rfft_c5_f64_262144 irfft_c5_f64_262144 fft_c64_131072 fft_c64_65536 rfft_f64_131072 fft_c64_32768; do
  for L in fbase fwg3; do echo -n "$L $c: "; DSC_MI355X_LIB=$PWD/tools/bin/lib$L.so python3 tools/run_op.py $c 2>/dev/null | tail -1 | cut -c1-200; done
done; done 2>&1 | tee gpurun_out/r3i/wg3.txt
