#!/bin/bash
mkdir -p gpurun_out/r3k
python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "fused or f64_262144 or config5 or every_element or two_pass or every_power" > gpurun_out/r3k/pytest.log 2>&1; tail -3 gpurun_out/r3k/pytest.log
for c in rfft_c5_f64_262144 irfft_c5_f64_262144 fft_c64_131072 ifft_c64_131072 fft_c64_65536 rfft_f64_131072 irfft_f64_131072 fft_c64_32768 rfft_f64_65536 fft_c32_65536 rfft_f32_131072 irfft_f32_131072 fft_c32_131072 rfft_f32_262144; do
  echo -n "$c: "; python3 tools/run_op.py $c 2>/dev/null | tail -1 | cut -c1-200
done 2>&1 | tee gpurun_out/r3k/fused_times.txt
python tools/stress_fused.py > gpurun_out/r3k/stress.log 2>&1; tail -3 gpurun_out/r3k/stress.log | cut -c1-200
DSC_MI355X_LIB=$PWD/tools/bin/libfprof.so python tools/run_op.py rfft_c5_f64_262144 --iters 1 --ramp-ms 0 > gpurun_out/r3k/raw.txt 2>&1; grep -E "fused_l2 members" gpurun_out/r3k/raw.txt | tail -64 > gpurun_out/r3k/fused_members.txt
