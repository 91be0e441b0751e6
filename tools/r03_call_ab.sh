#!/bin/bash
# PMC passes (HBM bytes, SQ counters) for the kernels added in the second half of round 3
bash tools/profile_families.sh r03fam2 "fft_axis0_65536x2048 rfft_axis0_65536x4096 irfft_axis0_65536x4096 rfft_f64_32768 fft_c64_16384 filter_f32_4096" "fft_axis0_65536x2048 rfft_axis0_65536x4096 irfft_axis0_65536x4096 rfft_f64_32768 fft_c64_16384 filter_f32_4096" > gpurun_out/r03fam2.log 2>&1
tail -3 gpurun_out/r03fam2.log | cut -c1-200
