#!/bin/bash
# round 3, GPU box: one rocprofv3 summary per kernel family (part 1 / part 2) + PMC passes for the BASELINE kernels
PART=$1
ALL=$(python3 tools/run_op.py --list)
set -- $ALL
N=$#; H=$((N / 2))
if [ "$PART" = 1 ]; then CASES=$(echo $ALL | cut -d' ' -f1-$H); PMC="rfft64k irfft64k filter64k";
else CASES=$(echo $ALL | cut -d' ' -f$((H + 1))-$N); PMC="rfft_c5_f64_262144 irfft_c5_f64_262144 rfft_f32_131072 fft_c32_65536"; fi
bash tools/profile_families.sh r03fam "$CASES" "$PMC" > gpurun_out/r03fam_part$PART.log 2>&1
tail -3 gpurun_out/r03fam_part$PART.log | cut -c1-200
