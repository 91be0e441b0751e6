#!/bin/bash
# split of n = n1 n2 for the four-step along an axis: n1 >> skew (pass 2 gets shorter lines and wider tiles)
mkdir -p gpurun_out/r3w
for S in 0 1 2; do
  echo "== DSC_COLS_4STEP_SKEW=$S"
  DSC_COLS_4STEP_SKEW=$S timeout -k 10 300 python tools/check_cols_4step_real.py --bench 2>&1 | grep "^axis 0"
  DSC_COLS_4STEP_SKEW=$S timeout -k 10 300 python tools/check_cols_4step.py --bench 2>&1 | grep "^fft axis 0"
done 2>&1 | tee gpurun_out/r3w/skew.txt
