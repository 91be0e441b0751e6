#!/bin/bash
# real four-step along a non-last axis: correctness, then its rate against the transpose route
mkdir -p gpurun_out/r3v
timeout -k 10 600 python tools/check_cols_4step_real.py --bench 2>&1 | grep -v "^dsc_ctx" | tee gpurun_out/r3v/check.txt
grep -q FAIL gpurun_out/r3v/check.txt && exit 1
echo "== without the route (DSC_COLS_4STEP_REAL_MIN=0)"
DSC_COLS_4STEP_REAL_MIN=0 timeout -k 10 600 python tools/check_cols_4step_real.py --bench 2>&1 | grep "^axis 0" | tee gpurun_out/r3v/old_routes.txt
