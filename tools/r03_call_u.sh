#!/bin/bash
# four-step route of the column kernel along a non-last axis: correctness, then its rate against the routes it replaces
mkdir -p gpurun_out/r3u
timeout -k 10 600 python tools/check_cols_4step.py --bench 2>&1 | grep -v "^dsc_ctx" | tee gpurun_out/r3u/check.txt
grep -q FAIL gpurun_out/r3u/check.txt && exit 1
echo "== without the route (DSC_COLS_4STEP_MIN=0)"
DSC_COLS_4STEP_MIN=0 timeout -k 10 600 python tools/check_cols_4step.py --bench 2>&1 | grep "^fft axis 0" | tee gpurun_out/r3u/old_routes.txt
