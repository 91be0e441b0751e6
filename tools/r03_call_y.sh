#!/bin/bash
# robustness of the final library: random shapes through the new axis routes, the seeded fuzz campaign, the long stress tools
mkdir -p gpurun_out/r3y
timeout -k 10 500 python tools/fuzz_axis_routes.py 80 2026 2>&1 | grep -v "^dsc_ctx" | tail -8 | tee gpurun_out/r3y/axis_fuzz.txt
timeout -k 10 500 python tools/fuzz_axis_routes.py 80 7 2>&1 | grep -v "^dsc_ctx" | tail -4 | tee -a gpurun_out/r3y/axis_fuzz.txt
timeout -k 10 500 python tests/fuzz_campaign.py 2>&1 | tail -5 | tee gpurun_out/r3y/fuzz_campaign.txt
timeout -k 10 500 python tools/stress_64k.py 2>&1 | tail -4 | tee gpurun_out/r3y/stress_64k.txt
timeout -k 10 500 python tools/stress_fused.py 2>&1 | tail -4 | tee gpurun_out/r3y/stress_fused.txt
