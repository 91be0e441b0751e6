#!/bin/bash
# round 3, GPU box: bench line + its rocprofv3 trace / PMC passes, the consolidated l2probe record
bash tools/profile_bench.sh r03bench > gpurun_out/r03bench.log 2>&1; tail -3 gpurun_out/r03bench.log | cut -c1-400
bash tools/l2probe_run.sh E1,E2,E3,E4,E5,E6,E7,E8 > gpurun_out/r03_l2probe_all.log 2>&1; tail -2 gpurun_out/r03_l2probe_all.log | cut -c1-200
cp gpurun_out/l2probe/summary.txt gpurun_out/r03_l2probe_raw.txt
