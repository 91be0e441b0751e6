#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, the rocprofv3 kernel-trace stats of the same
# command, the two PMC passes for HBM bytes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
# gfx950) and the same two passes on a known-byte-count kernel for calibration.
# Usage: tools/profile_bench.sh <tag>     -> gpurun_out/<tag>/ ; summaries -> gpurun_out/<tag>/summary.*
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-kernels > $OUT/trace_bench.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-kernels > /dev/null 2> $OUT/pmc_fetch.err || { tail -20 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-kernels > /dev/null 2> $OUT/pmc_write.err || { tail -20 $OUT/pmc_write.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- tools/bin/calib_copy > /dev/null 2> $OUT/cal_fetch.err || { tail -20 $OUT/cal_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- tools/bin/calib_copy > /dev/null 2> $OUT/cal_write.err || { tail -20 $OUT/cal_write.err; exit 1; }
python3 tools/summarize_profile.py $OUT
