#!/bin/bash
# Runs on the GPU box (via gpurun): one rocprofv3 kernel-trace summary per kernel family, plus PMC passes
# (HBM bytes, SQ wait/LDS counters) for the cases named in $PMC_CASES.  Program goes directly after `--`.
# Usage: tools/profile_families.sh <tag> ["case1 case2 ..."] ["pmc_case1 ..."]  -> gpurun_out/<tag>/<case>/
TAG=${1:-r02}
CASES=${2:-$(python3 tools/run_op.py --list)}
PMC_CASES=${3:-"rfft64k irfft64k filter64k"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in $CASES; do
  D=$OUT/$c; mkdir -p $D
  python3 tools/run_op.py $c > $D/plain.json 2> $D/plain.err || { echo "$c: plain run failed"; tail -5 $D/plain.err; continue; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 tools/run_op.py $c > $D/traced.json 2> $D/trace.err || { echo "$c: trace failed"; tail -5 $D/trace.err; continue; }
  f=$(find $D/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $D/kernel_stats.csv
  rm -rf $D/trace
  echo "$c: $(cat $D/plain.json)"
done
PASSES=("FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
        "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_INSTS_WAVE32_LDS GRBM_GUI_ACTIVE"
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
        "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
        "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum")
# (a TA_* pass — TA_BUSY_avr, TA_BUFFER_WAVEFRONTS_sum ... — aborts rocprofv3 with signal 6 on this image and then hangs: left out)
for c in $PMC_CASES; do
  D=$OUT/$c; mkdir -p $D
  i=0
  for set in "${PASSES[@]}"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $D/pmc$i -- python3 tools/run_op.py $c --iters 3 --ramp-ms 0 > /dev/null 2> $D/pmc$i.err || { echo "$c pass $i ($set) failed"; tail -2 $D/pmc$i.err; }
    echo "$c pmc pass $i done"

  done
done
# calibration of FETCH_SIZE / WRITE_SIZE on a known byte count in the same access widths
if [ -x tools/bin/calib_copy ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- tools/bin/calib_copy > /dev/null 2> $OUT/cal_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- tools/bin/calib_copy > /dev/null 2> $OUT/cal_write.err
fi
python3 tools/summarize_families.py $OUT > $OUT/summary.md
cat $OUT/summary.md
