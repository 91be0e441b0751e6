#!/bin/bash
# tools/ab3.sh <op> libA.so libB.so [libC.so ...]: interleaved timing of several builds on the same box (3 rounds)
OP=$1; shift
for i in 1 2 3; do
  for L in "$@"; do echo -n "$(basename $L): "; DSC_MI355X_LIB=$L python3 tools/time_rfft.py $OP 2>/dev/null; done
done
