#!/bin/bash
# tools/ab.sh <op> libA.so libB.so : interleaved timing of two builds on the same box
OP=$1; A=$2; B=$3
for i in 1 2 3; do
  echo -n "A: "; DSC_MI355X_LIB=$A python3 tools/time_rfft.py $OP
  echo -n "B: "; DSC_MI355X_LIB=$B python3 tools/time_rfft.py $OP
done
