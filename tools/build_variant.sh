#!/bin/bash
# tools/build_variant.sh <name> "<extra hipcc flags for the 64k kernels, e.g. -DDSC_X=1>"  -> tools/bin/lib<name>.so
# A/B builds of the library for tools/ab.sh (same box, interleaved): every .hip file takes the extra flags.
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
# SRC_REF=<git ref> builds that commit's sources instead of the working tree
if [ -n "$SRC_REF" ]; then git -C $ROOT archive $SRC_REF dsc_amd/csrc include | tar -x -C $T
else mkdir -p $T/dsc_amd $T/include; cp -r $ROOT/dsc_amd/csrc $T/dsc_amd/csrc; cp $ROOT/include/*.h $T/include/; fi
rm -rf $T/dsc_amd/csrc/build
sed -i "s|^OUT .*|OUT = $ROOT/tools/bin/lib$NAME.so|" $T/dsc_amd/csrc/Makefile
sed -i "s|^\(FLAGS_[a-z0-9_]* = .*\)$|\1 $EXTRA|" $T/dsc_amd/csrc/Makefile
make -s -j8 -C $T/dsc_amd/csrc 2>&1 | grep -E "error|\*\*\*" -A3 || true
rm -rf $T
ls -la $ROOT/tools/bin/lib$NAME.so
