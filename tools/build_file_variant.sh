#!/bin/bash
# tools/build_file_variant.sh <name> <file.hip (in dsc_amd/csrc)> "<-D flags>": rebuilds ONE kernel file with extra flags and links it with the objects
# of the current build -> tools/bin/lib<name>.so (A/B on one box; DSC_MI355X_LIB selects the library)
set -e
NAME=$1; FILE=$2; EXTRA=$3
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/dsc_amd/csrc; B=$(basename $FILE .hip)
make -s -j8 -C $C > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -I$ROOT/include -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $EXTRA -c $C/$FILE -o /tmp/${B}_$NAME.o
OBJS=$(ls $C/build/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/${B}_$NAME.o -ldl -o $ROOT/tools/bin/lib$NAME.so
python3 $C/check_store_hazard.py $ROOT/tools/bin/lib$NAME.so | head -1
