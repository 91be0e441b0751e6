#!/bin/bash
# tools/build_fused_variant.sh <name> "<-D flags>": rebuilds ONLY fft_xcd_fused.hip with extra flags and links it with the objects of the
# current build -> tools/bin/lib<name>.so (A/B of the team kernel's cache policies and knobs on one box; DSC_MI355X_LIB selects the library)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/dsc_amd/csrc
make -s -j8 -C $C > /dev/null
mkdir -p $ROOT/tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -I$ROOT/include -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $EXTRA -c $C/fft_xcd_fused.hip -o /tmp/fused_$NAME.o
OBJS=$(ls $C/build/*.o | grep -v fft_xcd_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/fused_$NAME.o -ldl -o $ROOT/tools/bin/lib$NAME.so
python3 $C/check_store_hazard.py $ROOT/tools/bin/lib$NAME.so | head -1
