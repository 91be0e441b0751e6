#!/bin/bash
# tools/build_mid_variant.sh <name> "<-D flags>": rebuilds ONLY fft_regs_mid.hip with extra flags and links it with the objects of the
# current build -> tools/bin/lib<name>.so (A/B of the mid-size kernels' group sizes on one box; DSC_MI355X_LIB selects the library)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/dsc_amd/csrc
make -s -j8 -C $C > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -I$ROOT/include -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $EXTRA -c $C/fft_regs_mid.hip -o /tmp/mid_$NAME.o
OBJS=$(ls $C/build/*.o | grep -v fft_regs_mid.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/mid_$NAME.o -ldl -o $ROOT/tools/bin/lib$NAME.so
python3 $C/check_store_hazard.py $ROOT/tools/bin/lib$NAME.so | head -1
