#!/bin/bash
# chunk-size sweep of config 5 for each library given (default: the in-tree build)
LIBS=${@:-dsc_amd/libdsc_mi355x.so}
for L in $LIBS; do for c in 16 32 64 128 all; do
  if [ $c = all ]; then env -u DSC_2PASS_CHUNK_ROWS DSC_MI355X_LIB=$L python3 tools/bench_c5.py 2>/dev/null; else DSC_2PASS_CHUNK_ROWS=$c DSC_MI355X_LIB=$L python3 tools/bench_c5.py 2>/dev/null; fi
done; done
