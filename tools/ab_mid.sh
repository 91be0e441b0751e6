# tools/ab_mid.sh libA.so libB.so — mid sizes (rfft / irfft / fft N = 4096, 8192 in f32 and f64, the fused filter), two interleaved rounds
for i in 1 2; do
  for L in "$@"; do
    echo "== $(basename $L)"
    DSC_MI355X_LIB=$L python3 tools/bench_mid.py 4096 8192 2>/dev/null | grep -v ctx_init
    DSC_MI355X_LIB=$L python3 tools/bench_mid.py 4096 --f64 2>/dev/null | grep -v ctx_init
    DSC_MI355X_LIB=$L python3 tools/bench_filter_mid.py 2>/dev/null | grep -v ctx_init | cut -c1-90
  done
done
