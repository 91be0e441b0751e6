set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/s2_dit_full.log 2>&1; tail -3 gpurun_out/s2_dit_full.log
for L in dsc_amd/libdsc_mi355x.so tools/bin/libdifall.so; do
  n=$(basename $L .so)
  DSC_MI355X_LIB=$L python tools/bench_mid.py > gpurun_out/s2_mid_f32_$n.txt 2>&1
  DSC_MI355X_LIB=$L python tools/bench_mid.py --f64 > gpurun_out/s2_mid_f64_$n.txt 2>&1
  DSC_MI355X_LIB=$L python tools/bench_filter_mid.py > gpurun_out/s2_filter_mid_$n.txt 2>&1
  DSC_MI355X_LIB=$L python tools/bench_c5.py > gpurun_out/s2_c5_$n.txt 2>&1
  DSC_MI355X_LIB=$L python tools/bench_axis0.py > gpurun_out/s2_axis0_$n.txt 2>&1
done
echo done
