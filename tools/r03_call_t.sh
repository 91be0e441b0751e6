#!/bin/bash
# full GPU suite + the mid-size tables after the one-exchange passes (library only)
mkdir -p gpurun_out/r3t
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r3t/tests.txt || exit 1
timeout -k 10 300 python tools/bench_mid.py 1024 2048 4096 8192 16384 32768 2>/dev/null | grep -E "fft" | cut -c1-100 | tee gpurun_out/r3t/mid_f32.txt
timeout -k 10 300 python tools/bench_mid.py 1024 2048 4096 8192 16384 32768 --f64 2>/dev/null | grep -E "fft" | cut -c1-100 | tee gpurun_out/r3t/mid_f64.txt
timeout -k 10 300 python tools/bench_filter_mid.py 512 1024 2048 4096 8192 16384 32768 2>/dev/null | grep -E "filter" | cut -c1-110 | tee gpurun_out/r3t/filter_f32.txt
timeout -k 10 300 python tools/bench_filter_mid.py 512 1024 2048 4096 8192 16384 32768 --f64 2>/dev/null | grep -E "filter" | cut -c1-110 | tee gpurun_out/r3t/filter_f64.txt
