#!/usr/bin/env python3
"""tools/bench_ops.py — HIP-event timings of every operator of SURVEY 8a at BASELINE's config sizes,
with algorithmic bytes against the 8 TB/s HBM roofline.  One line per operator."""
import sys
import numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

dsc.init(40 << 30, 12 << 30)
ctx = _get_ctx()


def timeit(f, reps=10, warm=2):
    """best of three rounds after a clock ramp (the clocks need tens of milliseconds of load to come up)"""
    import time
    f()
    dsc.synchronize()
    t0 = time.perf_counter()
    f()
    dsc.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    for _ in range(max(warm, int(0.05 / one))):
        f()
    dsc.synchronize()
    best = 1e30
    for _ in range(3):
        B.dsc_timer_start(ctx)
        for _ in range(reps):
            f()
        best = min(best, B.dsc_timer_stop(ctx) / reps)
    return best


def report(name, ms, nbytes, samples=None):
    gbs = nbytes / ms / 1e6
    extra = f'  {samples / ms / 1e6:8.1f} GSamples/s' if samples else ''
    print(f'{name:58s} {ms:8.3f} ms  {gbs:8.1f} GB/s  {gbs / 80:5.1f} % of 8 TB/s{extra}  [{dsc.last_fft_path()}]', flush=True)


N = 65536
# ---- config 2: rfft / irfft f32 N=65536 B=8192
x = dsc.empty((8192, N), dsc.Dtype.F32); X = dsc.empty((8192, N // 2 + 1), dsc.Dtype.C32)
bytes2 = 8192 * (N * 4 + (N // 2 + 1) * 8)
report('C2 rfft  f32 N=65536 B=8192', timeit(lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, -1)), bytes2, 8192 * N)
report('C2 irfft f32 N=65536 B=8192', timeit(lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, -1)), bytes2, 8192 * N)
del x, X
# ---- config 3: filter N=65536 B=4096
s = dsc.empty((4096, N), dsc.Dtype.F32); y = dsc.empty((4096, N), dsc.Dtype.F32)
H = dsc.from_numpy(np.ones(N // 2 + 1, np.complex64))
S = dsc.empty((4096, N // 2 + 1), dsc.Dtype.C32); P = dsc.empty((4096, N // 2 + 1), dsc.Dtype.C32)
report('C3 fused filter irfft(rfft(s)*H) N=65536 B=4096', timeit(lambda: B.dsc_filter_fft(ctx, s._c_ptr, H._c_ptr, y._c_ptr)), 4096 * N * 8, 4096 * N)


def composed():
    B.dsc_rfft(ctx, s._c_ptr, S._c_ptr, -1, -1)
    B.dsc_mul(ctx, S._c_ptr, H._c_ptr, P._c_ptr)
    B.dsc_irfft(ctx, P._c_ptr, y._c_ptr, -1, -1)


report('C3 same as three operators (rfft, mul, irfft)', timeit(composed), 4096 * N * 8, 4096 * N)
# ---- mul c32 [4096, 32769] x [32769] and x same shape
report('mul c32 [4096,32769] x [32769] (broadcast row)', timeit(lambda: B.dsc_mul(ctx, S._c_ptr, H._c_ptr, P._c_ptr)), 4096 * 32769 * 16)
report('mul c32 [4096,32769] x [4096,32769]', timeit(lambda: B.dsc_mul(ctx, S._c_ptr, P._c_ptr, P._c_ptr)), 4096 * 32769 * 24)
# ---- reductions over the spectrum
o0 = dsc.empty((1, N // 2 + 1), dsc.Dtype.C32); o1 = dsc.empty((4096, 1), dsc.Dtype.C32)
report('sum c32 [4096,32769] axis 0', timeit(lambda: B.dsc_sum(ctx, S._c_ptr, o0._c_ptr, 0, True)), 4096 * 32769 * 8)
report('sum c32 [4096,32769] axis 1', timeit(lambda: B.dsc_sum(ctx, S._c_ptr, o1._c_ptr, 1, True)), 4096 * 32769 * 8)
report('max c32 [4096,32769] axis 1', timeit(lambda: B.dsc_max(ctx, S._c_ptr, o1._c_ptr, 1, True)), 4096 * 32769 * 8)
del s, y, S, P, o0, o1
# ---- config 5: rfft f64 N=262144 B=2048
N5 = 262144
x5 = dsc.empty((2048, N5), dsc.Dtype.F64); X5 = dsc.empty((2048, N5 // 2 + 1), dsc.Dtype.C64)
bytes5 = 2048 * (N5 * 8 + (N5 // 2 + 1) * 16)
report('C5 rfft  f64 N=262144 B=2048', timeit(lambda: B.dsc_rfft(ctx, x5._c_ptr, X5._c_ptr, -1, -1), reps=3, warm=1), bytes5, 2048 * N5)
report('C5 irfft f64 N=262144 B=2048', timeit(lambda: B.dsc_irfft(ctx, X5._c_ptr, x5._c_ptr, -1, -1), reps=3, warm=1), bytes5, 2048 * N5)
del x5, X5
# ---- other sizes through the generic path
for n, b in ((1024, 262144), (4096, 65536), (16384, 16384)):
    xs = dsc.empty((b, n), dsc.Dtype.F32); Xs = dsc.empty((b, n // 2 + 1), dsc.Dtype.C32)
    report(f'rfft f32 N={n} B={b}', timeit(lambda: B.dsc_rfft(ctx, xs._c_ptr, Xs._c_ptr, -1, -1), reps=5), b * (n * 4 + (n // 2 + 1) * 8), b * n)
    del xs, Xs
