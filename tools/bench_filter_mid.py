"""Times the fused mid-size filter against the three-operator composition (1 GiB of samples per case; --f64 for doubles)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

dsc.init(12 << 30, 2 << 30)
ctx = _get_ctx()


def timeit(f, reps=20, warm=10):
    for _ in range(warm):
        f()
    dsc.synchronize()
    best = 1e9
    for _ in range(3):
        B.dsc_timer_start(ctx)
        for _ in range(reps):
            f()
        best = min(best, B.dsc_timer_stop(ctx) / reps)
    return best


F64 = '--f64' in sys.argv
RD, CD, RT, CT = (np.float64, np.complex128, dsc.Dtype.F64, dsc.Dtype.C64) if F64 else (np.float32, np.complex64, dsc.Dtype.F32, dsc.Dtype.C32)
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024, 4096, 16384, 32768]:
    b = ((1 << 27) if F64 else (1 << 28)) // n
    s = dsc.from_numpy(np.random.default_rng(0).standard_normal((b, n)).astype(RD))
    H = dsc.from_numpy(np.fft.rfft(np.random.default_rng(1).standard_normal(n)).astype(CD))
    y = dsc.empty((b, n), RT)
    ms = timeit(lambda: B.dsc_filter_fft(ctx, s._c_ptr, H._c_ptr, y._c_ptr))
    path = dsc.last_fft_path()
    S = dsc.empty((b, n // 2 + 1), CT)
    P = dsc.empty((b, n // 2 + 1), CT)

    def composed():
        B.dsc_rfft(ctx, s._c_ptr, S._c_ptr, -1, -1)
        B.dsc_mul(ctx, S._c_ptr, H._c_ptr, P._c_ptr)
        B.dsc_irfft(ctx, P._c_ptr, y._c_ptr, -1, -1)
    ms2 = timeit(composed)
    nb = b * n * (16 if F64 else 8)
    print(f'filter {"f64" if F64 else "f32"} N={n:6d} B={b:6d}: fused {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}% of 8 TB/s) [{path}]   three operators {ms2:7.3f} ms ({ms2 / ms:.2f}x)', flush=True)
    del s, H, y, S, P
