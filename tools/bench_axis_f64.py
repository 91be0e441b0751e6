"""f64 along axis 0 through the four-step routes (on / off via DSC_COLS_4STEP_MIN=0 DSC_COLS_4STEP_REAL_MIN=0)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

dsc.init(14 << 30, 2 << 30)
ctx = _get_ctx()
rng = np.random.default_rng(5)


def timeit(f, reps=10, warm=5):
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


for n, C in ((4096, 16384), (8192, 8192), (65536, 1024), (262144, 256)):
    z = dsc.from_numpy((rng.standard_normal((n, C)) + 0j).astype(np.complex128))
    out = dsc.empty((n, C), dsc.Dtype.C64)
    ms = timeit(lambda: B.dsc_fft(ctx, z._c_ptr, out._c_ptr, -1, 0))
    line = f'f64 axis 0 n={n:6d} x {C:5d}: fft {ms:7.3f} ms ({100 * 2 * z.ne * 16 / ms / 8e9:4.1f}%) [{dsc.last_fft_path()}]'
    del z, out
    x = dsc.from_numpy(rng.standard_normal((n, 2 * C)).astype(np.float64))
    X = dsc.empty((n // 2 + 1, 2 * C), dsc.Dtype.C64)
    nb = x.ne * 8 + X.ne * 16
    ms = timeit(lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0))
    p = dsc.last_fft_path()
    ms2 = timeit(lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, 0))
    print(line + f'   rfft {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}%) [{p}]  irfft {ms2:7.3f} ms ({100 * nb / ms2 / 8e9:4.1f}%)', flush=True)
    del x, X
