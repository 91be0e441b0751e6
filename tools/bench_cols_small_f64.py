"""irfft / rfft / fft along axis 0 at n = 64 .. 512 in f64 (the shortest column-kernel forms): one line per case."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(12 << 30, 1 << 30)
ctx = _get_ctx()


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


rng = np.random.default_rng(0)
for n in (64, 128, 512):
    cols = (1 << 27) // n
    x = dsc.from_numpy(np.tile(rng.standard_normal((n, 2048)), (1, cols // 2048)))
    X = dsc.empty((n // 2 + 1, cols), dsc.Dtype.C64)
    ms = timeit(lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0))
    nb = n * cols * 8 + (n // 2 + 1) * cols * 16
    ms_i = timeit(lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, 0))
    print(f'axis 0 f64 n={n:4d}: rfft {ms:.3f} ms {100 * nb / ms / 8e9:5.1f}%  irfft {ms_i:.3f} ms {100 * nb / ms_i / 8e9:5.1f}%  [{dsc.last_fft_path()}]', flush=True)
    del x, X
