"""tools/bench_cols.py — the column kernel (transforms along axis 0) at larger tensors, one line per case."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(24 << 30, 4 << 30)
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
for n, cols in ((256, 1 << 20), (512, 1 << 19), (1024, 1 << 18), (2048, 1 << 17), (4096, 1 << 16), (8192, 1 << 15)):
    blk = rng.standard_normal((n, 4096)).astype(np.float32)
    x = dsc.from_numpy(np.tile(blk, (1, cols // 4096)))
    X = dsc.empty((n // 2 + 1, cols), dsc.Dtype.C32)
    ms = timeit(lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0))
    nb = n * cols * 4 + (n // 2 + 1) * cols * 8
    p1 = dsc.last_fft_path()
    ms_i = timeit(lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, 0))
    del x, X
    zc = cols // 2
    z = dsc.from_numpy(np.tile((blk[:, :2048] + 1j * blk[:, 2048:]).astype(np.complex64), (1, zc // 2048)))
    Z = dsc.empty((n, zc), dsc.Dtype.C32)
    ms_c = timeit(lambda: B.dsc_fft(ctx, z._c_ptr, Z._c_ptr, -1, 0))
    p2 = dsc.last_fft_path()
    print(f'axis 0, n={n:5d}: rfft f32 [{n},{cols}] {ms:.3f} ms {100 * nb / ms / 8e9:5.1f}% [{p1}]  irfft {ms_i:.3f} ms {100 * nb / ms_i / 8e9:5.1f}%  '
          f'fft c32 [{n},{zc}] {ms_c:.3f} ms {100 * 2 * n * zc * 8 / ms_c / 8e9:5.1f}% [{p2}]', flush=True)
    del z, Z
