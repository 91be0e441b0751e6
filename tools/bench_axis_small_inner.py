"""Four-step axis routes against the transpose route at SMALL inner extents (few columns per line): where is the crossover?
usage: python tools/bench_axis_small_inner.py    (run twice: plain, and with DSC_COLS_4STEP_MIN=0 DSC_COLS_4STEP_REAL_MIN=0)"""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

dsc.init(12 << 30, 2 << 30)
ctx = _get_ctx()
rng = np.random.default_rng(3)


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


for n in (4096, 65536):
    for C in (8, 16, 32, 64, 128):
        S = max(1, (1 << 26) // (n * C))
        z = dsc.from_numpy((rng.standard_normal((S, n, C)) + 0j).astype(np.complex64))
        out = dsc.empty((S, n, C), dsc.Dtype.C32)
        ms = timeit(lambda: B.dsc_fft(ctx, z._c_ptr, out._c_ptr, -1, 1))
        p = dsc.last_fft_path()
        nb = 2 * z.ne * 8
        line = f'n={n:6d} C={C:4d} S={S:5d}: fft {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}%) [{p}]'
        del z, out
        if C >= 16 and n >= 8192:
            x = dsc.from_numpy(rng.standard_normal((S, n, C)).astype(np.float32))
            X = dsc.empty((S, n // 2 + 1, C), dsc.Dtype.C32)
            ms = timeit(lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 1))
            p = dsc.last_fft_path()
            nb = x.ne * 4 + X.ne * 8
            ms2 = timeit(lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, 1))
            line += f'   rfft {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}%) [{p}]  irfft {ms2:7.3f} ms ({100 * nb / ms2 / 8e9:4.1f}%)'
            del x, X
        print(line, flush=True)
