"""Times rfft / irfft / fft of contiguous rows at mid sizes (2 GiB of real samples per case).
usage: python tools/bench_mid.py [n ...]      DSC_NO_REGS_MID=1 / DSC_NO_TWO_PASS=1 select the generic kernels."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

f64 = '--f64' in sys.argv
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
dsc.init(16 << 30, 5 << 30)
ctx = _get_ctx()


def timeit(f, reps=20, warm=10, rounds=3):
    """best-of-rounds mean: the clocks need some ten milliseconds of load to ramp up"""
    for _ in range(warm):
        f()
    dsc.synchronize()
    best = 1e9
    for _ in range(rounds):
        B.dsc_timer_start(ctx)
        for _ in range(reps):
            f()
        best = min(best, B.dsc_timer_stop(ctx) / reps)
    return best


for n in sizes:
    b = (1 << (28 if f64 else 29)) // n
    rdt, cdt, esz, nm = (np.float64, dsc.Dtype.C64, 8, 'f64') if f64 else (np.float32, dsc.Dtype.C32, 4, 'f32')
    x = dsc.from_numpy(np.random.default_rng(0).standard_normal((b, n)).astype(rdt))
    X = dsc.empty((b, n // 2 + 1), cdt)
    ms = timeit(lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, -1))
    path = dsc.last_fft_path()
    nb = b * (n * esz + (n // 2 + 1) * 2 * esz)
    print(f'rfft  {nm} N={n:6d} B={b:6d}: {ms:7.3f} ms  {nb / ms / 1e9:6.3f} TB/s  {100 * nb / ms / 8e9:5.1f}% of 8 TB/s  [{path}]', flush=True)
    ms = timeit(lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, -1))
    print(f'irfft {nm} N={n:6d} B={b:6d}: {ms:7.3f} ms  {nb / ms / 1e9:6.3f} TB/s  {100 * nb / ms / 8e9:5.1f}% of 8 TB/s  [{dsc.last_fft_path()}]', flush=True)
    del x, X
    z = dsc.empty((b, n // 2), cdt)
    Z = dsc.empty((b, n // 2), cdt)
    ms = timeit(lambda: B.dsc_fft(ctx, z._c_ptr, Z._c_ptr, -1, -1))
    nb = b * n * 2 * esz
    print(f'fft   c{nm[1:]} L={n // 2:6d} B={b:6d}: {ms:7.3f} ms  {nb / ms / 1e9:6.3f} TB/s  {100 * nb / ms / 8e9:5.1f}% of 8 TB/s  [{dsc.last_fft_path()}]', flush=True)
    del z, Z
