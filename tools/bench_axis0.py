"""Times transforms along axis 0 (strided lines) against transpose -> last-axis transform -> transpose."""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

dsc.init(16 << 30, 4 << 30)
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


for shape in ((4096, 4096), (1024, 65536), (65536, 512), (256, 262144)):
    z = dsc.from_numpy((np.random.default_rng(0).standard_normal(shape) + 0j).astype(np.complex64))
    nb = 2 * z.ne * 8
    ms = timeit(lambda: dsc.fft(z, axis=0))
    path = dsc.last_fft_path()
    ms2 = timeit(lambda: dsc.transpose(dsc.fft(dsc.transpose(z))))
    print(f'fft axis 0 c32 {shape}: direct {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}%) [{path}]   via transposes {ms2:7.3f} ms ({100 * nb / ms2 / 8e9:4.1f}%)', flush=True)
    x = dsc.from_numpy(np.random.default_rng(0).standard_normal(shape).astype(np.float32))
    ms = timeit(lambda: dsc.rfft(x, axis=0))
    path = dsc.last_fft_path()
    ms2 = timeit(lambda: dsc.transpose(dsc.rfft(dsc.transpose(x))))
    nb = x.ne * 4 + (shape[0] // 2 + 1) * shape[1] * 8
    print(f'rfft axis 0 f32 {shape}: direct {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}%) [{path}]   via transposes {ms2:7.3f} ms ({100 * nb / ms2 / 8e9:4.1f}%)', flush=True)
    del z, x
