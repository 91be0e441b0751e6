"""dsc_fft / dsc_ifft along a non-last axis at lengths that take the four-step route of the column kernel (cols_4step): every element
against numpy (f64 reference), c32 / c64 / real input, 2-D and 3-D tensors, inner sizes that do not fill the last tile.
usage: python tools/check_cols_4step.py [--bench]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

dsc.init(12 << 30, 2 << 30)
ctx = _get_ctx()
rng = np.random.default_rng(7)
worst_all = 0.0
for dt, tol in ((np.complex64, 2e-6), (np.complex128, 1e-13)):
    for shape, axis in (((4096, 72), 0), ((8192, 33), 0), ((3, 16384, 17), 1), ((65536, 24), 0), ((2, 32768, 8), 1), ((131072, 9), 0)):
        z = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
        for name in ('fft', 'ifft'):
            got = getattr(dsc, name)(dsc.from_numpy(z), axis=axis)
            path = dsc.last_fft_path()
            want = getattr(np.fft, name)(z.astype(np.complex128), axis=axis)
            err = float(np.max(np.abs(got.numpy() - want)) / np.max(np.abs(want)))
            l2 = float(np.linalg.norm(got.numpy() - want) / np.linalg.norm(want))
            worst_all = max(worst_all, l2 / tol)
            print(f'{name} {np.dtype(dt).name} {shape} axis {axis}: [{path}] max {err:.2e} l2 {l2:.2e} {"ok" if l2 <= tol and path == "cols_4step" else "FAIL"}', flush=True)
        x = z.real.copy()
        got = dsc.fft(dsc.from_numpy(x), axis=axis)
        path = dsc.last_fft_path()
        want = np.fft.fft(x.astype(np.float64), axis=axis)
        l2 = float(np.linalg.norm(got.numpy() - want) / np.linalg.norm(want))
        worst_all = max(worst_all, l2 / tol)
        print(f'fft(real) {x.dtype.name} {shape} axis {axis}: [{path}] l2 {l2:.2e} {"ok" if l2 <= tol and path == "cols_4step" else "FAIL"}', flush=True)
print('worst / tolerance', worst_all)

if '--bench' in sys.argv:
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
    for shape in ((2048, 65536), (4096, 32768), (8192, 16384), (16384, 8192), (65536, 2048), (262144, 512), (1048576, 128)):
        z = dsc.from_numpy((rng.standard_normal(shape) + 0j).astype(np.complex64))
        out = dsc.empty(shape, dsc.Dtype.C32)
        nb = 2 * z.ne * 8
        ms = timeit(lambda: B.dsc_fft(ctx, z._c_ptr, out._c_ptr, -1, 0))
        print(f'fft axis 0 c32 {shape}: {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}%) [{dsc.last_fft_path()}]', flush=True)
        del z, out
