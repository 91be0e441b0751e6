"""dsc_rfft / dsc_irfft along a non-last axis at lengths that take the real four-step route (cols_4step_real: two columns as one complex
column): every element against numpy (f64), f32 / f64, 2-D and 3-D, column counts that leave the last tile ragged, the imaginary
parts of bins 0 and n/2 ignored by irfft.   usage: python tools/check_cols_4step_real.py [--bench]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx

dsc.init(12 << 30, 2 << 30)
ctx = _get_ctx()
rng = np.random.default_rng(11)
bad = 0
for dt, tol in ((np.float32, 2e-6), (np.float64, 1e-13)):
    for shape, axis in (((8192, 40), 0), ((16384, 34), 0), ((3, 32768, 18), 1), ((65536, 24), 0), ((2, 131072, 16), 1), ((262144, 22), 0)):
        x = rng.standard_normal(shape).astype(dt)
        got = dsc.rfft(dsc.from_numpy(x), axis=axis)
        path = dsc.last_fft_path()
        want = np.fft.rfft(x.astype(np.float64), axis=axis)
        l2 = float(np.linalg.norm(got.numpy() - want) / np.linalg.norm(want))
        mx = float(np.max(np.abs(got.numpy() - want)) / np.max(np.abs(want)))
        g = got.numpy()
        edge = np.take(g, 0, axis=axis).imag, np.take(g, -1, axis=axis).imag
        ok = l2 <= tol and mx <= 10 * tol and path == 'cols_4step_real' and not edge[0].any() and not edge[1].any()
        bad += not ok
        print(f'rfft {np.dtype(dt).name} {shape} axis {axis}: [{path}] l2 {l2:.2e} max {mx:.2e} {"ok" if ok else "FAIL"}', flush=True)
        Y = want.astype(np.complex64 if dt == np.float32 else np.complex128)
        idx0 = [slice(None)] * Y.ndim
        idx0[axis] = 0
        idxl = list(idx0)
        idxl[axis] = -1
        Yq = Y.copy()
        Yq[tuple(idx0)] += 2j                      # ignored (dsc_fft.h:227-228)
        Yq[tuple(idxl)] -= 3j
        back = dsc.irfft(dsc.from_numpy(Yq), axis=axis)
        path = dsc.last_fft_path()
        wantb = np.fft.irfft(Y.astype(np.complex128), axis=axis)
        l2 = float(np.linalg.norm(back.numpy() - wantb) / np.linalg.norm(wantb))
        mx = float(np.max(np.abs(back.numpy() - wantb)) / np.max(np.abs(wantb)))
        ok = l2 <= tol and mx <= 10 * tol and path == 'cols_4step_real'
        bad += not ok
        print(f'irfft {np.dtype(dt).name} {shape} axis {axis}: [{path}] l2 {l2:.2e} max {mx:.2e} {"ok" if ok else "FAIL"}', flush=True)
print('failures', bad)

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
    for shape in ((4096, 65536), (8192, 32768), (16384, 16384), (65536, 4096), (262144, 1024), (1048576, 256)):
        x = dsc.from_numpy(rng.standard_normal(shape).astype(np.float32))
        X = dsc.empty((shape[0] // 2 + 1, shape[1]), dsc.Dtype.C32)
        nb = x.ne * 4 + X.ne * 8
        ms = timeit(lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0))
        p1 = dsc.last_fft_path()
        ms2 = timeit(lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, 0))
        print(f'axis 0 f32 {shape}: rfft {ms:7.3f} ms ({100 * nb / ms / 8e9:4.1f}%) [{p1}]   irfft {ms2:7.3f} ms ({100 * nb / ms2 / 8e9:4.1f}%) [{dsc.last_fft_path()}]', flush=True)
        del x, X
sys.exit(1 if bad else 0)
