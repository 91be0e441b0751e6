"""tools/check_tiny_axis0.py — fft_tiny_cols_kernel (complex lengths 2 .. 16 along a non-last axis) against numpy: every transform,
full / padded / cropped lines, odd inner sizes, a middle axis."""
import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(2 << 30, 1 << 30)
rng = np.random.default_rng(4)
ok = True
paths = set()
for dt in (np.float32, np.float64):
    tol = 3e-6 if dt == np.float32 else 3e-14
    cdt = np.complex64 if dt == np.float32 else np.complex128
    for n in (4, 8, 16, 32):
        for ls in sorted({n, n - 1, 3, n + 3}):
            for cols in (1, 37, 1000):
                x = rng.standard_normal((ls, cols)).astype(dt)
                z = (rng.standard_normal((ls, cols)) + 1j * rng.standard_normal((ls, cols))).astype(cdt)
                checks = [('rfft', dsc.rfft(dsc.from_numpy(x), n=n, axis=0), np.fft.rfft(x.astype(np.float64), n=n, axis=0))]
                paths.add(dsc.last_fft_path())
                if n <= 16:
                    checks += [('fft', dsc.fft(dsc.from_numpy(z), n=n, axis=0), np.fft.fft(z.astype(np.complex128), n=n, axis=0)),
                               ('ifft', dsc.ifft(dsc.from_numpy(z), n=n, axis=0), np.fft.ifft(z.astype(np.complex128), n=n, axis=0)),
                               ('fft(real)', dsc.fft(dsc.from_numpy(x), n=n, axis=0), np.fft.fft(x.astype(np.float64), n=n, axis=0))]
                bins = n // 2 + 1
                lb = max(2, min(ls, bins + 2))
                Y = (rng.standard_normal((lb, cols)) + 1j * rng.standard_normal((lb, cols))).astype(cdt)
                Yp = np.zeros((bins, cols), np.complex128); Yp[:min(lb, bins)] = Y[:bins]
                checks.append(('irfft', dsc.irfft(dsc.from_numpy(Y), n=bins, axis=0), np.fft.irfft(Yp, n=n, axis=0)))
                paths.add(dsc.last_fft_path())
                for name, got, want in checks:
                    e = float(np.max(np.abs(got.numpy() - want)) / max(np.max(np.abs(want)), 1e-30))
                    if got.numpy().shape != want.shape or e > tol:
                        ok = False; print('BAD', name, dt.__name__, n, ls, cols, e, got.numpy().shape, want.shape)
    x3 = rng.standard_normal((5, 16, 37)).astype(dt)
    e = float(np.max(np.abs(dsc.rfft(dsc.from_numpy(x3), axis=1).numpy() - np.fft.rfft(x3.astype(np.float64), axis=1))))
    paths.add(dsc.last_fft_path()); ok &= e < 20 * tol
dsc.synchronize()
print(paths)
print('TINYAXIS', 'OK' if ok else 'FAILED')
