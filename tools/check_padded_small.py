import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(4 << 30, 1 << 30)
rng = np.random.default_rng(8)
ok = True
for dt in (np.float32, np.float64):
    tol = 2e-6 if dt == np.float32 else 2e-14
    cdt = np.complex64 if dt == np.float32 else np.complex128
    for n in (4, 8, 16, 32, 64, 128, 256, 512):
        for ls in sorted({n, max(1, n - 1), max(1, n - 37), n // 2 + 1, 3, n + 5, 2 * n}):
            for rows in (1, 7, 1000):
                x = rng.standard_normal((rows, ls)).astype(dt)
                got = dsc.rfft(dsc.from_numpy(x), n=n).numpy(); p = dsc.last_fft_path()
                want = np.fft.rfft(x.astype(np.float64), n=n, axis=-1)
                e = np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-30)
                if e > tol: ok = False; print('rfft BAD', dt.__name__, n, ls, rows, p, e)
                z = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(cdt)
                got = dsc.fft(dsc.from_numpy(z), n=n).numpy(); p2 = dsc.last_fft_path()
                want = np.fft.fft(z.astype(np.complex128), n=n, axis=-1)
                e = np.max(np.abs(got - want)) / np.max(np.abs(want))
                if e > tol: ok = False; print('fft BAD', dt.__name__, n, ls, rows, p2, e)
                got = dsc.ifft(dsc.from_numpy(x), n=n).numpy(); p3 = dsc.last_fft_path()
                want = np.fft.ifft(x.astype(np.float64), n=n, axis=-1)
                e = np.max(np.abs(got - want)) / np.max(np.abs(want))
                if e > tol: ok = False; print('ifft(real) BAD', dt.__name__, n, ls, rows, p3, e)
                bins = n // 2 + 1
                lb = max(2, min(ls, bins + 3))
                Y = (rng.standard_normal((rows, lb)) + 1j * rng.standard_normal((rows, lb))).astype(cdt)
                got = dsc.irfft(dsc.from_numpy(Y), n=bins).numpy(); p4 = dsc.last_fft_path()
                Yp = np.zeros((rows, bins), np.complex128); Yp[:, :min(lb, bins)] = Y[:, :bins]
                want = np.fft.irfft(Yp, n=n, axis=-1)
                e = np.max(np.abs(got - want)) / np.max(np.abs(want))
                if e > tol: ok = False; print('irfft BAD', dt.__name__, n, lb, rows, p4, e)
        print(dt.__name__, n, 'paths', p, p2, p3, p4, flush=True)
dsc.synchronize()
print('PADSMALL', 'OK' if ok else 'FAILED')
