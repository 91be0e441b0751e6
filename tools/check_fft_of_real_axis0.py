import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(6 << 30, 1 << 30)
rng = np.random.default_rng(9)
ok = True
for dt in (np.float32, np.float64):
    tol = 3e-6 if dt == np.float32 else 3e-14
    for n in (32, 64, 256, 1024, 2048, 4096, 8192, 32768, 65536):
        cols = 24 if n >= 8192 else 200
        for ls in (n, n - 5):
            x = rng.standard_normal((ls, cols)).astype(dt)
            for name, f, ref in (('fft', dsc.fft, np.fft.fft), ('ifft', dsc.ifft, np.fft.ifft)):
                got = f(dsc.from_numpy(x), n=n, axis=0).numpy(); p = dsc.last_fft_path()
                want = ref(x.astype(np.float64), n=n, axis=0)
                e = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
                if e > tol: ok = False; print('BAD', name, dt.__name__, n, ls, p, e)
        print(dt.__name__, n, p, flush=True)
    x3 = rng.standard_normal((5, 256, 37)).astype(dt)           # middle axis
    got = dsc.fft(dsc.from_numpy(x3), axis=1).numpy(); p = dsc.last_fft_path()
    e = float(np.max(np.abs(got - np.fft.fft(x3.astype(np.float64), axis=1))) / 40)
    print('middle axis', dt.__name__, p, e); ok &= e < tol
dsc.synchronize()
print('CASTAXIS', 'OK' if ok else 'FAILED')
