import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(8 << 30, 1 << 30)
rng = np.random.default_rng(2)
ok = True
for dt, L, rows in ((np.float32, 32768, 40), (np.float32, 65536, 40), (np.float32, 131072, 20), (np.float32, 262144, 6), (np.float64, 32768, 30), (np.float64, 65536, 20),
                    (np.float64, 131072, 10), (np.float64, 262144, 5)):
    x = rng.standard_normal((rows, L)).astype(dt)
    got = dsc.fft(dsc.from_numpy(x)).numpy(); p = dsc.last_fft_path()
    want = np.fft.fft(x.astype(np.float64), axis=-1)
    e = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    print('fft(real)', np.dtype(dt).name, L, p, e); ok &= e < (3e-6 if dt == np.float32 else 1e-13)
    xs = x[:, :L - 1000]
    got = dsc.fft(dsc.from_numpy(np.ascontiguousarray(xs)), n=L).numpy(); p = dsc.last_fft_path()
    want = np.fft.fft(xs.astype(np.float64), n=L, axis=-1)
    e = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    print('fft(real, padded)', np.dtype(dt).name, L, p, e); ok &= e < (3e-6 if dt == np.float32 else 1e-13)
    got = dsc.ifft(dsc.from_numpy(x)).numpy(); p = dsc.last_fft_path()
    want = np.fft.ifft(x.astype(np.float64), axis=-1)
    e = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    print('ifft(real)', np.dtype(dt).name, L, p, e); ok &= e < (3e-6 if dt == np.float32 else 1e-13)
dsc.synchronize()
print('CAST', 'OK' if ok else 'FAILED')
