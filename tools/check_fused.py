import sys, os, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
from tests.helpers import rel_l2, max_rel
dsc.init(4 << 30, 1 << 30)
rng = np.random.default_rng(5)
N = 131072
ok = True
for rows in (1, 3, 50, 150):
    x = rng.standard_normal((rows, N)).astype(np.float32)
    X = dsc.rfft(dsc.from_numpy(x)); p = dsc.last_fft_path()
    want = np.fft.rfft(x.astype(np.float64), axis=-1)
    e = rel_l2(X.numpy(), want), max_rel(X.numpy(), want)
    print('rfft', rows, p, e); ok &= e[0] < 1e-5
    y = dsc.irfft(X); p = dsc.last_fft_path()
    wy = np.fft.irfft(want, axis=-1)
    e = rel_l2(y.numpy(), wy), max_rel(y.numpy(), wy)
    print('irfft', rows, p, e); ok &= e[0] < 1e-5
    z = (rng.standard_normal((rows, 65536)) + 1j * rng.standard_normal((rows, 65536))).astype(np.complex64)
    Z = dsc.fft(dsc.from_numpy(z)); p = dsc.last_fft_path()
    wz = np.fft.fft(z.astype(np.complex128), axis=-1)
    e = rel_l2(Z.numpy(), wz), max_rel(Z.numpy(), wz)
    print('fft', rows, p, e); ok &= e[0] < 1e-5
    zi = dsc.ifft(Z); p = dsc.last_fft_path()
    e = rel_l2(zi.numpy(), np.fft.ifft(wz, axis=-1)), 0
    print('ifft', rows, p, e); ok &= e[0] < 1e-5
# padding / cropping
x = rng.standard_normal((5, 100000)).astype(np.float32)
e = rel_l2(dsc.rfft(dsc.from_numpy(x), n=131072).numpy(), np.fft.rfft(x.astype(np.float64), n=131072, axis=-1)); print('rfft padded', dsc.last_fft_path(), e); ok &= e < 1e-5
x = rng.standard_normal((5, 140000)).astype(np.float32)
e = rel_l2(dsc.rfft(dsc.from_numpy(x), n=131072).numpy(), np.fft.rfft(x.astype(np.float64), n=131072, axis=-1)); print('rfft cropped', dsc.last_fft_path(), e); ok &= e < 1e-5
Xs = (rng.standard_normal((4, 40000)) + 1j * rng.standard_normal((4, 40000))).astype(np.complex64)
e = rel_l2(dsc.irfft(dsc.from_numpy(Xs), n=65537).numpy(), np.fft.irfft(np.pad(Xs.astype(np.complex128), ((0, 0), (0, 65537 - Xs.shape[1]))), n=131072, axis=-1)); print('irfft short', dsc.last_fft_path(), e); ok &= e < 1e-5
z = (rng.standard_normal((4, 50000)) + 1j * rng.standard_normal((4, 50000))).astype(np.complex64)
e = rel_l2(dsc.fft(dsc.from_numpy(z), n=65536).numpy(), np.fft.fft(z.astype(np.complex128), n=65536, axis=-1)); print('fft padded', dsc.last_fft_path(), e); ok &= e < 1e-5
for rows in (2, 70):
    x = rng.standard_normal((rows, N))
    X = dsc.rfft(dsc.from_numpy(x)); p = dsc.last_fft_path(); want = np.fft.rfft(x.astype(np.float64), axis=-1)
    e = rel_l2(X.numpy(), want); print('f64 rfft', rows, p, e); ok &= e < 1e-14
    y = dsc.irfft(X); p = dsc.last_fft_path()
    e = rel_l2(y.numpy(), np.fft.irfft(want, axis=-1)); print('f64 irfft', rows, p, e); ok &= e < 1e-14
    z = rng.standard_normal((rows, 65536)) + 1j * rng.standard_normal((rows, 65536))
    Z = dsc.fft(dsc.from_numpy(z)); p = dsc.last_fft_path(); wz = np.fft.fft(z.astype(np.complex128), axis=-1)
    e = rel_l2(Z.numpy(), wz); print('f64 fft', rows, p, e); ok &= e < 1e-14
    e = rel_l2(dsc.ifft(Z).numpy(), np.fft.ifft(wz, axis=-1)); print('f64 ifft', rows, dsc.last_fft_path(), e); ok &= e < 1e-14
for rows in (1, 3, 40):
    x = rng.standard_normal((rows, 262144))
    X = dsc.rfft(dsc.from_numpy(x)); p = dsc.last_fft_path(); want = np.fft.rfft(x.astype(np.float64), axis=-1)
    e = rel_l2(X.numpy(), want); print('f64 rfft 262144', rows, p, e); ok &= e < 1e-14
    y = dsc.irfft(X); p = dsc.last_fft_path()
    e = rel_l2(y.numpy(), np.fft.irfft(want, axis=-1)); print('f64 irfft 262144', rows, p, e); ok &= e < 1e-14
    z = rng.standard_normal((rows, 131072)) + 1j * rng.standard_normal((rows, 131072))
    Z = dsc.fft(dsc.from_numpy(z)); p = dsc.last_fft_path(); wz = np.fft.fft(z.astype(np.complex128), axis=-1)
    e = rel_l2(Z.numpy(), wz); print('f64 fft 131072', rows, p, e); ok &= e < 1e-14
    e = rel_l2(dsc.ifft(Z).numpy(), np.fft.ifft(wz, axis=-1)); print('f64 ifft 131072', rows, dsc.last_fft_path(), e); ok &= e < 1e-14
x = rng.standard_normal((3, 200000))
e = rel_l2(dsc.rfft(dsc.from_numpy(x), n=262144).numpy(), np.fft.rfft(x, n=262144, axis=-1)); print('f64 rfft padded', dsc.last_fft_path(), e); ok &= e < 1e-14
Xs = rng.standard_normal((2, 100000)) + 1j * rng.standard_normal((2, 100000))
e = rel_l2(dsc.irfft(dsc.from_numpy(Xs), n=131073).numpy(), np.fft.irfft(np.pad(Xs, ((0, 0), (0, 131073 - Xs.shape[1]))), n=262144, axis=-1)); print('f64 irfft short', dsc.last_fft_path(), e); ok &= e < 1e-14
dsc.synchronize()
print('ALL OK' if ok else 'FAILED')
