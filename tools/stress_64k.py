"""tools/stress_64k.py — every element of every row, repeatedly: the 65536-point f32 kernels, the mid-size f64 kernel and the f64 two-pass
kernels against numpy (float64).  A rare wrong store (a handful of elements in one row) is invisible to an L2-norm check."""
import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(10 << 30, 2 << 30)
rng = np.random.default_rng(21)
bad = 0


def check(name, got, want, tol):
    global bad
    scale = np.max(np.abs(want), axis=1, keepdims=True)
    err = np.max(np.abs(got - want) / scale, axis=1)
    rows_bad = np.nonzero(err > tol)[0]
    if len(rows_bad):
        bad += 1
        print(name, 'bad rows', rows_bad[:10].tolist(), 'worst', float(err.max()), dsc.last_fft_path(), flush=True)


rows = 1024
x = rng.standard_normal((rows, 65536)).astype(np.float32)
wr = np.fft.rfft(x.astype(np.float64), axis=-1)
H = (rng.standard_normal(32769) + 1j * rng.standard_normal(32769)).astype(np.complex64)
wf = np.fft.irfft(wr * H.astype(np.complex128), axis=-1)
tx, tX, tH = dsc.from_numpy(x), dsc.from_numpy(wr.astype(np.complex64)), dsc.from_numpy(H)
z = (rng.standard_normal((rows, 32768)) + 1j * rng.standard_normal((rows, 32768))).astype(np.complex64)
wz = np.fft.fft(z.astype(np.complex128), axis=-1)
tz = dsc.from_numpy(z)
for rep in range(8):
    check('rfft64k', dsc.rfft(tx).numpy(), wr, 2e-5)
    check('irfft64k', dsc.irfft(tX).numpy(), x.astype(np.float64), 2e-5)
    check('filter64k', dsc.filter_fft(tx, tH).numpy(), wf, 5e-5)
    check('c2c32k', dsc.fft(tz).numpy(), wz, 2e-5)
del tx, tX, tz
# f64: mid-size register kernel (2 waves per SIMD, 128-bit stores) and the two-pass kernels
for n, rows in ((4096, 4096), (1024, 8192), (8192, 2048), (16384, 1024), (32768, 512), (65536, 256), (524288, 24)):
    xd = rng.standard_normal((rows, n))
    wd = np.fft.rfft(xd, axis=-1)
    td, tD = dsc.from_numpy(xd), dsc.from_numpy(wd)
    for rep in range(5):
        check(f'rfft f64 {n}', dsc.rfft(td).numpy(), wd, 1e-12)
        check(f'irfft f64 {n}', dsc.irfft(tD).numpy(), xd, 1e-12)
    zd = rng.standard_normal((rows, n // 2)) + 1j * rng.standard_normal((rows, n // 2))
    wzd = np.fft.fft(zd, axis=-1)
    tzd = dsc.from_numpy(zd)
    for rep in range(5):
        check(f'fft c64 {n // 2}', dsc.fft(tzd).numpy(), wzd, 1e-12)
        check(f'ifft c64 {n // 2}', dsc.ifft(tzd).numpy(), np.fft.ifft(zd, axis=-1), 1e-12)
    del td, tD, tzd
# f64 along axis 0 (column kernel, 128-bit stores)
for n, cols in ((256, 4096), (1024, 2048)):
    xd = rng.standard_normal((n, cols))
    zd = xd + 1j * rng.standard_normal((n, cols))
    td, tzd = dsc.from_numpy(xd), dsc.from_numpy(zd)
    wd, wzd = np.fft.rfft(xd, axis=0), np.fft.fft(zd, axis=0)
    for rep in range(5):
        check(f'rfft f64 axis0 {n}', dsc.rfft(td, axis=0).numpy().T, wd.T, 1e-12)
        check(f'fft c64 axis0 {n}', dsc.fft(tzd, axis=0).numpy().T, wzd.T, 1e-12)
print('STRESS64K', 'FAILED' if bad else 'OK', bad)
