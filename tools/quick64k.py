import numpy as np, time, sys
sys.path.insert(0, '.')
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
from oracle import port
dsc.init(12 << 30, 2 << 30)
rng = np.random.default_rng(0)
for rows in (1, 3, 300):
    x = rng.standard_normal((rows, 65536)).astype(np.float32)
    X = dsc.rfft(dsc.from_numpy(x))
    got = X.numpy(); path = dsc.last_fft_path()
    want = port.rfft(x[:8])
    err = np.linalg.norm(got[:8] - want) / np.linalg.norm(want)
    ref64 = np.fft.rfft(x.astype(np.float64), axis=-1)
    err64 = np.linalg.norm(got - ref64) / np.linalg.norm(ref64)
    worst = np.max(np.abs(got - ref64)) / np.max(np.abs(ref64))
    print(f'rows={rows} path={path} rel-L2 vs oracle {err:.3e} vs f64 {err64:.3e} max-rel {worst:.3e} DC imag {np.abs(got[:,0].imag).max()} Nyq imag {np.abs(got[:,-1].imag).max()}')
    if err > 1e-5:
        bad = np.argwhere(np.abs(got[0] - ref64[0]) > 1e-3 * np.max(np.abs(ref64[0])))
        print('bad bins', len(bad), bad[:20].ravel())
# timing
Bn = 8192
x = dsc.empty((Bn, 65536), dsc.Dtype.F32)
out = dsc.empty((Bn, 32769), dsc.Dtype.C32)
ctx = _get_ctx()
for _ in range(3): B.dsc_rfft(ctx, x._c_ptr, out._c_ptr, -1, -1)
dsc.synchronize()
B.dsc_timer_start(ctx)
K = 20
for _ in range(K): B.dsc_rfft(ctx, x._c_ptr, out._c_ptr, -1, -1)
ms = B.dsc_timer_stop(ctx) / K
bytes_ = Bn * (65536 * 4 + 32769 * 8)
print(f'rfft B={Bn}: {ms:.3f} ms/launch, {Bn*65536/ms/1e6:.1f} GSamples/s, {bytes_/ms/1e6:.1f} GB/s = {bytes_/ms/1e6/8000*100:.1f}% of 8 TB/s  path={dsc.last_fft_path()}')
# ---- inverse
for rows in (1, 3, 300):
    x = rng.standard_normal((rows, 65536)).astype(np.float32)
    Xh = port.rfft(x[:8]) if rows <= 8 else None
    Xfull = np.fft.rfft(x.astype(np.float64), axis=-1).astype(np.complex64)
    # garbage imaginary parts in bins 0 and M must be ignored (dsc_fft.h:227-228)
    Xfull[:, 0] += 3j; Xfull[:, -1] -= 2j
    y = dsc.irfft(dsc.from_numpy(Xfull))
    got = y.numpy(); path = dsc.last_fft_path()
    want = port.irfft(Xfull[:8])
    err = np.linalg.norm(got[:8] - want) / np.linalg.norm(want)
    err_x = np.linalg.norm(got - x) / np.linalg.norm(x)
    print(f'irfft rows={rows} path={path} rel-L2 vs oracle {err:.3e} vs original signal {err_x:.3e} max {np.max(np.abs(got-x)):.3e}')
Xd = dsc.empty((Bn, 32769), dsc.Dtype.C32)
xo = dsc.empty((Bn, 65536), dsc.Dtype.F32)
for _ in range(3): B.dsc_irfft(ctx, Xd._c_ptr, xo._c_ptr, -1, -1)
dsc.synchronize()
B.dsc_timer_start(ctx)
for _ in range(K): B.dsc_irfft(ctx, Xd._c_ptr, xo._c_ptr, -1, -1)
ms = B.dsc_timer_stop(ctx) / K
print(f'irfft B={Bn}: {ms:.3f} ms/launch, {Bn*65536/ms/1e6:.1f} GSamples/s, {bytes_/ms/1e6:.1f} GB/s = {bytes_/ms/1e6/8000*100:.1f}% of 8 TB/s  path={dsc.last_fft_path()}')
