import numpy as np, sys
sys.path.insert(0, '.')
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
from oracle import port
dsc.init(12 << 30, 2 << 30)
rng = np.random.default_rng(5)
lb = 537
tt = np.arange(lb) - (lb - 1) / 2
b = np.zeros(65536, np.float32); b[:lb] = (np.sinc(0.2 * tt) * np.hamming(lb) * 0.2).astype(np.float32)
Hh = port.rfft(b)
Hh[0] += 0.25j; Hh[-1] -= 0.5j          # complex DC / Nyquist taps: irfft must ignore the imaginary part of the PRODUCT there
H = dsc.from_numpy(Hh)
for rows in (1, 5, 300):
    s = rng.standard_normal((rows, 65536)).astype(np.float32)
    y = dsc.filter_fft(dsc.from_numpy(s), H)
    got = y.numpy(); path = dsc.last_fft_path()
    want = port.irfft(port.mul(port.rfft(s[:6]), Hh))
    err = np.linalg.norm(got[:6] - want) / np.linalg.norm(want)
    comp = dsc.irfft(dsc.rfft(dsc.from_numpy(s)) * H).numpy()
    err2 = np.linalg.norm(got - comp) / np.linalg.norm(comp)
    print(f'filter rows={rows} path={path} rel-L2 vs oracle composition {err:.3e}, vs GPU 3-op composition {err2:.3e}')
ctx = _get_ctx()
Bn = 4096
x = dsc.empty((Bn, 65536), dsc.Dtype.F32); out = dsc.empty((Bn, 65536), dsc.Dtype.F32)
for _ in range(3): B.dsc_filter_fft(ctx, x._c_ptr, H._c_ptr, out._c_ptr)
dsc.synchronize(); B.dsc_timer_start(ctx)
K = 20
for _ in range(K): B.dsc_filter_fft(ctx, x._c_ptr, H._c_ptr, out._c_ptr)
ms = B.dsc_timer_stop(ctx) / K
print(f'fused filter B={Bn}: {ms:.3f} ms/launch, {Bn*65536/ms/1e6:.1f} GSamples/s, {Bn*65536*8/ms/1e6:.1f} GB/s algorithmic = {Bn*65536*8/ms/1e6/80:.1f}% of 8 TB/s  path={dsc.last_fft_path()}')
S = dsc.empty((Bn, 32769), dsc.Dtype.C32); P = dsc.empty((Bn, 32769), dsc.Dtype.C32)
B.dsc_timer_start(ctx)
for _ in range(K):
    B.dsc_rfft(ctx, x._c_ptr, S._c_ptr, -1, -1); B.dsc_mul(ctx, S._c_ptr, H._c_ptr, P._c_ptr); B.dsc_irfft(ctx, P._c_ptr, out._c_ptr, -1, -1)
ms3 = B.dsc_timer_stop(ctx) / K
print(f'3-op composition B={Bn}: {ms3:.3f} ms  (fused is {ms3/ms:.2f}x faster)')
