import sys, os
sys.path.insert(0, '.')
fd = os.dup(1); os.dup2(2, 1)
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(12 << 30, 1 << 28)
os.dup2(fd, 1)
ctx = _get_ctx()
op = sys.argv[1] if len(sys.argv) > 1 else 'rfft'
Bn = 8192
import numpy as np
rng = np.random.default_rng(1)
blk = rng.standard_normal((512, 65536), dtype=np.float32)
x = dsc.from_numpy(np.tile(blk, (Bn // 512, 1)))           # random data: clocks depend on the operands
X = dsc.empty((Bn, 32769), dsc.Dtype.C32)
B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, -1)                # a real spectrum for the inverse
if op == 'fft':                                             # 32768-point complex rows, same bytes per launch
    z = dsc.from_numpy(np.tile((blk[:, :32768] + 1j * blk[:, 32768:]).astype(np.complex64), (Bn // 512, 1)))
    Z = dsc.empty((Bn, 32768), dsc.Dtype.C32)
    f = lambda: B.dsc_fft(ctx, z._c_ptr, Z._c_ptr, -1, -1)
elif op == 'filter':                                        # fused filter on half the batch
    rng2 = np.random.default_rng(7)
    H = dsc.from_numpy((rng2.standard_normal(32769) + 1j * rng2.standard_normal(32769)).astype(np.complex64))
    s_half = dsc.from_numpy(np.tile(blk, (Bn // 1024, 1)))
    y_half = dsc.empty((Bn // 2, 65536), dsc.Dtype.F32)
    f = lambda: B.dsc_filter_fft(ctx, s_half._c_ptr, H._c_ptr, y_half._c_ptr)
else:
    f = (lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, -1)) if op == 'rfft' else (lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, -1))
for _ in range(5): f()
dsc.synchronize()
best = 1e9; tot = 0
for rep in range(5):
    B.dsc_timer_start(ctx)
    for _ in range(20): f()
    ms = B.dsc_timer_stop(ctx) / 20
    best = min(best, ms); tot += ms
nb = Bn * 524288 if op == 'fft' else Bn // 2 * 524288 if op == 'filter' else Bn * 524296
print(f'{op} mean {tot/5:.4f} ms  best {best:.4f} ms  -> {nb/best/1e6/80:.2f} % of 8 TB/s (best)')
