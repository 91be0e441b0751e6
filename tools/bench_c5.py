import sys
sys.path.insert(0, '.')
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(20 << 30, 12 << 30)
ctx = _get_ctx()
N5 = 262144
x5 = dsc.empty((2048, N5), dsc.Dtype.F64); X5 = dsc.empty((2048, N5 // 2 + 1), dsc.Dtype.C64)
bytes5 = 2048 * (N5 * 8 + (N5 // 2 + 1) * 16)
for f, name in ((lambda: B.dsc_rfft(ctx, x5._c_ptr, X5._c_ptr, -1, -1), 'rfft'), (lambda: B.dsc_irfft(ctx, X5._c_ptr, x5._c_ptr, -1, -1), 'irfft')):
    for _ in range(4): f()
    dsc.synchronize()
    B.dsc_timer_start(ctx)
    for _ in range(6): f()
    ms = B.dsc_timer_stop(ctx) / 6
    print(f'C5 {name} f64 N=262144 B=2048: {ms:.3f} ms {bytes5/ms/1e6:.0f} GB/s {bytes5/ms/1e6/80:.1f}% of 8 TB/s')
