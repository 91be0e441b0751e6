import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(12 << 30, 2 << 30)
ctx = _get_ctx()
for n, rows in ((1024, 65536), (4096, 16384), (32768, 4096), (65536, 2048), (131072, 1024), (262144, 512)):
    x = dsc.from_numpy(np.random.default_rng(1).standard_normal((rows, n)).astype(np.float32))
    X = dsc.empty((rows, n), dsc.Dtype.C32)
    f = lambda: B.dsc_fft(ctx, x._c_ptr, X._c_ptr, -1, -1)
    for _ in range(3): f()
    dsc.synchronize()
    B.dsc_timer_start(ctx)
    for _ in range(10): f()
    ms = B.dsc_timer_stop(ctx) / 10
    nbytes = rows * n * (4 + 8)
    print(f'fft(real f32) n={n} rows={rows}: {ms:.3f} ms path={dsc.last_fft_path()} {nbytes/ms/1e6/80:.1f}% of 8 TB/s', flush=True)
