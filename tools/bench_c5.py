"""tools/bench_c5.py — BASELINE config 5 (rfft / irfft f64 N=262144 B=2048) on random data; DSC_2PASS_CHUNK_ROWS selects the rows
per launch pair (the intermediate of a chunk is 2 MiB per row).  One line per direction."""
import os
import sys
sys.path.insert(0, '.')
fd = os.dup(1); os.dup2(2, 1)
import numpy as np
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(20 << 30, 6 << 30)
os.dup2(fd, 1)
ctx = _get_ctx()
N5 = 262144
rng = np.random.default_rng(99)
blk = rng.standard_normal((64, N5))
x5 = dsc.from_numpy(np.tile(blk, (32, 1)))
X5 = dsc.empty((2048, N5 // 2 + 1), dsc.Dtype.C64)
bytes5 = 2048 * (N5 * 8 + (N5 // 2 + 1) * 16)
B.dsc_rfft(ctx, x5._c_ptr, X5._c_ptr, -1, -1)
tag = f"chunk_rows={os.environ.get('DSC_2PASS_CHUNK_ROWS', 'all')} lib={os.path.basename(os.environ.get('DSC_MI355X_LIB', 'default'))}"
for f, name in ((lambda: B.dsc_rfft(ctx, x5._c_ptr, X5._c_ptr, -1, -1), 'rfft'), (lambda: B.dsc_irfft(ctx, X5._c_ptr, x5._c_ptr, -1, -1), 'irfft')):
    for _ in range(6): f()
    dsc.synchronize()
    best = 1e9
    for _ in range(3):
        B.dsc_timer_start(ctx)
        for _ in range(8): f()
        best = min(best, B.dsc_timer_stop(ctx) / 8)
    print(f'C5 {name} f64 N=262144 B=2048 [{tag}]: {best:.3f} ms {bytes5/best/1e6:.0f} GB/s {bytes5/best/1e6/80:.1f}% of 8 TB/s', flush=True)
