import sys, os
sys.path.insert(0, '.')
fd = os.dup(1); os.dup2(2, 1)
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(6 << 30, 1 << 28)
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
f = (lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, -1)) if op == 'rfft' else (lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, -1))
for _ in range(5): f()
dsc.synchronize()
best = 1e9; tot = 0
for rep in range(5):
    B.dsc_timer_start(ctx)
    for _ in range(20): f()
    ms = B.dsc_timer_stop(ctx) / 20
    best = min(best, ms); tot += ms
print(f'{op} mean {tot/5:.4f} ms  best {best:.4f} ms  -> {Bn*524296/best/1e6/80:.2f} % of 8 TB/s (best)')
