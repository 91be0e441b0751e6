import sys
sys.path.insert(0, '.')
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
n, b = int(sys.argv[1]), int(sys.argv[2])
dsc.init(8 << 30, 1 << 30)
ctx = _get_ctx()
xs = dsc.empty((b, n), dsc.Dtype.F32); Xs = dsc.empty((b, n // 2 + 1), dsc.Dtype.C32)
for _ in range(4):
    B.dsc_rfft(ctx, xs._c_ptr, Xs._c_ptr, -1, -1)
dsc.synchronize()
