"""tools/op_map.py — the elementwise / reduction / slicing operators of the path on typical and awkward shapes: ms and fraction of
the HBM roofline (algorithmic bytes: operands read once, result written once)."""
import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(10 << 30, 2 << 30)
D = dsc.Dtype
SZ = {D.F32: 4, D.F64: 8, D.C32: 8, D.C64: 16}


def timed(f, nbytes, label):
    for _ in range(3): r = f()
    dsc.synchronize()
    from dsc_amd import _bindings as B
    from dsc_amd.context import _get_ctx
    ctx = _get_ctx()
    B.dsc_timer_start(ctx)
    for _ in range(10): r = f()
    ms = B.dsc_timer_stop(ctx) / 10
    print(f'{label:70s} {ms:8.3f} ms  {nbytes / ms / 1e6 / 80:5.1f} %', flush=True)


def nb(*ts):
    return sum(int(np.prod(t.shape)) * SZ[t.dtype] for t in ts)


A = dsc.empty((4096, 32769), D.C32)
out = dsc.empty((4096, 32769), D.C32)
for shape, name in (((4096, 32769), 'same shape'), ((32769,), 'row [K]'), ((4096, 1), 'column [B,1]'), ((1,), 'one element')):
    Bt = dsc.empty(shape, D.C32)
    timed(lambda: dsc.mul(A, Bt, out=out), nb(A, Bt, out), f'mul c32 [4096,32769] x {name}')
timed(lambda: dsc.mul(A, 2.5, out=out), nb(A, out), 'mul c32 x python scalar')
Bf = dsc.empty((4096, 32769), D.F32)
timed(lambda: dsc.mul(A, Bf, out=out), nb(A, Bf, out), 'mul c32 x f32 same shape (promotion)')
Bd = dsc.empty((32769,), D.F64)
timed(lambda: dsc.mul(A, Bd, out=out), nb(A, Bd, out), 'mul c32 x f64 row (promotion to c32)')
X4 = dsc.empty((64, 8, 256, 512), D.F32)
Y4 = dsc.empty((64, 1, 256, 1), D.F32)
O4 = dsc.empty((64, 8, 256, 512), D.F32)
timed(lambda: dsc.add(X4, Y4, out=O4), nb(X4, Y4, O4), 'add f32 [64,8,256,512] + [64,1,256,1]')
Z4 = dsc.empty((8, 1, 512), D.F32)
W4 = dsc.empty((64, 1, 256, 1), D.F32)
O5 = dsc.empty((64, 8, 256, 512), D.F32)
timed(lambda: dsc.add(Z4, W4, out=O5), nb(Z4, W4, O5), 'add f32 [8,1,512] + [64,1,256,1] -> [64,8,256,512]')
timed(lambda: dsc.true_div(A, A, out=out), nb(A, A, out), 'div c32 same shape')
for name, f in (('abs', dsc.absolute), ('angle', dsc.angle), ('conj', dsc.conj), ('real', dsc.real), ('imag', dsc.imag)):
    r = f(A)
    timed(lambda: f(A), nb(A, r), f'{name} c32 [4096,32769]')
timed(lambda: A.cast(D.C64), nb(A) * 3, 'cast c32 -> c64')
timed(lambda: Bf.cast(D.C32), nb(Bf) * 3, 'cast f32 -> c32')
for axis in (0, 1):
    for name, f in (('sum', dsc.sum), ('mean', dsc.mean), ('max', dsc.max), ('min', dsc.min)):
        r = f(A, axis=axis)
        timed(lambda: f(A, axis=axis), nb(A, r), f'{name} c32 [4096,32769] axis {axis}')
T3 = dsc.empty((256, 512, 1024), D.F32)
for axis in (0, 1, 2):
    r = dsc.sum(T3, axis=axis)
    timed(lambda: dsc.sum(T3, axis=axis), nb(T3, r), f'sum f32 [256,512,1024] axis {axis}')
r = dsc.transpose(T3)
timed(lambda: dsc.transpose(T3), nb(T3) * 2, 'transpose f32 [256,512,1024] (reverse axes)')
timed(lambda: dsc.transpose(T3, (0, 2, 1)), nb(T3) * 2, 'transpose f32 [256,512,1024] axes (0,2,1)')
S = dsc.empty((4096, 65536), D.F32)
r = S[:, :60000]
timed(lambda: S[:, :60000], nb(r) * 2, 'slice f32 [4096,65536][:, :60000]')
r = S[::2]
timed(lambda: S[::2], nb(r) * 2, 'slice f32 [4096,65536][::2]')
r = S[:, ::2]
timed(lambda: S[:, ::2], nb(r) * 2, 'slice f32 [4096,65536][:, ::2]')
r = S[100]
timed(lambda: S[100], nb(r) * 2, 'index f32 [4096,65536][100]')
