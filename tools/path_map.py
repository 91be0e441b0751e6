"""tools/path_map.py — which kernel family serves which (operator, dtype, length) on contiguous rows, and at what fraction of the HBM
roofline (algorithmic bytes, 256 MiB of input per case): the map that shows what still falls to the generic paths."""
import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
from dsc_amd import _bindings as B
from dsc_amd.context import _get_ctx
dsc.init(8 << 30, 2 << 30)
ctx = _get_ctx()
D = dsc.Dtype
SZ = {D.F32: 4, D.F64: 8, D.C32: 8, D.C64: 16}


def run(op, dt, n, pad=False):
    in_dt = dt
    if op == 'rfft':
        out_dt = D.C32 if dt == D.F32 else D.C64; in_n, out_n = n, n // 2 + 1
    elif op == 'irfft':
        in_dt = D.C32 if dt == D.F32 else D.C64; out_dt = dt; in_n, out_n = n // 2 + 1, n
    else:
        out_dt = D.C32 if dt in (D.F32, D.C32) else D.C64; in_n, out_n = n, n
    if pad:
        in_n = max(1, in_n - 3)
    rows = max(1, (128 << 20) // (in_n * SZ[in_dt]))
    x = dsc.empty((rows, in_n), in_dt)
    y = dsc.empty((rows, out_n), out_dt)
    f = getattr(B, 'dsc_' + op)
    narg = (n // 2 + 1) if op == 'irfft' else n
    call = lambda: f(ctx, x._c_ptr, y._c_ptr, narg if pad else -1, -1)
    for _ in range(2): call()
    dsc.synchronize()
    B.dsc_timer_start(ctx)
    for _ in range(5): call()
    ms = B.dsc_timer_stop(ctx) / 5
    nbytes = rows * (in_n * SZ[in_dt] + out_n * SZ[out_dt])
    return dsc.last_fft_path(), nbytes / ms / 1e6 / 80


names = {D.F32: 'f32', D.F64: 'f64', D.C32: 'c32', D.C64: 'c64'}
for op, dts in () if (len(sys.argv) > 1 and sys.argv[1] == 'axis0') else (('rfft', (D.F32, D.F64)), ('irfft', (D.F32, D.F64)), ('fft', (D.C32, D.C64, D.F32, D.F64)), ('ifft', (D.C32, D.F32))):
    for dt in dts:
        for pad in (False, True):
            cells = []
            for lg in range(1, 21):
                n = 1 << lg
                if op in ('rfft', 'irfft') and n < 4:
                    continue
                try:
                    p, frac = run(op, dt, n, pad)
                    cells.append(f'{n}:{p}:{frac:.0f}')
                except Exception as e:
                    cells.append(f'{n}:ERR')
            print(op, names[dt], 'padded' if pad else 'full', ' '.join(cells), flush=True)


def run0(op, dt, n):
    """the same along axis 0 of a [n, cols] tensor (strided lines)"""
    in_dt = dt
    if op == 'rfft':
        out_dt = D.C32 if dt == D.F32 else D.C64; in_n, out_n = n, n // 2 + 1
    elif op == 'irfft':
        in_dt = D.C32 if dt == D.F32 else D.C64; out_dt = dt; in_n, out_n = n // 2 + 1, n
    else:
        out_dt = D.C32 if dt in (D.F32, D.C32) else D.C64; in_n, out_n = n, n
    cols = max(8, (128 << 20) // (in_n * SZ[in_dt]))
    x = dsc.empty((in_n, cols), in_dt)
    y = dsc.empty((out_n, cols), out_dt)
    f = getattr(B, 'dsc_' + op)
    call = lambda: f(ctx, x._c_ptr, y._c_ptr, -1, 0)
    for _ in range(2): call()
    dsc.synchronize()
    B.dsc_timer_start(ctx)
    for _ in range(5): call()
    ms = B.dsc_timer_stop(ctx) / 5
    return dsc.last_fft_path(), cols * (in_n * SZ[in_dt] + out_n * SZ[out_dt]) / ms / 1e6 / 80


if len(sys.argv) > 1 and sys.argv[1] == 'axis0':
    for op, dts in (('rfft', (D.F32, D.F64)), ('irfft', (D.F32,)), ('fft', (D.C32, D.C64, D.F32)), ('ifft', (D.C32,))):
        for dt in dts:
            cells = []
            for lg in range(2, 18):
                n = 1 << lg
                try:
                    p, frac = run0(op, dt, n)
                    cells.append(f'{n}:{p}:{frac:.0f}')
                except Exception as e:
                    cells.append(f'{n}:ERR')
            print('axis0', op, names[dt], ' '.join(cells), flush=True)
