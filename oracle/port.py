"""ctypes front-end to oracle/liboracle.so (the C restatement in dsc_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by dsc_amd/.  Works on numpy arrays.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_int, c_size_t, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboracle.so')

F32, F64, C32, C64 = 0, 1, 2, 3
NP_TO_DT = {np.dtype(np.float32): F32, np.dtype(np.float64): F64,
            np.dtype(np.complex64): C32, np.dtype(np.complex128): C64}
DT_TO_NP = {v: k for k, v in NP_TO_DT.items()}
SUM, MEAN, MAX, MIN = 0, 1, 2, 3


class OrcTensor(Structure):
    _fields_ = [('shape', c_int * 4), ('stride', c_int * 4), ('data', c_void_p),
                ('ne', c_int), ('n_dim', c_int), ('dtype', c_int)]


def build():
    """(Re)build liboracle.so with the committed Makefile (gcc, a second or two)."""
    subprocess.check_call(['make', '-s', '-C', _HERE, 'port'])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        P = POINTER(OrcTensor)
        I4 = POINTER(c_int)
        _lib.orc_tensor_init.argtypes = [P, c_int, I4, c_int, c_void_p]
        _lib.orc_fft_out_shape.argtypes = [P, c_int, c_int, I4, I4]
        _lib.orc_rfft_out_shape.argtypes = [P, c_int, c_int, c_int, I4, I4]
        for name in ('orc_fft', 'orc_ifft', 'orc_rfft', 'orc_irfft'):
            getattr(_lib, name).argtypes = [P, P, c_int, c_int]
        _lib.orc_mul_out_shape.argtypes = [P, P, I4, I4, I4]
        _lib.orc_mul.argtypes = [P, P, P]
        _lib.orc_binary.argtypes = [P, P, P, c_int]
        _lib.orc_unary.argtypes = [P, P, c_int]
        _lib.orc_unary_out_dtype.argtypes = [c_int, c_int]
        _lib.orc_reduce_out_shape.argtypes = [P, c_int, c_int, I4, I4]
        _lib.orc_reduce.argtypes = [P, P, c_int, c_int]
        _lib.orc_cast.argtypes = [P, P]
        _lib.orc_fft_storage.argtypes = [c_int, c_int, c_int]
        _lib.orc_fft_storage.restype = c_size_t
        _lib.orc_init_plan.argtypes = [c_void_p, c_int, c_int, c_int]
    return _lib


def _wrap(a):
    a = np.ascontiguousarray(a)
    if a.ndim == 0:
        a = a.reshape(1)
    t = OrcTensor()
    shape = (c_int * 4)(*a.shape)
    lib().orc_tensor_init(ctypes.byref(t), a.ndim, shape, NP_TO_DT[a.dtype], a.ctypes.data)
    return t, a


def _alloc(shape4, n_dim, dtype):
    shape = tuple(shape4[4 - n_dim:]) if n_dim > 0 else (1,)
    out = np.empty(shape, dtype=DT_TO_NP[dtype])
    t, out = _wrap(out)
    return t, out


def _fft_like(name, x, n, axis, rfft_forward=None):
    tx, x = _wrap(x)
    shape = (c_int * 4)()
    dt = c_int()
    if rfft_forward is None:
        rc = lib().orc_fft_out_shape(ctypes.byref(tx), n, axis, shape, ctypes.byref(dt))
    else:
        rc = lib().orc_rfft_out_shape(ctypes.byref(tx), n, axis, int(rfft_forward), shape, ctypes.byref(dt))
    if rc:
        raise ValueError(f'{name}: the reference aborts on this input (dtype/axis)')
    to, out = _alloc(list(shape), x.ndim, dt.value)
    rc = getattr(lib(), name)(ctypes.byref(tx), ctypes.byref(to), n, axis)
    assert rc == 0
    return out


def fft(x, n=-1, axis=-1):
    return _fft_like('orc_fft', x, n, axis)


def ifft(x, n=-1, axis=-1):
    return _fft_like('orc_ifft', x, n, axis)


def rfft(x, n=-1, axis=-1):
    return _fft_like('orc_rfft', x, n, axis, rfft_forward=True)


def irfft(x, n=-1, axis=-1):
    return _fft_like('orc_irfft', x, n, axis, rfft_forward=False)


ADD, SUB, MUL, DIV = 0, 1, 2, 3


def binary(a, b, op):
    ta, a = _wrap(a)
    tb, b = _wrap(b)
    shape = (c_int * 4)()
    nd, dt = c_int(), c_int()
    if lib().orc_mul_out_shape(ctypes.byref(ta), ctypes.byref(tb), shape, ctypes.byref(nd), ctypes.byref(dt)):
        raise ValueError('binary: shapes do not broadcast')
    to, out = _alloc(list(shape), nd.value, dt.value)
    assert lib().orc_binary(ctypes.byref(ta), ctypes.byref(tb), ctypes.byref(to), op) == 0
    return out


def mul(a, b):
    ta, a = _wrap(a)
    tb, b = _wrap(b)
    shape = (c_int * 4)()
    nd, dt = c_int(), c_int()
    if lib().orc_mul_out_shape(ctypes.byref(ta), ctypes.byref(tb), shape, ctypes.byref(nd), ctypes.byref(dt)):
        raise ValueError('mul: shapes do not broadcast')
    to, out = _alloc(list(shape), nd.value, dt.value)
    assert lib().orc_mul(ctypes.byref(ta), ctypes.byref(tb), ctypes.byref(to)) == 0
    return out


ABS, ANGLE, CONJ, REALPART, IMAGPART = 0, 1, 2, 3, 4


def unary(x, op):
    tx, x = _wrap(x)
    dt = lib().orc_unary_out_dtype(NP_TO_DT[x.dtype], op)
    to, out = _alloc([1] * (4 - x.ndim) + list(x.shape), x.ndim, dt)
    assert lib().orc_unary(ctypes.byref(tx), ctypes.byref(to), op) == 0
    return out


def reduce(x, op, axis=-1, keepdims=True):
    tx, x = _wrap(x)
    shape = (c_int * 4)()
    nd = c_int()
    if lib().orc_reduce_out_shape(ctypes.byref(tx), axis, int(keepdims), shape, ctypes.byref(nd)):
        raise ValueError('reduce: bad axis')
    to, out = _alloc(list(shape), nd.value, NP_TO_DT[x.dtype])
    assert lib().orc_reduce(ctypes.byref(tx), ctypes.byref(to), axis, op) == 0
    return out


def cast(x, dtype):
    tx, x = _wrap(x)
    to, out = _alloc([1] * (4 - x.ndim) + list(x.shape), x.ndim, NP_TO_DT[np.dtype(dtype)])
    lib().orc_cast(ctypes.byref(tx), ctypes.byref(to))
    return out


def plan_table(n, dtype, fft_type):
    """Twiddle table of a plan as a flat real array (fft_type 0 REAL, 1 COMPLEX)."""
    dt = NP_TO_DT[np.dtype(dtype)]
    nbytes = lib().orc_fft_storage(n, dt, fft_type)
    real = np.float32 if dt in (F32, C32) else np.float64
    tw = np.empty(nbytes // np.dtype(real).itemsize, dtype=real)
    lib().orc_init_plan(tw.ctypes.data, n, dt, fft_type)
    return tw
