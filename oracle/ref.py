"""ctypes front-end to oracle/_ref/libdsc_ref.so — the reference ITSELF, compiled by
oracle/Makefile (`make ref`) from the sources where they lie under /root/reference.

TEST INFRASTRUCTURE ONLY.  Used to pin the C restatement (tests/test_oracle_vs_ref.py),
to generate tests/golden/ (tests/golden/make_golden.py) and, on the GPU box, as the
"reference" CPU baseline in bench.py.  The .so is git-ignored but travels with gpurun;
the reference's sources never leave /root/reference.

The binding below is ours (written against the C ABI in dsc/include/dsc.h:85-428); it
does not import the reference's Python package.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_bool, c_int, c_size_t, c_uint8, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_ref', 'libdsc_ref.so')

F32, F64, C32, C64 = 0, 1, 2, 3
NP_TO_DT = {np.dtype(np.float32): F32, np.dtype(np.float64): F64,
            np.dtype(np.complex64): C32, np.dtype(np.complex128): C64}
DT_TO_NP = {v: k for k, v in NP_TO_DT.items()}


class RefTensor(Structure):            # dsc/include/dsc.h:96-108
    _fields_ = [('shape', c_int * 4), ('stride', c_int * 4), ('buffer', c_void_p),
                ('data', c_void_p), ('ne', c_int), ('n_dim', c_int),
                ('dtype', c_uint8), ('backend', c_uint8)]


TP = POINTER(RefTensor)


def available():
    return os.path.exists(LIB_PATH)


class Ref:
    """One reference context (the reference allows one live context per process:
    its allocators are function-local statics, dsc/src/dsc_allocator.cpp:212,295)."""

    _instance = None

    @classmethod
    def get(cls, main_mem=1 << 30, scratch_mem=1 << 28):
        if cls._instance is None:
            cls._instance = cls(main_mem, scratch_mem)
        return cls._instance

    def __init__(self, main_mem, scratch_mem):
        if not available():
            raise RuntimeError(f'{LIB_PATH} missing: run `make -C oracle ref` where /root/reference exists')
        L = self.L = ctypes.CDLL(LIB_PATH)
        L.dsc_ctx_init.argtypes = [c_size_t, c_size_t]
        L.dsc_ctx_init.restype = c_void_p
        L.dsc_ctx_clear.argtypes = [c_void_p]
        L.dsc_tensor_free.argtypes = [c_void_p, TP]
        for nd in range(1, 5):
            f = getattr(L, f'dsc_tensor_{nd}d')
            f.argtypes = [c_void_p, c_uint8] + [c_int] * nd
            f.restype = TP
        for name in ('dsc_fft', 'dsc_ifft', 'dsc_rfft', 'dsc_irfft'):
            f = getattr(L, name)
            f.argtypes = [c_void_p, TP, TP, c_int, c_int]
            f.restype = TP
        for name in ('dsc_add', 'dsc_sub', 'dsc_mul', 'dsc_div'):
            f = getattr(L, name)
            f.argtypes = [c_void_p, TP, TP, TP]
            f.restype = TP
        for name in ('dsc_sum', 'dsc_mean', 'dsc_max', 'dsc_min'):
            f = getattr(L, name)
            f.argtypes = [c_void_p, TP, TP, c_int, c_bool]
            f.restype = TP
        L.dsc_abs.argtypes = [c_void_p, TP, TP]
        L.dsc_abs.restype = TP
        for name in ('dsc_angle', 'dsc_conj', 'dsc_real', 'dsc_imag'):
            f = getattr(L, name)
            f.argtypes = [c_void_p, TP]
            f.restype = TP
        L.dsc_cast.argtypes = [c_void_p, TP, c_uint8]
        L.dsc_cast.restype = TP
        self.ctx = L.dsc_ctx_init(main_mem, scratch_mem)

    # -- host <-> arena ------------------------------------------------------
    def put(self, a):
        a = np.ascontiguousarray(a)
        if a.ndim == 0:
            a = a.reshape(1)
        t = getattr(self.L, f'dsc_tensor_{a.ndim}d')(self.ctx, NP_TO_DT[a.dtype], *a.shape)
        ctypes.memmove(t.contents.data, a.ctypes.data, a.nbytes)
        return t

    def take(self, t, free=True):
        c = t.contents
        shape = tuple(c.shape[4 - c.n_dim:]) if c.n_dim > 0 else (1,)
        out = np.empty(shape, dtype=DT_TO_NP[c.dtype])
        ctypes.memmove(out.ctypes.data, c.data, out.nbytes)
        if free:
            self.L.dsc_tensor_free(self.ctx, t)
        return out

    def free(self, *ts):
        for t in ts:
            self.L.dsc_tensor_free(self.ctx, t)

    # -- ops on numpy arrays ---------------------------------------------------
    def _fft(self, name, x, n, axis):
        tx = self.put(x)
        out = self.take(getattr(self.L, name)(self.ctx, tx, None, n, axis))
        self.free(tx)
        return out

    def fft(self, x, n=-1, axis=-1):
        return self._fft('dsc_fft', x, n, axis)

    def ifft(self, x, n=-1, axis=-1):
        return self._fft('dsc_ifft', x, n, axis)

    def rfft(self, x, n=-1, axis=-1):
        return self._fft('dsc_rfft', x, n, axis)

    def irfft(self, x, n=-1, axis=-1):
        return self._fft('dsc_irfft', x, n, axis)

    def mul(self, a, b):
        ta, tb = self.put(a), self.put(b)
        out = self.take(self.L.dsc_mul(self.ctx, ta, tb, None))
        self.free(ta, tb)
        return out

    def binary(self, a, b, op):
        ta, tb = self.put(a), self.put(b)
        f = getattr(self.L, 'dsc_' + ('add', 'sub', 'mul', 'div')[op])
        out = self.take(f(self.ctx, ta, tb, None))
        self.free(ta, tb)
        return out

    def unary(self, x, op):
        tx = self.put(x)
        if op == 0:
            to = self.L.dsc_abs(self.ctx, tx, None)
        else:
            to = getattr(self.L, 'dsc_' + ('', 'angle', 'conj', 'real', 'imag')[op])(self.ctx, tx)
        same = ctypes.cast(to, c_void_p).value == ctypes.cast(tx, c_void_p).value     # conj / real of a real tensor return x
        out = self.take(to, free=not same)
        self.free(tx)
        return out

    def reduce(self, x, op, axis=-1, keepdims=True):
        tx = self.put(x)
        f = getattr(self.L, 'dsc_' + ('sum', 'mean', 'max', 'min')[op])
        out = self.take(f(self.ctx, tx, None, axis, keepdims))
        self.free(tx)
        return out

    def cast(self, x, dtype):
        tx = self.put(x)
        dt = NP_TO_DT[np.dtype(dtype)]
        to = self.L.dsc_cast(self.ctx, tx, dt)
        same = ctypes.cast(to, c_void_p).value == ctypes.cast(tx, c_void_p).value
        out = self.take(to, free=not same)
        self.free(tx)
        return out

    # -- indexing / slicing (variadic C functions: dsc.h:244-260) --------------------
    NONE = 2 ** 31 - 1

    class Slice(Structure):
        _fields_ = [('start', c_int), ('stop', c_int), ('step', c_int)]

    @classmethod
    def c_slice(cls, s):
        if isinstance(s, slice):
            f = lambda i: cls.NONE if i is None else int(i)      # noqa: E731
            return cls.Slice(f(s.start), f(s.stop), f(s.step))
        return cls.Slice(int(s), int(s), int(s))

    def _setup_index(self):
        L = self.L
        if getattr(self, '_idx_ready', False):
            return
        L.dsc_tensor_get_idx.argtypes = [c_void_p, TP, c_int]
        L.dsc_tensor_get_idx.restype = TP
        L.dsc_tensor_get_slice.argtypes = [c_void_p, TP, c_int]
        L.dsc_tensor_get_slice.restype = TP
        L.dsc_tensor_set_idx.argtypes = [c_void_p, TP, TP, c_int]
        L.dsc_tensor_set_idx.restype = None
        L.dsc_tensor_set_slice.argtypes = [c_void_p, TP, TP, c_int]
        L.dsc_tensor_set_slice.restype = None
        self._idx_ready = True

    def get_idx(self, x, *idx):
        self._setup_index()
        tx = self.put(x)
        out = self.take(self.L.dsc_tensor_get_idx(self.ctx, tx, len(idx), *[c_int(i) for i in idx]))
        self.free(tx)
        return out

    def get_slice(self, x, *key):
        self._setup_index()
        tx = self.put(x)
        out = self.take(self.L.dsc_tensor_get_slice(self.ctx, tx, len(key), *[self.c_slice(k) for k in key]))
        self.free(tx)
        return out

    def set_idx(self, xa, xb, *idx):
        self._setup_index()
        ta, tb = self.put(xa), self.put(xb)
        self.L.dsc_tensor_set_idx(self.ctx, ta, tb, len(idx), *[c_int(i) for i in idx])
        out = self.take(ta)
        self.free(tb)
        return out

    def set_slice(self, xa, xb, *key):
        self._setup_index()
        ta, tb = self.put(xa), self.put(xb)
        self.L.dsc_tensor_set_slice(self.ctx, ta, tb, len(key), *[self.c_slice(k) for k in key])
        out = self.take(ta)
        self.free(tb)
        return out

    def transpose(self, x, axes=None):
        from ctypes import c_double  # noqa: F401
        L = self.L
        L.dsc_transpose.argtypes = [c_void_p, TP, c_int]
        L.dsc_transpose.restype = TP
        tx = self.put(x)
        axes = tuple(axes) if axes is not None else ()
        to = L.dsc_transpose(self.ctx, tx, len(axes), *[c_int(a) for a in axes])
        out = self.take(to)
        self.free(tx)
        return out

    def fftfreq(self, n, d, dtype, real_bins=False):
        from ctypes import c_double
        f = self.L.dsc_rfftfreq if real_bins else self.L.dsc_fftfreq
        f.argtypes = [c_void_p, c_int, c_double, c_uint8]
        f.restype = TP
        return self.take(f(self.ctx, n, d, NP_TO_DT[np.dtype(dtype)]))

    # -- raw handles, for timing loops in bench.py -----------------------------
    def rfft_raw(self, tx, tout):
        return self.L.dsc_rfft(self.ctx, tx, tout, -1, -1)
