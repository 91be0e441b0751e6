"""`dsc.Tensor` and the hot-path operators (mirror of python/dsc/tensor.py for that subset:
Tensor :159-331, from_numpy :371-377, mul :477-483, sum/mean/max/min :579-612,
fft/ifft/rfft/irfft :693-726).

Differences forced by the device arena (see include/dsc_mi355x.h): `numpy()` and
`from_numpy()` COPY through dsc_copy_to_host / dsc_copy_from_host instead of viewing /
memmoving the data pointer."""
from typing import List, Tuple, Union

import numpy as np

from . import _bindings as B
from .context import _current as _current_ctx
from .context import _get_ctx
from .dtype import DTYPE_CONVERSION_TABLES, DTYPE_TO_NP, NP_TO_DTYPE, Dtype, ScalarType

TensorType = Union['Tensor', np.ndarray]
_DSC_MAX_DIMS = 4


def _c_ptr_or_none(x):
    return x._c_ptr if x is not None else None


class Tensor:
    def __init__(self, c_ptr, view: bool = False):
        # `view=True`: the operator returned the caller's `out`; make a second handle on the
        # same buffer so that both can be freed (python/dsc/tensor.py:160-161).
        c_ptr = c_ptr if not view else B.dsc_view(_get_ctx(), c_ptr)
        c = c_ptr.contents
        self._dtype = Dtype(c.dtype)
        self._n_dim = c.n_dim
        self._shape = tuple(c.shape[_DSC_MAX_DIMS - c.n_dim:])
        self._ne = c.ne
        self._c_ptr = c_ptr
        # the context (and its clear() epoch) this handle belongs to: it is freed against THAT context only
        self._owner = _current_ctx()
        self._epoch = self._owner.epoch if self._owner is not None else -1

    def __del__(self):
        # Never create a context from a destructor, never free into a context that has been cleared or replaced since
        # this handle was made (dsc_ctx_clear already released it; its header address may belong to a new tensor).
        try:
            owner = self._owner
            if owner is not None and owner is _current_ctx() and owner._ctx and owner.epoch == self._epoch:
                B.dsc_tensor_free(owner._ctx, self._c_ptr)
        except Exception:      # interpreter teardown
            pass

    @property
    def dtype(self) -> Dtype:
        return self._dtype

    @property
    def shape(self) -> Tuple[int, ...]:
        return self._shape

    @property
    def n_dim(self) -> int:
        return self._n_dim

    @property
    def ne(self) -> int:
        return self._ne

    def __len__(self) -> int:
        return self._shape[0]

    def __str__(self) -> str:
        return str(self.numpy())

    def __mul__(self, other):
        return mul(self, other)

    def __rmul__(self, other):
        return mul(other, self)

    def __add__(self, other):
        return add(self, other)

    def __radd__(self, other):
        return add(other, self)

    def __sub__(self, other):
        return sub(self, other)

    def __rsub__(self, other):
        return sub(other, self)

    def __truediv__(self, other):
        return true_div(self, other)

    def __rtruediv__(self, other):
        return true_div(other, self)

    def __getitem__(self, item):
        """python/dsc/tensor.py:193-229: ints -> dsc_tensor_get_idx (a fully indexed element is unwrapped to a
        Python scalar), slices or mixed -> dsc_tensor_get_slice.  The copy happens on the device."""
        ctx = _get_ctx()
        if isinstance(item, int):
            return _unwrap(Tensor(B.dsc_tensor_get_idx(ctx, self._c_ptr, item)))
        if isinstance(item, tuple) and all(isinstance(i, int) for i in item):
            return _unwrap(Tensor(B.dsc_tensor_get_idx(ctx, self._c_ptr, *item)))
        if isinstance(item, slice):
            return Tensor(B.dsc_tensor_get_slice(ctx, self._c_ptr, _c_slice(item)))
        if isinstance(item, tuple) and all(isinstance(i, (int, slice)) for i in item):
            return Tensor(B.dsc_tensor_get_slice(ctx, self._c_ptr, *[_c_slice(i) for i in item]))
        raise RuntimeError(f'cannot index Tensor with object {item}')

    def __setitem__(self, key, value):
        """python/dsc/tensor.py:231-270"""
        ctx = _get_ctx()
        val = _wrap(value, self._dtype)
        if val.dtype != self._dtype:
            val = val.cast(self._dtype)
        if isinstance(key, int):
            B.dsc_tensor_set_idx(ctx, self._c_ptr, val._c_ptr, key)
        elif isinstance(key, tuple) and all(isinstance(i, int) for i in key):
            B.dsc_tensor_set_idx(ctx, self._c_ptr, val._c_ptr, *key)
        elif isinstance(key, slice):
            B.dsc_tensor_set_slice(ctx, self._c_ptr, val._c_ptr, _c_slice(key))
        elif isinstance(key, tuple) and all(isinstance(i, (int, slice)) for i in key):
            B.dsc_tensor_set_slice(ctx, self._c_ptr, val._c_ptr, *[_c_slice(i) for i in key])
        else:
            raise RuntimeError(f'cannot index Tensor with object {key}')

    def numpy(self) -> np.ndarray:
        """Device -> host copy (the reference returns a zero-copy view, tensor.py:305-323)."""
        out = np.empty(self._shape if self._n_dim > 0 else (1,), dtype=DTYPE_TO_NP[self._dtype])
        B.dsc_copy_to_host(_get_ctx(), self._c_ptr, out.ctypes.data, out.nbytes)
        return out

    def tobytes(self) -> bytes:
        return self.numpy().tobytes()

    def __bytes__(self) -> bytes:
        return self.tobytes()

    def cast(self, dtype: Dtype) -> 'Tensor':
        out_ptr = B.dsc_cast(_get_ctx(), self._c_ptr, dtype.value)
        same = B.ctypes.cast(out_ptr, B.c_void_p).value == B.ctypes.cast(self._c_ptr, B.c_void_p).value
        return Tensor(out_ptr, view=same)


def _unwrap(x: Tensor):
    """python/dsc/tensor.py:91-103: a 1-element 1-D tensor becomes a Python scalar (one device -> host copy)."""
    if x.n_dim != 1 or len(x) != 1:
        return x
    v = x.numpy()[0]
    return complex(v) if np.iscomplexobj(v) else float(v)


def _c_slice(x) -> 'B._DscSlice':
    """python/dsc/tensor.py:106-118: None -> DSC_VALUE_NONE; an int inside a mixed key -> (i, i, i)."""
    if isinstance(x, slice):
        f = lambda i: B.DSC_VALUE_NONE if i is None else int(i)      # noqa: E731
        return B._DscSlice(f(x.start), f(x.stop), f(x.step))
    return B._DscSlice(int(x), int(x), int(x))


def _create_tensor(dtype: Dtype, *dims: int) -> Tensor:
    if not 1 <= len(dims) <= _DSC_MAX_DIMS:
        raise RuntimeError(f'cannot create a Tensor with {len(dims)} dimensions, max is {_DSC_MAX_DIMS}')
    f = (B.dsc_tensor_1d, B.dsc_tensor_2d, B.dsc_tensor_3d, B.dsc_tensor_4d)[len(dims) - 1]
    return Tensor(f(_get_ctx(), dtype.value, *dims))


def empty(shape, dtype: Dtype = Dtype.F32) -> Tensor:
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    return _create_tensor(dtype, *shape)


def from_numpy(x: np.ndarray) -> Tensor:
    if x.dtype not in NP_TO_DTYPE:
        raise RuntimeError(f'NumPy dtype {x.dtype} is not supported')
    x = np.ascontiguousarray(x)
    out = _create_tensor(NP_TO_DTYPE[x.dtype], *(x.shape if x.ndim > 0 else (1,)))
    B.dsc_copy_from_host(_get_ctx(), out._c_ptr, x.ctypes.data, x.nbytes)
    return out


def _wrap(x, dtype: Union[Dtype, None] = None) -> Tensor:
    if isinstance(x, np.ndarray):
        return from_numpy(x)
    if isinstance(x, Tensor):
        return x
    ctx = _get_ctx()
    if isinstance(x, complex):
        if dtype == Dtype.C64:
            return Tensor(B.dsc_wrap_c64(ctx, B._C64(x.real, x.imag)))
        return Tensor(B.dsc_wrap_c32(ctx, B._C32(x.real, x.imag)))
    if dtype == Dtype.F64:
        return Tensor(B.dsc_wrap_f64(ctx, float(x)))
    if dtype == Dtype.C32:
        return Tensor(B.dsc_wrap_c32(ctx, B._C32(float(x), 0.0)))
    if dtype == Dtype.C64:
        return Tensor(B.dsc_wrap_c64(ctx, B._C64(float(x), 0.0)))
    return Tensor(B.dsc_wrap_f32(ctx, float(x)))


def _wrap_operands(xa, xb) -> Tuple[Tensor, Tensor]:
    # python/dsc/tensor.py:435-458
    def _dtype(x) -> Dtype:
        if isinstance(x, Tensor):
            return x.dtype
        if isinstance(x, np.ndarray):
            return NP_TO_DTYPE[x.dtype]
        if isinstance(x, (int, float)):
            return Dtype.F32
        return Dtype.C32

    if (isinstance(xa, Tensor) and isinstance(xb, Tensor)) or (isinstance(xa, np.ndarray) and isinstance(xb, np.ndarray)):
        return _wrap(xa), _wrap(xb)
    wrap_dtype = DTYPE_CONVERSION_TABLES[_dtype(xa).value][_dtype(xb).value]
    return _wrap(xa, wrap_dtype), _wrap(xb, wrap_dtype)


def _binary(f, xa, xb, out) -> Tensor:
    xa, xb = _wrap_operands(xa, xb)
    return Tensor(f(_get_ctx(), xa._c_ptr, xb._c_ptr, _c_ptr_or_none(out)), out is not None)


def mul(xa, xb, out: Union[Tensor, None] = None) -> Tensor:
    return _binary(B.dsc_mul, xa, xb, out)


def add(xa, xb, out: Union[Tensor, None] = None) -> Tensor:         # python/dsc/tensor.py:461-467
    return _binary(B.dsc_add, xa, xb, out)


def sub(xa, xb, out: Union[Tensor, None] = None) -> Tensor:         # :469-475
    return _binary(B.dsc_sub, xa, xb, out)


def true_div(xa, xb, out: Union[Tensor, None] = None) -> Tensor:    # :485-491
    return _binary(B.dsc_div, xa, xb, out)


def _same_ptr(a, b) -> bool:
    return B.ctypes.cast(a, B.c_void_p).value == B.ctypes.cast(b, B.c_void_p).value


def absolute(x: Tensor, out: Union[Tensor, None] = None) -> Tensor:       # python/dsc/tensor.py:530-534
    return Tensor(B.dsc_abs(_get_ctx(), x._c_ptr, _c_ptr_or_none(out)), out is not None)


def angle(x: Tensor) -> Tensor:                                           # :537-538
    return Tensor(B.dsc_angle(_get_ctx(), x._c_ptr))


def conj(x: Tensor) -> Tensor:                                            # :541-544 (a real tensor comes back as a view of itself)
    p = B.dsc_conj(_get_ctx(), x._c_ptr)
    return Tensor(p, view=_same_ptr(p, x._c_ptr))


def real(x: Tensor) -> Tensor:                                            # :547-550
    p = B.dsc_real(_get_ctx(), x._c_ptr)
    return Tensor(p, view=_same_ptr(p, x._c_ptr))


def imag(x: Tensor) -> Tensor:                                            # :553-554
    return Tensor(B.dsc_imag(_get_ctx(), x._c_ptr))


def _reduce(f, x: Tensor, out, axis: int, keepdims: bool) -> Tensor:
    return Tensor(f(_get_ctx(), x._c_ptr, _c_ptr_or_none(out), axis, keepdims), out is not None)


def sum(x: Tensor, out=None, axis: int = -1, keepdims: bool = True) -> Tensor:
    return _reduce(B.dsc_sum, x, out, axis, keepdims)


def mean(x: Tensor, out=None, axis: int = -1, keepdims: bool = True) -> Tensor:
    return _reduce(B.dsc_mean, x, out, axis, keepdims)


def max(x: Tensor, out=None, axis: int = -1, keepdims: bool = True) -> Tensor:
    return _reduce(B.dsc_max, x, out, axis, keepdims)


def min(x: Tensor, out=None, axis: int = -1, keepdims: bool = True) -> Tensor:
    return _reduce(B.dsc_min, x, out, axis, keepdims)


def plan_fft(n: int, fft_type: int = 1, dtype: Dtype = Dtype.F64):
    """Build (or touch) the plan for an n-point transform.  fft_type: 0 REAL, 1 COMPLEX.
    (The reference's Python plan_fft passes dtype in the fft_type slot and aborts,
    _bindings.py:88-93 vs dsc.h:139-141; this one follows the C signature.)"""
    return B.dsc_plan_fft(_get_ctx(), n, fft_type, dtype.value)


def _fft_like(f, x: Tensor, out, n: int, axis: int) -> Tensor:
    return Tensor(f(_get_ctx(), x._c_ptr, _c_ptr_or_none(out), n, axis), out is not None)


def fft(x: Tensor, out=None, n: int = -1, axis: int = -1) -> Tensor:
    return _fft_like(B.dsc_fft, x, out, n, axis)


def ifft(x: Tensor, out=None, n: int = -1, axis: int = -1) -> Tensor:
    return _fft_like(B.dsc_ifft, x, out, n, axis)


def rfft(x: Tensor, out=None, n: int = -1, axis: int = -1) -> Tensor:
    return _fft_like(B.dsc_rfft, x, out, n, axis)


def irfft(x: Tensor, out=None, n: int = -1, axis: int = -1) -> Tensor:
    return _fft_like(B.dsc_irfft, x, out, n, axis)


def filter_fft(s: Tensor, H: Tensor, out=None) -> Tensor:
    """irfft(rfft(s, n) * H) with n = 2 * (len(H) - 1), fused where a kernel exists."""
    return Tensor(B.dsc_filter_fft(_get_ctx(), s._c_ptr, H._c_ptr, _c_ptr_or_none(out)), out is not None)


def transpose(x: Tensor, axes=None) -> Tensor:                            # python/dsc/tensor.py:407-416
    if axes is None or (isinstance(axes, (tuple, list)) and all(isinstance(a, int) for a in axes)):
        return Tensor(B.dsc_transpose(_get_ctx(), x._c_ptr, *(tuple(axes) if axes is not None else ())))
    raise RuntimeError(f'cannot transpose axes {axes}')


def fftfreq(n: int, d: float = 1.0, dtype: Dtype = Dtype.F32) -> Tensor:   # python/dsc/tensor.py:729-730
    return Tensor(B.dsc_fftfreq(_get_ctx(), n, d, dtype.value))


def rfftfreq(n: int, d: float = 1.0, dtype: Dtype = Dtype.F32) -> Tensor:  # python/dsc/tensor.py:733-734
    return Tensor(B.dsc_rfftfreq(_get_ctx(), n, d, dtype.value))
