"""CPU restatement of the reference's indexing / slicing (dsc/src/dsc.cpp:829-1169 with the
iterator of dsc/include/dsc_iter.h:125-190), on numpy arrays.  Index arithmetic only, written
as the reference writes it (parse, then walk the region in row-major order), not with numpy's
own slicing.

TEST INFRASTRUCTURE ONLY — pinned bit-exact against oracle/_ref in tests/test_oracle_vs_ref.py
and against tests/golden/slice.npz; never imported by dsc_amd/.

A violated reference assertion raises `Abort` (the reference prints and exits)."""
import numpy as np

NONE = 2 ** 31 - 1                      # DSC_VALUE_NONE, dsc.h:78


class Abort(Exception):
    pass


def _check(cond, what):
    if not cond:
        raise Abort(what)


def raw_slice(k):
    """python/dsc/tensor.py:106-118"""
    if isinstance(k, slice):
        f = lambda i: NONE if i is None else int(i)      # noqa: E731
        return [f(k.start), f(k.stop), f(k.step)]
    return [int(k)] * 3


def parse_slices(shape, raw):
    """dsc.cpp:883-933.  shape: the tensor's logical shape; raw: [start, stop, step] per leading dim.
    Returns (parsed, collapse)."""
    parsed, collapse = [], []
    for i, (start, stop, step) in enumerate(raw):
        dim = shape[i]
        col = False
        if start == stop and start == step and start != NONE:
            col = True
            step = 1
            if start < 0:
                start += dim
                stop += dim + 1
            else:
                stop += 1
        _check(step != 0, 'step != 0')
        if step == NONE:
            step = 1
        if start == NONE:
            start = 0 if step > 0 else dim - 1
        if stop == NONE:
            stop = dim if step > 0 else -dim - 1
        if start < 0:
            start += dim
        if stop < 0:
            stop += dim
        _check(abs(stop - start) <= dim, 'abs(stop - start) <= dim')
        _check((step > 0 and start < stop) or (step < 0 and start > stop), 'empty slice')
        _check(abs(step) <= dim, 'abs(step) <= dim')
        parsed.append((start, stop, step))
        collapse.append(col)
    return parsed, collapse


def _walk(shape, parsed):
    """dsc_iter.h:125-190: per-dimension index lists of the region, in iteration order."""
    lists = []
    for i, dim in enumerate(shape):
        if i < len(parsed):
            start, stop, step = parsed[i]
            idx, out = start, []
            while (step > 0 and idx < stop) or (step < 0 and idx > stop):
                _check(0 <= idx < dim, 'index inside the dimension')      # the reference reads out of bounds here
                out.append(idx)
                idx += step
            lists.append(out)
        else:
            lists.append(list(range(dim)))
    return lists


def _flat_offsets(shape, lists):
    strides = [int(np.prod(shape[i + 1:])) for i in range(len(shape))]
    offs = np.zeros(1, dtype=np.int64)
    for lst, st in zip(lists, strides):
        offs = (offs[:, None] + np.asarray(lst, dtype=np.int64)[None, :] * st).reshape(-1)
    return offs


def get_slice(x, *key):
    """dsc.cpp:935-992"""
    x = np.ascontiguousarray(x)
    _check(len(key) <= 4, 'slices <= DSC_MAX_DIMS')
    _check(len(key) <= x.ndim, 'too many slices')
    parsed, collapse = parse_slices(x.shape, [raw_slice(k) for k in key])
    lists = _walk(x.shape, parsed)
    out_shape = [len(lst) for i, lst in enumerate(lists) if not (i < len(key) and collapse[i])]
    _check(len(out_shape) >= 1, 'at least one dimension left')
    return x.reshape(-1)[_flat_offsets(x.shape, lists)].reshape(out_shape)


def get_idx(x, *idx):
    """dsc.cpp:832-866"""
    x = np.ascontiguousarray(x)
    _check(1 <= len(idx) <= 4, 'indexes <= DSC_MAX_DIMS')
    _check(len(idx) <= x.ndim, 'too many indexes')
    el = []
    for i, v in enumerate(idx):
        if v < 0:
            v += x.shape[i]
        _check(0 <= v < x.shape[i], 'index in range')
        el.append(v)
    strides = [int(np.prod(x.shape[i + 1:])) for i in range(x.ndim)]
    offset = sum(s * v for s, v in zip(strides, el))
    count = strides[len(idx) - 1]
    out_shape = x.shape[len(idx):] if x.ndim > len(idx) else (1,)
    return x.reshape(-1)[offset:offset + count].reshape(out_shape).copy()


def _tensor_set(xa, xb, parsed):
    """dsc.cpp:1010-1041: the region in iteration order takes xb cyclically"""
    lists = _walk(xa.shape, parsed)
    offs = _flat_offsets(xa.shape, lists)
    flat = xa.reshape(-1)
    src = np.ascontiguousarray(xb).reshape(-1)
    flat[offs] = src[np.arange(len(offs)) % src.size]
    return flat.reshape(xa.shape)


def set_slice(xa, xb, *key):
    """dsc.cpp:1108-1169; returns the modified copy of xa"""
    xa = np.ascontiguousarray(xa).copy()
    xb = np.ascontiguousarray(xb)
    _check(len(key) <= xa.ndim, 'slices <= n_dim')
    _check(xa.dtype == xb.dtype, 'same dtype')
    parsed, _ = parse_slices(xa.shape, [raw_slice(k) for k in key])
    slice_shape = [len(range(*p)) if i < len(parsed) else xa.shape[i] for i, p in enumerate(parsed + [None] * (xa.ndim - len(parsed)))]
    xb_scalar = xb.ndim == 1 and xb.shape[-1] == 1
    if not xb_scalar:
        for i in range(min(xa.ndim, xb.ndim)):
            _check(slice_shape[i] == 1 or xb.shape[i] == 1 or slice_shape[i] == xb.shape[i], 'broadcastable')
    return _tensor_set(xa, xb, parsed)


def set_idx(xa, xb, *idx):
    """dsc.cpp:1043-1106; returns the modified copy of xa"""
    xa = np.ascontiguousarray(xa).copy()
    xb = np.ascontiguousarray(xb)
    _check(len(idx) <= xa.ndim, 'indexes <= n_dim')
    _check(xa.dtype == xb.dtype, 'same dtype')
    parsed = []
    for i, v in enumerate(idx):
        start, stop = v, v + 1
        if v < 0:
            start += xa.shape[i]
            stop += xa.shape[i]
        parsed.append((start, stop, 1))
    sub_ndim = xa.ndim - len(idx)
    sub_shape = [xa.shape[i - len(idx)] for i in range(len(idx), xa.ndim)]      # dsc.cpp:1071-1073, as written
    xb_scalar = xb.ndim == 1 and xb.shape[-1] == 1
    if sub_ndim == 0:
        _check(xb_scalar, 'scalar value for a single element')
    if not xb_scalar:
        _check(xb.ndim == sub_ndim, 'xb.n_dim == remaining dims')
        for i in range(sub_ndim):
            _check(sub_shape[i] == xb.shape[i], 'xb shape')
    return _tensor_set(xa, xb, parsed)


def transpose(x, axes=None):
    """dsc.cpp:764-827 with copy_with_stride (:748-762): out is dense, x is read through the permuted strides."""
    x = np.ascontiguousarray(x)
    if x.ndim == 1:
        return x.copy()
    if axes is None or len(axes) == 0:
        axes = [x.ndim - (i + 1) for i in range(x.ndim)]
    _check(len(axes) == x.ndim, 'axes == n_dim')
    for a in axes:
        _check(0 <= a < x.ndim, 'axis in range')
    strides = [int(np.prod(x.shape[i + 1:])) for i in range(x.ndim)]
    sw_shape = [x.shape[a] for a in axes]
    sw_stride = [strides[a] for a in axes]
    offs = np.zeros(1, dtype=np.int64)
    for n, st in zip(sw_shape, sw_stride):
        offs = (offs[:, None] + np.arange(n, dtype=np.int64)[None, :] * st).reshape(-1)
    return x.reshape(-1)[offs].reshape(sw_shape)


def fftfreq(n, d=1.0, dtype=np.float32):
    """dsc.cpp:2262-2302, evaluated in the output precision T: factor = 1 / (n * d); i * factor"""
    _check(n > 0, 'n > 0')
    T = np.dtype(dtype).type
    factor = T(1) / (T(n) * T(d))
    odd = n & 1
    n2 = (n - 1) >> 1 if odd else n >> 1
    out = np.empty(n, dtype=dtype)
    for i in range(n2 + odd):
        out[i] = T(i) * factor
    for i in range(n2):
        out[n2 + odd + i] = T(-n2 + i) * factor
    return out


def rfftfreq(n, d=1.0, dtype=np.float32):
    """dsc.cpp:2304-2340"""
    _check(n > 0, 'n > 0')
    T = np.dtype(dtype).type
    factor = T(1) / (T(n) * T(d))
    count = ((n - 1) >> 1) + 1 if n & 1 else (n >> 1) + 1
    return np.array([T(i) * factor for i in range(count)], dtype=dtype)
