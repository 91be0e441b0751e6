"""Seeded differential fuzzing of the HIP path (through the C ABI) against the CPU oracle: random ranks, shapes, axes, lengths
(`n=` pad / crop, non powers of two), dtypes and broadcast patterns — the combinations the fixed parity tests do not spell out.
Every case prints enough to be replayed (`seed`, `case index`).  Tolerances as tests/helpers.py (1e-5 rel f32, 1e-12 f64)."""
import numpy as np
import pytest

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu

REAL = (np.float32, np.float64)
CPLX = (np.complex64, np.complex128)


@pytest.fixture(scope='module')
def dsc():
    import dsc_amd
    try:
        dsc_amd.init(12 << 30, 4 << 30)
    except RuntimeWarning:
        pass
    yield dsc_amd


@pytest.fixture(autouse=True)
def _sync(dsc):
    yield
    dsc.synchronize()


def _rand(rng, shape, dt):
    x = rng.standard_normal(shape)
    if np.dtype(dt).kind == 'c':
        x = x + 1j * rng.standard_normal(shape)
    return x.astype(dt)


def _pow2ceil(n):
    p = 1
    while p < n:
        p *= 2
    return p


def _axis_length(rng):
    kind = rng.integers(0, 5)
    if kind == 0:
        return int(rng.integers(1, 70))
    if kind == 1:
        return int(rng.integers(70, 5000))
    if kind == 2:
        return 1 << int(rng.integers(0, 18))
    if kind == 3:
        return (1 << int(rng.integers(3, 17))) + int(rng.integers(-2, 3))        # around a power of two
    return int(rng.integers(5000, 70000))


def _shape_with_axis(rng, budget=1 << 19):
    nd = int(rng.integers(1, 5))
    axis = int(rng.integers(0, nd))
    la = _axis_length(rng)
    rest = max(1, budget // _pow2ceil(2 * la))
    shape = [1] * nd
    shape[axis] = la
    for d in rng.permutation(nd):
        if d == axis:
            continue
        shape[d] = int(rng.integers(1, min(rest, 40) + 1))
        rest = max(1, rest // shape[d])
    if rng.integers(0, 2):
        axis -= nd                                                                  # negative spelling of the same axis
    return shape, axis, la


@pytest.mark.parametrize('seed', [101, 202, 303])
def test_fuzz_transforms(dsc, seed):
    """dsc_fft / dsc_ifft / dsc_rfft / dsc_irfft (dsc.cpp:1958-2260) on random ranks, axes, lengths and `n=`."""
    from oracle import port
    rng = np.random.default_rng(seed)
    paths = set()
    done = 0
    for case in range(90):
        op = ('fft', 'ifft', 'rfft', 'irfft')[int(rng.integers(0, 4))]
        shape, axis, la = _shape_with_axis(rng)
        if op == 'rfft':
            dt = REAL[int(rng.integers(0, 2))]
        elif op == 'irfft':
            dt = CPLX[int(rng.integers(0, 2))]
        else:
            dt = (REAL + CPLX)[int(rng.integers(0, 4))]
        n = -1 if rng.random() < 0.55 else int(rng.integers(1, 2 * la + 2))
        x = _rand(rng, shape, dt)
        what = f'seed={seed} case={case} {op} {np.dtype(dt).name} shape={shape} axis={axis} n={n}'
        try:
            want = getattr(port, op)(x, n, axis)
        except ValueError:
            continue                                                                # the reference aborts here (e.g. a zero-length plan)
        got = getattr(dsc, op)(dsc.from_numpy(x), n=n, axis=axis)
        paths.add(dsc.last_fft_path())
        assert_close(got.numpy(), want, what=f'{what} path={dsc.last_fft_path()}')
        done += 1
    assert done >= 60, done
    assert len(paths) >= 4, paths                                                   # the draw reaches several kernel families


def _broadcast_pair(rng):
    nd = int(rng.integers(1, 5))
    full = [int(rng.integers(1, 9)) for _ in range(nd)]
    if rng.integers(0, 3) == 0:
        full[-1] = int(rng.integers(1, 3000))
    sa, sb = list(full), list(full)
    for d in range(nd):
        r = rng.integers(0, 6)
        if r == 0:
            sa[d] = 1
        elif r == 1:
            sb[d] = 1
    cut = int(rng.integers(0, nd))                                                  # one operand may have fewer dimensions
    if rng.integers(0, 2):
        sa = sa[cut:]
    else:
        sb = sb[cut:]
    return sa, sb


@pytest.mark.parametrize('seed', [11, 12])
def test_fuzz_binary_ops(dsc, seed):
    """dsc_add / sub / mul / div with NumPy-style broadcasting over right-aligned dims and the promotion table
    (dsc.cpp:44-69, 1186-1310; dsc_dtype.h:52-78), tensor-tensor and tensor-scalar."""
    from oracle import port
    rng = np.random.default_rng(seed)
    fns = ((dsc.add, port.ADD), (dsc.sub, port.SUB), (dsc.mul, port.MUL), (dsc.true_div, port.DIV))
    for case in range(120):
        f, op = fns[int(rng.integers(0, 4))]
        sa, sb = _broadcast_pair(rng)
        da, db = (REAL + CPLX)[int(rng.integers(0, 4))], (REAL + CPLX)[int(rng.integers(0, 4))]
        a, b = _rand(rng, sa, da), _rand(rng, sb, db)
        if op == port.DIV:
            b = (b + np.sign(b.real) * 0.5 + (b.real == 0)).astype(db)              # keep the divisor away from zero
        what = f'seed={seed} case={case} op={op} {np.dtype(da).name}{sa} x {np.dtype(db).name}{sb}'
        want = port.binary(a, b, op)
        got = f(dsc.from_numpy(a), dsc.from_numpy(b))
        assert_close(got.numpy(), want, what=what)


@pytest.mark.parametrize('seed', [21, 22])
def test_fuzz_reductions_and_unary(dsc, seed):
    """dsc_sum / mean / max / min on every axis and keep_dims (dsc.cpp:1774-1953) and abs / angle / conj / real / imag
    (dsc.cpp:1480-1622) on random ranks."""
    from oracle import port
    rng = np.random.default_rng(seed)
    for case in range(80):
        nd = int(rng.integers(1, 5))
        shape = [int(rng.integers(1, 12)) for _ in range(nd)]
        if rng.integers(0, 3) == 0:
            shape[int(rng.integers(0, nd))] = int(rng.integers(12, 4000))
        dt = (REAL + CPLX)[int(rng.integers(0, 4))]
        x = _rand(rng, shape, dt)
        axis = int(rng.integers(-nd, nd))
        keep = bool(rng.integers(0, 2))
        what = f'seed={seed} case={case} {np.dtype(dt).name}{shape} axis={axis} keep={keep}'
        n_axis = shape[axis]
        tol = (2e-5 if np.dtype(dt).itemsize in (4, 8) and np.dtype(dt).name in ('float32', 'complex64') else 1e-12)
        for name, op in (('sum', port.SUM), ('mean', port.MEAN)):
            got = getattr(dsc, name)(dsc.from_numpy(x), axis=axis, keepdims=keep).numpy()
            want = port.reduce(x, op, axis, keep)
            assert got.shape == want.shape and got.dtype == want.dtype, what
            assert np.max(np.abs(got - want)) <= tol * max(1.0, float(np.max(np.abs(x)))) * np.sqrt(n_axis), (name, what)
        xq = (np.round(x.real * 2) / 2 + (1j * x.imag if np.dtype(dt).kind == 'c' else 0)).astype(dt)    # ties on the real part
        for name, op in (('max', port.MAX), ('min', port.MIN)):
            got = getattr(dsc, name)(dsc.from_numpy(xq), axis=axis, keepdims=keep).numpy()
            assert np.array_equal(got, port.reduce(xq, op, axis, keep)), (name, what)
        for f, op in ((dsc.absolute, port.ABS), (dsc.angle, port.ANGLE), (dsc.conj, port.CONJ), (dsc.real, port.REALPART), (dsc.imag, port.IMAGPART)):
            assert_close(f(dsc.from_numpy(x)).numpy(), port.unary(x, op), what=f'{f.__name__} {what}')
