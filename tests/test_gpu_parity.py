"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes, dsc_amd), against
(1) the committed golden fixtures = outputs of the reference itself, and (2) the CPU oracle on
seeded inputs.  Tolerance is BASELINE.json's: relative L2 <= 1e-5 for f32, 1e-12 for f64
(tests/helpers.py:TOL), max/min exact."""
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import assert_close, rel_l2

pytestmark = pytest.mark.gpu


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


OPS = ('fft', 'ifft', 'rfft', 'irfft')


@pytest.mark.parametrize('group', ['fft_small', 'fft_large'])
def test_golden_fft_family(dsc, golden, group):
    n = 0
    for rec, xs, y in golden.cases(group):
        got = getattr(dsc, rec['op'])(dsc.from_numpy(xs[0]), n=rec['n'], axis=rec['axis']).numpy()
        assert_close(got, y, what=f"{rec['key']} n={rec['n']} axis={rec['axis']} in={xs[0].shape} path={dsc.last_fft_path()}")
        n += 1
    assert n > 0


def test_golden_mul(dsc, golden):
    for rec, xs, y in golden.cases('mul'):
        got = dsc.mul(dsc.from_numpy(xs[0]), dsc.from_numpy(xs[1])).numpy()
        assert_close(got, y, what=rec['key'])


def test_golden_add_sub_div(dsc, golden):
    f = {'add': dsc.add, 'sub': dsc.sub, 'div': dsc.true_div}
    for rec, xs, y in golden.cases('binary'):
        got = f[rec['op']](dsc.from_numpy(xs[0]), dsc.from_numpy(xs[1])).numpy()
        assert_close(got, y, what=rec['key'])
    a = dsc.from_numpy(np.arange(6, dtype=np.float32).reshape(2, 3))
    assert np.array_equal((a + 1).numpy(), np.arange(6, dtype=np.float32).reshape(2, 3) + 1)
    assert np.array_equal((1 - a).numpy(), 1 - np.arange(6, dtype=np.float32).reshape(2, 3))
    assert np.allclose((a / 2).numpy(), np.arange(6, dtype=np.float32).reshape(2, 3) / 2)


def test_golden_unary(dsc, golden):
    """abs / angle / conj / real / imag (dsc.cpp:1480-1622) incl. the dtype rules and the
    'real input comes back as itself' rule of conj / real."""
    f = {'abs': dsc.absolute, 'angle': dsc.angle, 'conj': dsc.conj, 'real': dsc.real, 'imag': dsc.imag}
    for rec, xs, y in golden.cases('unary'):
        got = f[rec['op']](dsc.from_numpy(xs[0])).numpy()
        assert got.dtype == y.dtype and got.shape == y.shape, rec['key']
        tol = 1e-5 if y.dtype in (np.float32, np.complex64) else 1e-12
        assert np.allclose(got, y, rtol=tol, atol=tol), rec['key']
    x = dsc.from_numpy(np.ones(4, np.float32))
    assert dsc.conj(x)._c_ptr.contents.data == x._c_ptr.contents.data          # same buffer, second handle


def test_golden_reductions(dsc, golden):
    for rec, xs, y in golden.cases('reduce'):
        got = getattr(dsc, rec['op'])(dsc.from_numpy(xs[0]), axis=rec['axis'], keepdims=rec['keepdims']).numpy()
        if rec['op'] in ('max', 'min'):
            assert got.shape == y.shape and np.array_equal(got, y), rec['key']
        else:
            assert_close(got, y, what=f"{rec['key']} axis={rec['axis']}")


def test_golden_filter_pipeline(dsc, golden):
    """README filterFFT (README.md:113-135) as the four operator calls."""
    for rec, xs, y in golden.cases('filter'):
        s, b = dsc.from_numpy(xs[0]), dsc.from_numpy(xs[1])
        S, B = dsc.rfft(s, n=rec['n']), dsc.rfft(b, n=rec['n'])
        got = dsc.irfft(S * B).numpy()
        assert_close(got, y, what=rec['key'])


def test_all_axes_like_reference_test(dsc):
    """python/tests/test_ops.py:458-489: [8,8,8,8] with the transformed axis = n, every axis,
    crop / copy / pad through n=, forward then inverse — for f64 (as there) and f32."""
    from oracle import port
    rng = np.random.default_rng(42)
    for dt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
        n_ = 6
        for axis in range(4):
            shape = [8] * 4
            shape[axis] = 2 ** n_
            for change in (-1, 0, 1):
                n = 2 ** (n_ + change)
                x = rng.standard_normal(shape).astype(dt)
                X = dsc.rfft(dsc.from_numpy(x), n=n, axis=axis)
                assert_close(X.numpy(), port.rfft(x, n, axis), what=f'rfft {dt.__name__} axis={axis} n={n}')
                assert_close(dsc.irfft(X, axis=axis).numpy(), port.irfft(port.rfft(x, n, axis), -1, axis), what='irfft')
                xc = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(cdt)
                F = dsc.fft(dsc.from_numpy(xc), n=n, axis=axis)
                assert_close(F.numpy(), port.fft(xc, n, axis), what=f'fft axis={axis} n={n}')
                assert_close(dsc.ifft(F, axis=axis).numpy(), port.ifft(port.fft(xc, n, axis), -1, axis), what='ifft')
                # the reference's own oracle
                assert np.allclose(X.numpy(), np.fft.rfft(x, n=n, axis=axis), atol=1e-5 if dt == np.float64 else 2e-3, rtol=1e-5)


@pytest.mark.parametrize('N', [1, 2, 4, 8, 32, 512, 8192, 16384, 32768, 131072])
def test_sizes_f32(dsc, N):
    """Every kernel path: LDS (L <= 8192) and four-step (above), f32."""
    from oracle import port
    rng = np.random.default_rng(N)
    rows = 3 if N <= 32768 else 2
    x = rng.standard_normal((rows, N)).astype(np.float32)
    if N >= 2:
        X = dsc.rfft(dsc.from_numpy(x))
        assert_close(X.numpy(), port.rfft(x), what=f'rfft N={N} {dsc.last_fft_path()}')
        assert_close(dsc.irfft(X).numpy(), port.irfft(port.rfft(x)), what=f'irfft N={N} {dsc.last_fft_path()}')
    xc = (x + 1j * rng.standard_normal(x.shape)).astype(np.complex64)
    F = dsc.fft(dsc.from_numpy(xc))
    assert_close(F.numpy(), port.fft(xc), what=f'fft N={N} {dsc.last_fft_path()}')
    assert_close(dsc.ifft(F).numpy(), port.ifft(port.fft(xc)), what=f'ifft N={N}')


@pytest.mark.parametrize('N', [16, 4096, 8192, 262144])
def test_sizes_f64(dsc, N):
    from oracle import port
    rng = np.random.default_rng(N + 1)
    x = rng.standard_normal((2, N))
    X = dsc.rfft(dsc.from_numpy(x))
    assert_close(X.numpy(), port.rfft(x), what=f'rfft f64 N={N} {dsc.last_fft_path()}')
    assert_close(dsc.irfft(X).numpy(), port.irfft(port.rfft(x)), what=f'irfft f64 N={N}')
    xc = x + 1j * rng.standard_normal(x.shape)
    assert_close(dsc.fft(dsc.from_numpy(xc)).numpy(), port.fft(xc), what=f'fft f64 N={N}')


def test_out_argument_and_views(dsc):
    """`out=` is written in place and the result is a second handle on the same buffer
    (tensor.py:160-161, dsc.cpp:399-401)."""
    from oracle import port
    x = np.random.default_rng(1).standard_normal((4, 256)).astype(np.float32)
    out = dsc.empty((4, 129), dsc.Dtype.C32)
    res = dsc.rfft(dsc.from_numpy(x), out=out)
    assert_close(out.numpy(), port.rfft(x))
    assert_close(res.numpy(), port.rfft(x))
    assert res._c_ptr.contents.data == out._c_ptr.contents.data
    assert out._c_ptr.contents.backend == 1


@pytest.mark.parametrize('dt', [np.complex64, np.complex128])
def test_complex_transforms_in_place(dsc, dt):
    """dsc_fft / dsc_ifft with out == x: the reference gathers a line into scratch before it scatters (dsc.cpp:1990-2040), so in place
    is legal there; every route a complex transform can take here (register rows of several sizes, the team kernel, the two-kernel
    route, the column kernel in one pass and as a four-step) must read a line, a tile or the whole tensor before it overwrites it."""
    rng = np.random.default_rng(12)
    tol = 2e-6 if dt == np.complex64 else 1e-13
    for shape, axis in (((300, 64), -1), ((70, 1024), -1), ((6, 32768), -1), ((5, 65536), -1), ((3, 262144), -1), ((2, 1048576), -1),
                        ((256, 300), 0), ((2048, 40), 0), ((4096, 72), 0), ((65536, 24), 0), ((3, 8192, 20), 1), ((16, 40), 0)):
        z = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
        for name in ('fft', 'ifft'):
            t = dsc.from_numpy(z)
            r = getattr(dsc, name)(t, out=t, axis=axis)
            want = getattr(np.fft, name)(z.astype(np.complex128), axis=axis)
            assert rel_l2(t.numpy(), want) <= tol, (name, shape, axis, dsc.last_fft_path())
            assert r._c_ptr.contents.data == t._c_ptr.contents.data


def test_arena_reuse_and_double_free(dsc):
    import dsc_amd._bindings as B
    from dsc_amd.context import _get_ctx
    before = dsc.used_mem()
    t = dsc.empty((1024, 1024), dsc.Dtype.F32)
    assert dsc.used_mem() >= before + 4 * 1024 * 1024
    p = t._c_ptr
    B.dsc_tensor_free(_get_ctx(), p)
    B.dsc_tensor_free(_get_ctx(), p)          # tolerated, as the reference's allocator does
    del t
    assert dsc.used_mem() == before


def test_mul_broadcast_and_scalars(dsc):
    from oracle import port
    rng = np.random.default_rng(9)
    a = (rng.standard_normal((64, 513)) + 1j * rng.standard_normal((64, 513))).astype(np.complex64)
    h = (rng.standard_normal(513) + 1j * rng.standard_normal(513)).astype(np.complex64)
    assert_close((dsc.from_numpy(a) * dsc.from_numpy(h)).numpy(), port.mul(a, h))
    assert_close((dsc.from_numpy(a) * 2.5).numpy(), port.mul(a, np.array([2.5], np.float32).astype(np.complex64)))
    assert_close((3 * dsc.from_numpy(a)).numpy(), port.mul(np.array([3], np.complex64), a))
    assert_close((dsc.from_numpy(a) * (1 + 2j)).numpy(), port.mul(a, np.array([1 + 2j], np.complex64)))
    d = rng.standard_normal((64, 513))
    assert (dsc.from_numpy(d) * dsc.from_numpy(a)).dtype == dsc.Dtype.C32       # F64 x C32 -> C32


def _rand(rng, shape, dt):
    x = rng.standard_normal(shape)
    if np.dtype(dt).kind == 'c':
        x = x + 1j * rng.standard_normal(shape)
    return x.astype(dt)


@pytest.mark.parametrize('shape', [(6, 1000), (5, 1001)])       # even element count: packed kernels; odd: the scalar tail / fallback
def test_binary_ops_every_dtype_pair_in_registers(dsc, shape):
    """Equal shapes, every (dtype, dtype) pair and operator: operands of different dtypes are promoted in registers
    (binary_mixed_pack_kernel) — must give exactly what the reference's cast-both-then-operate gives (dsc.cpp:1186-1223)."""
    from oracle import port
    rng = np.random.default_rng(21)
    dts = (np.float32, np.float64, np.complex64, np.complex128)
    for da in dts:
        for db in dts:
            a, b = _rand(rng, shape, da), _rand(rng, shape, db)
            for op, f in ((0, dsc.add), (1, dsc.sub), (2, dsc.mul), (3, dsc.true_div)):
                got = f(dsc.from_numpy(a), dsc.from_numpy(b)).numpy()
                want = port.binary(a, b, op)
                assert got.dtype == want.dtype, (da, db, op)
                if op >= 2:
                    assert_close(got, want, what=f'op {op} {da} {db}')          # complex products contract into FMAs, divisions differ
                else:
                    assert np.array_equal(got, want), (da, db, op, shape)


@pytest.mark.parametrize('dt', [np.float32, np.float64, np.complex64, np.complex128])
def test_binary_ops_small_operand_packed(dsc, dt):
    """A full operand against a scalar or a trailing-dims operand (either side), row lengths odd and even, plus views that are
    not 16-byte aligned (the packed kernel must not be taken): exact against the oracle."""
    from oracle import port
    rng = np.random.default_rng(22)
    for shape, small in (((8, 513), (513,)), ((8, 512), (512,)), ((3, 4, 130), (4, 130)), ((16, 6), (6,)), ((4, 1024), (1,))):
        a, h = _rand(rng, shape, dt), _rand(rng, small, dt)
        for op, f in ((0, dsc.add), (1, dsc.sub), (2, dsc.mul), (3, dsc.true_div)):
            for got, want in ((f(dsc.from_numpy(a), dsc.from_numpy(h)).numpy(), port.binary(a, h, op)),
                              (f(dsc.from_numpy(h), dsc.from_numpy(a)).numpy(), port.binary(h, a, op))):
                if op >= 2:
                    assert_close(got, want, what=f'op {op} {shape} {small}')
                else:
                    assert np.array_equal(got, want), (shape, small, op)
    a = _rand(rng, (9, 64), dt)
    t = dsc.from_numpy(a)
    odd = t[1:]                                                 # a copy at a fresh address: still exercised through the fast path
    assert_close((odd * 3).numpy(), port.binary(a[1:], np.array([3]).astype(dt), 2))


@pytest.mark.parametrize('dt', [np.float32, np.float64, np.complex64, np.complex128])
def test_binary_ops_general_broadcast_packed(dsc, dt):
    """General broadcasts whose innermost rows are whole 16-byte packs (binary_bcast_pack_kernel) and ones that are not."""
    from oracle import port
    rng = np.random.default_rng(24)
    for sa, sb in (((4, 3, 8, 64), (4, 1, 8, 1)), ((3, 1, 64), (4, 1, 8, 1)), ((16, 1), (16, 72)), ((5, 1, 24), (1, 7, 24)),
                   ((12, 513), (12, 1)), ((6, 1, 1), (6, 5, 22)), ((2, 3, 1, 1), (2, 3, 7, 10)), ((9, 2), (9, 1)),      # column broadcasts

                   ((2, 3, 4, 6), (3, 1, 6)), ((4, 1, 8, 1), (4, 3, 8, 63))):
        a, b = _rand(rng, sa, dt), _rand(rng, sb, dt)
        for op, f in ((0, dsc.add), (1, dsc.sub), (2, dsc.mul), (3, dsc.true_div)):
            got, want = f(dsc.from_numpy(a), dsc.from_numpy(b)).numpy(), port.binary(a, b, op)
            if op >= 2:
                assert_close(got, want, what=f'op {op} {sa} {sb}')
            else:
                assert np.array_equal(got, want), (sa, sb, op)


def test_unary_and_cast_packed_and_tails(dsc):
    """abs / angle / conj / real / imag and casts on element counts that are and are not multiples of the pack width."""
    from oracle import port
    rng = np.random.default_rng(23)
    dts = (np.float32, np.float64, np.complex64, np.complex128)
    for n in (4096, 4099, 3):
        for dt in dts:
            x = _rand(rng, (n,), dt)
            t = dsc.from_numpy(x)
            for op, f in ((0, dsc.absolute), (1, dsc.angle), (2, dsc.conj), (3, dsc.real), (4, dsc.imag)):
                got, want = f(t).numpy(), port.unary(x, op)
                assert got.dtype == want.dtype, (dt, op)
                if op in (0, 1):
                    assert_close(got, want, what=f'unary {op} {dt} n={n}')       # sqrt / atan2: library functions
                else:
                    assert np.array_equal(got, want), (dt, op, n)
            for to, dto in ((dsc.Dtype.F32, np.float32), (dsc.Dtype.F64, np.float64), (dsc.Dtype.C32, np.complex64), (dsc.Dtype.C64, np.complex128)):
                assert np.array_equal(t.cast(to).numpy(), port.cast(x, dto)), (dt, dto, n)


def test_reduce_large_rows(dsc):
    """Row-reduction kernel (inner == 1, tree order) and sequential kernel on bigger inputs."""
    from oracle import port
    rng = np.random.default_rng(10)
    for dt in (np.float32, np.complex64, np.float64):
        x = rng.standard_normal((37, 5000)).astype(dt)
        if np.dtype(dt).kind == 'c':
            x = (x + 1j * rng.standard_normal(x.shape)).astype(dt)
        for axis in (0, 1):
            for name, op in (('sum', port.SUM), ('mean', port.MEAN)):
                got = getattr(dsc, name)(dsc.from_numpy(x), axis=axis).numpy()
                want = port.reduce(x, op, axis, True)
                tol = 2e-5 if dt != np.float64 else 1e-12     # different summation order on axis=1
                assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(x)) * np.sqrt(x.shape[axis]), (name, dt, axis)
            xq = (np.round(x.real * 4) / 4 + (1j * x.imag if np.dtype(dt).kind == 'c' else 0)).astype(dt)
            for name, op in (('max', port.MAX), ('min', port.MIN)):
                got = getattr(dsc, name)(dsc.from_numpy(xq), axis=axis, keepdims=False).numpy()
                assert np.array_equal(got, port.reduce(xq, op, axis, False)), (name, dt, axis)


def test_errors_abort_like_the_reference():
    """Invalid arguments print and exit(EXIT_FAILURE) (dsc.h:14-28): rfft of a complex tensor
    (dsc.cpp:2211), irfft of a real one (:2215), shape mismatch of `out` (:2221-2223)."""
    cases = {
        'rfft_complex': 'dsc.rfft(dsc.from_numpy(np.ones(8, np.complex64)))',
        'irfft_real': 'dsc.irfft(dsc.from_numpy(np.ones(9, np.float32)))',
        'bad_out': 'dsc.rfft(dsc.from_numpy(np.ones(8, np.float32)), out=dsc.empty((4,), dsc.Dtype.C32))',
        'no_broadcast': 'dsc.mul(dsc.from_numpy(np.ones((2, 3), np.float32)), dsc.from_numpy(np.ones((2, 4), np.float32)))',
    }
    for name, stmt in cases.items():
        code = f'import numpy as np, dsc_amd as dsc\ndsc.init(1 << 26, 1 << 26)\n{stmt}\ndsc.synchronize()\nprint("SURVIVED")'
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        assert r.returncode != 0 and 'SURVIVED' not in r.stdout, (name, r.stdout, r.stderr)
        assert 'RFFT input must be real' in r.stderr or 'IRFFT input must be complex' in r.stderr or 'DSC_ASSERT' in r.stderr, name


def test_reduce_segmented_axis(dsc):
    """Few outputs, long axis (e.g. the mean spectrum over a batch): the axis is cut into segments
    reduced in parallel.  Sums within tolerance, max/min exact including the tie rules."""
    from oracle import port
    rng = np.random.default_rng(12)
    for dt in (np.complex64, np.float32, np.float64):
        x = rng.standard_normal((3000, 700)).astype(dt)
        if np.dtype(dt).kind == 'c':
            x = (x + 1j * rng.standard_normal(x.shape)).astype(dt)
        for name, op in (('sum', port.SUM), ('mean', port.MEAN)):
            got = getattr(dsc, name)(dsc.from_numpy(x), axis=0).numpy()
            want = port.reduce(x, op, 0, True)
            tol = 2e-5 if dt != np.float64 else 1e-12
            assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(x)) * np.sqrt(x.shape[0]), (name, dt)
        xq = (np.round(x.real * 2) / 2 + (1j * x.imag if np.dtype(dt).kind == 'c' else 0)).astype(dt)
        for name, op in (('max', port.MAX), ('min', port.MIN)):
            got = getattr(dsc, name)(dsc.from_numpy(xq), axis=0, keepdims=False).numpy()
            assert np.array_equal(got, port.reduce(xq, op, 0, False)), (name, dt)


def test_golden_indexing_and_slicing(dsc, golden):
    """dsc_tensor_get_idx / get_slice / set_idx / set_slice on the device against the reference's
    outputs (tests/golden/slice.npz): exact."""
    from tests.helpers import decode_sel
    n = 0
    for rec, xs, y in golden.cases('slice'):
        key = decode_sel(rec['sel'])
        key1 = key[0] if len(key) == 1 else key
        t = dsc.from_numpy(xs[0])
        if rec['op'].startswith('get'):
            got = t[key1]
            got = got.numpy() if isinstance(got, dsc.Tensor) else np.array([got], dtype=y.dtype)
        else:
            t[key1] = dsc.from_numpy(xs[1])
            got = t.numpy()
        assert got.dtype == y.dtype and got.shape == y.shape, (rec['key'], got.shape, y.shape)
        assert np.array_equal(got, y), rec['key']
        n += 1
    assert n == 100


def test_slicing_random_keys_and_wide_rows(dsc):
    """Random keys against the oracle, plus the shapes the hot path uses: the `[:output_length]` crop of
    irfft rows (README.md:133), strided / reversed wide rows and block placement into a padded buffer."""
    from oracle import indexing as ix
    rng = np.random.default_rng(5)
    n_ok = 0
    for trial in range(200):
        nd = int(rng.integers(1, 5))
        shape = tuple(int(v) for v in rng.integers(1, 9, nd))
        dt = (np.float32, np.float64, np.complex64, np.complex128)[trial % 4]
        x = rng.standard_normal(shape).astype(dt)
        key = []
        for d in range(int(rng.integers(1, nd + 1))):
            if rng.random() < 0.25:
                key.append(int(rng.integers(-shape[d], shape[d])))
            else:
                f = lambda: None if rng.random() < 0.5 else int(rng.integers(-shape[d], shape[d] + 1))   # noqa: E731
                key.append(slice(f(), f(), None if rng.random() < 0.5 else int(rng.integers(-3, 4))))
        try:
            want = ix.get_slice(x, *key)
        except ix.Abort:
            continue
        k1 = key[0] if len(key) == 1 else tuple(key)
        got = dsc.from_numpy(x)[k1]
        got = got.numpy() if isinstance(got, dsc.Tensor) else np.array([got], dtype=dt)
        if want.size == got.size and want.shape != got.shape:      # all-int key: the wrapper calls get_idx (1-element tensor)
            want = want.reshape(got.shape)
        assert got.shape == want.shape and np.array_equal(got, want), (shape, key)
        v = rng.standard_normal(1 if trial % 2 else int(rng.integers(1, 9))).astype(dt)
        try:
            want_set = ix.set_slice(x, v, *key)
        except ix.Abort:
            continue
        t = dsc.from_numpy(x)
        if all(isinstance(k, int) for k in key):
            continue                                               # int-only keys go to set_idx (its own shape rule)
        t[k1] = dsc.from_numpy(v)
        assert np.array_equal(t.numpy(), want_set), (shape, key, v.shape)
        n_ok += 1
    assert n_ok > 40
    # wide rows (row kernel) and the filterFFT crop
    y = rng.standard_normal((37, 4096)).astype(np.float32)
    t = dsc.from_numpy(y)
    assert np.array_equal(t[:, :3001].numpy(), y[:, :3001])
    assert np.array_equal(t[::-2, 5:4000:3].numpy(), y[::-2, 5:4000:3])
    assert np.array_equal(t[3:30, ::-1].numpy(), y[3:30, ::-1])
    pad = dsc.from_numpy(np.zeros((37, 8192), np.float32))
    pad[:, 100:4196] = t                                           # place a block into a zero-padded buffer
    want = np.zeros((37, 8192), np.float32)
    want[:, 100:4196] = y
    assert np.array_equal(pad.numpy(), want)
    z = (rng.standard_normal((5, 2049)) + 1j * rng.standard_normal((5, 2049))).astype(np.complex128)
    assert np.array_equal(dsc.from_numpy(z)[1:4, 1:-1].numpy(), z[1:4, 1:-1])
    # the whole README pipeline on the device: irfft(rfft(s) * rfft(b))[:, :ls + lb - 1]
    s = rng.standard_normal((4, 3000)).astype(np.float32)
    b = rng.standard_normal(97).astype(np.float32)
    n = 4096
    out = dsc.irfft(dsc.rfft(dsc.from_numpy(s), n=n) * dsc.rfft(dsc.from_numpy(b), n=n))[:, :3000 + 97 - 1]
    ref_full = np.stack([np.convolve(r.astype(np.float64), b.astype(np.float64)) for r in s])
    assert out.shape == (4, 3096) and np.abs(out.numpy() - ref_full).max() <= 1e-3


def test_indexing_errors_abort_like_the_reference():
    """Out-of-range / empty / mismatched selections exit (dsc.cpp:849, 926-931, 1050, 1146)."""
    cases = {
        'idx_range': 't[3]',
        'slice_too_long': 't[0:9]',
        'slice_empty': 't[2:1]',
        'too_many': 't[0, 0, 0]',
        'set_shape': 't[:] = dsc.from_numpy(np.zeros((2, 4), np.float32))',
        'past_end': 't[2:5]',
    }
    for name, stmt in cases.items():
        code = ('import numpy as np, dsc_amd as dsc\ndsc.init(1 << 26, 1 << 26)\n'
                f't = dsc.from_numpy(np.zeros((3, 4), np.float32))\n{stmt}\ndsc.synchronize()\nprint("SURVIVED")')
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        assert r.returncode != 0 and 'SURVIVED' not in r.stdout, (name, r.stdout, r.stderr)
        assert 'dsc_tensor_' in r.stderr or 'DSC_ASSERT' in r.stderr, (name, r.stderr)      # "<entry point>: <the rule that failed>"


def test_golden_transpose_and_fftfreq(dsc, golden):
    """dsc_transpose (tiled last-two-axes kernel and the general permutation) and dsc_fftfreq / dsc_rfftfreq against
    the reference's outputs: exact."""
    n = 0
    for rec, xs, y in golden.cases('layout'):
        if rec['op'] == 'transpose':
            got = dsc.transpose(dsc.from_numpy(xs[0]), rec['axes']).numpy()
        else:
            dt = dsc.Dtype.F32 if y.dtype == np.float32 else dsc.Dtype.F64
            got = getattr(dsc, rec['op'])(rec['n'], rec['d'], dt).numpy()
        assert got.dtype == y.dtype and got.shape == y.shape, rec['key']
        assert np.array_equal(got, y), rec['key']
        n += 1
    assert n == 56
    rng = np.random.default_rng(8)
    for shape in ((1000, 37), (33, 4097), (3, 130, 65)):                     # ragged tiles
        for dt in (np.float32, np.complex64, np.complex128):
            x = rng.standard_normal(shape).astype(dt)
            ax = tuple(range(len(shape) - 2)) + (len(shape) - 1, len(shape) - 2)
            assert np.array_equal(dsc.transpose(dsc.from_numpy(x), ax).numpy(), x.transpose(ax))
    # a transform along axis 0 equals transpose -> last-axis transform -> transpose
    x = rng.standard_normal((256, 24)).astype(np.float32)
    a = dsc.rfft(dsc.from_numpy(x), axis=0).numpy()
    b = dsc.transpose(dsc.rfft(dsc.transpose(dsc.from_numpy(x)))).numpy()
    assert rel_l2(a, b) <= 1e-6


def test_general_broadcast_with_long_rows(dsc):
    """Broadcast patterns that are neither equal shapes nor a trailing-dims operand (column vectors, operands broadcast along several
    axes, both operands smaller than the result) on results with a long innermost axis: the row-wise kernel (one division chain per
    block).  Every operator, all four dtypes, against the oracle."""
    from oracle import port
    rng = np.random.default_rng(91)
    ops = ((dsc.add, port.ADD), (dsc.sub, port.SUB), (dsc.mul, port.MUL), (dsc.true_div, port.DIV))
    for dt in (np.float32, np.float64, np.complex64, np.complex128):
        for sa, sb in (((37, 1000), (37, 1)), ((5, 3, 70, 200), (5, 1, 70, 1)), ((3, 1, 129), (4, 1, 50, 1)), ((1, 2500), (6, 1)), ((6, 1), (1, 2500)),
                       ((2, 3, 4, 64), (3, 1, 64))):
            a = rng.standard_normal(sa).astype(dt)
            b = (rng.standard_normal(sb) + 2.5).astype(dt)
            if np.dtype(dt).kind == 'c':
                a = (a + 1j * rng.standard_normal(sa)).astype(dt)
                b = (b + 1j * rng.standard_normal(sb)).astype(dt)
            for f, op in ops:
                assert_close(f(dsc.from_numpy(a), dsc.from_numpy(b)).numpy(), port.binary(a, b, op), what=f'{f.__name__} {np.dtype(dt).name} {sa} x {sb}')


def test_transpose_every_permutation_and_aligned_slices(dsc):
    """dsc_transpose for every permutation of 2 .. 4 axes (dsc.cpp:764-827; permutations that move the last axis go through 32 x 32 LDS
    tiles, the others through the strided copy) and slices whose rows start on 16-byte boundaries (moved 16 bytes per lane):
    exact against numpy, odd extents, every dtype."""
    import itertools
    rng = np.random.default_rng(90)
    for dt in (np.float32, np.float64, np.complex64, np.complex128):
        # extents that are multiples of 4: the 16-byte tile kernel (transpose_plane_vec_kernel), with partial tiles
        for shape in ((33, 65), (3, 37, 41), (2, 3, 35, 37), (5, 33, 4, 66), (68, 132), (4, 72, 136), (2, 4, 36, 68), (8, 100)):
            x = rng.standard_normal(shape).astype(dt)
            if np.dtype(dt).kind == 'c':
                x = (x + 1j * rng.standard_normal(shape)).astype(dt)
            t = dsc.from_numpy(x)
            assert np.array_equal(dsc.transpose(t).numpy(), x.transpose())
            for perm in itertools.permutations(range(len(shape))):
                got = dsc.transpose(t, perm).numpy()
                assert got.shape == x.transpose(perm).shape and np.array_equal(got, x.transpose(perm)), (dt, shape, perm)
        y = rng.standard_normal((37, 4096)).astype(dt)
        ty = dsc.from_numpy(y)
        for key in ((slice(None), slice(0, 4000)), (slice(None, None, 2),), (slice(3, 30), slice(16, 4080)), (slice(None), slice(1, 4001)),
                    (slice(None), slice(None, None, 2)), (5,)):
            assert np.array_equal(ty[key].numpy(), y[key]), (dt, key)
        z = dsc.from_numpy(y.copy())
        z[:, 16:4016] = dsc.from_numpy(np.ascontiguousarray(y[:, :4000]))
        w = y.copy(); w[:, 16:4016] = y[:, :4000]
        assert np.array_equal(z.numpy(), w)


def test_tracing_records_host_calls_and_device_spans(dsc, tmp_path):
    """dsc.profile() (python/dsc/profiler.py:58-63): the dump is a Perfetto JSON array with the reference's fields; every
    operator call appears as a B/E pair on the host track and as a complete event with a positive duration on the HIP
    stream track, in call order; nothing is recorded outside the context manager."""
    import json
    x = dsc.from_numpy(np.random.default_rng(0).standard_normal((64, 4096)).astype(np.float32))
    dsc.rfft(x)                                            # not recorded
    path = str(tmp_path / 'traces.json')
    with dsc.profile(path):
        X = dsc.rfft(x)
        P = X * X
        y = dsc.irfft(P)
        m = dsc.mean(y, axis=0)
        c = y[:, :100]
    dsc.rfft(x)                                            # not recorded
    ev = json.load(open(path))
    host = [e for e in ev if e.get('tid') == 0 and e['ph'] in 'BE']
    gpu = [e for e in ev if e.get('tid') == 1 and e['ph'] == 'X']
    names = [e['name'] for e in host if e['ph'] == 'B']
    assert names == ['dsc_rfft', 'dsc_mul', 'dsc_irfft', 'dsc_mean', 'dsc_tensor_get_slice'], names
    assert [e['name'] for e in gpu] == names
    assert all(e['dur'] > 0 for e in gpu)
    assert all(set(('name', 'cat', 'ph', 'ts', 'pid', 'tid')) <= set(e) for e in host)
    b = [e for e in host if e['ph'] == 'B']
    e_ = [e for e in host if e['ph'] == 'E']
    assert len(b) == len(e_) and all(x1['ts'] <= x2['ts'] for x1, x2 in zip(b, e_))
    assert b[0]['args']['x']['shape'] == [64, 4096] and b[0]['args']['x']['dtype'] == 'f32'
    # cleared: a second session starts empty
    with dsc.profile(path):
        dsc.rfft(x)
    assert sum(1 for e in json.load(open(path)) if e['ph'] == 'B') == 1
    del X, P, y, m, c
