"""The C restatement against the reference build itself (oracle/_ref/libdsc_ref.so),
on fresh random inputs.  Skipped where _ref has not been built.  CPU only."""
import numpy as np
import pytest

from oracle import port, ref

pytestmark = pytest.mark.skipif(not ref.available(), reason='oracle/_ref not built (needs /root/reference)')

DTS = [np.float32, np.float64, np.complex64, np.complex128]


def rnd(rng, shape, dt):
    x = rng.standard_normal(shape)
    if np.dtype(dt).kind == 'c':
        x = x + 1j * rng.standard_normal(shape)
    return x.astype(dt)


def same(a, b, dt):
    """Same flags, same operation order: expect identity; allow a few ulp in case the two
    compilers contract differently."""
    assert a.shape == b.shape and a.dtype == b.dtype
    tol = 2e-6 if np.dtype(dt).itemsize in (4,) or np.dtype(dt) == np.complex64 else 4e-15
    den = max(np.max(np.abs(b)), 1e-300)
    assert np.max(np.abs(a - b)) / den <= tol


def test_fft_all_axes_like_reference_test():
    """Mirror of python/tests/test_ops.py:458-489 (every axis of an [8,8,8,8] tensor, crop /
    copy / pad), for f64 as there and for f32 as well."""
    R = ref.Ref.get()
    rng = np.random.default_rng(5)
    for dt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
        n_ = 5
        for axis in range(4):
            shape = [8] * 4
            shape[axis] = 2 ** n_
            for change in (-1, 0, 1):
                n = 2 ** (n_ + change)
                x = rnd(rng, shape, dt)
                same(port.rfft(x, n, axis), R.rfft(x, n, axis), dt)
                X = R.rfft(x, n, axis)
                same(port.irfft(X, -1, axis), R.irfft(X, -1, axis), dt)
                xc = rnd(rng, shape, cdt)
                same(port.fft(xc, n, axis), R.fft(xc, n, axis), cdt)
                same(port.ifft(xc, n, axis), R.ifft(xc, n, axis), cdt)
                same(port.fft(x, n, axis), R.fft(x, n, axis), cdt)


def test_headline_size():
    R = ref.Ref.get()
    rng = np.random.default_rng(6)
    x = rnd(rng, (3, 65536), np.float32)
    X = R.rfft(x)
    same(port.rfft(x), X, np.float32)
    same(port.irfft(X), R.irfft(X), np.float32)


def test_mul_reduce_cast():
    R = ref.Ref.get()
    rng = np.random.default_rng(7)
    for da in DTS:
        for db in DTS:
            for sa, sb in (((4, 5), (4, 5)), ((4, 5), (5,)), ((4, 5), (1,)), ((1,), (4, 5)), ((3, 1, 5), (1, 4, 1))):
                a, b = rnd(rng, sa, da), rnd(rng, sb, db)
                same(port.mul(a, b), R.mul(a, b), np.result_type(da))
                for op in (port.ADD, port.SUB, port.DIV):
                    same(port.binary(a, b, op), R.binary(a, b, op), np.result_type(da))
            x = rnd(rng, (3, 4), da)
            assert np.array_equal(port.cast(x, db), R.cast(x, db))
        for op in range(5):
            xu = rnd(rng, (5, 7), da)
            pu, ru = port.unary(xu, op), R.unary(xu, op)
            assert pu.dtype == ru.dtype and np.array_equal(pu, ru)
    for dt in DTS:
        for axis in range(-3, 3):
            for keep in (True, False):
                for op in range(4):
                    x = rnd(rng, (7, 5, 3), dt)
                    same(port.reduce(x, op, axis, keep), R.reduce(x, op, axis, keep), dt)


def test_plan_table_layout():
    """dsc_fft.h:33-55 / 109-135: table size and the stage layout."""
    tw = port.plan_table(32768, np.float32, 0)
    assert tw.size == 131070                      # 524,280 B
    assert tw[0] == 1.0 and tw[1] == 0.0          # stage m=2, k=0
    base = 2 * (32768 - 1)                        # stage m=65536 used by the real post-pass
    k = np.arange(32768)
    want = np.exp(-2j * np.pi * k / 65536)
    got = tw[base::2] + 1j * tw[base + 1::2]
    assert np.max(np.abs(got - want)) < 3e-7


def test_indexing_random_keys():
    """oracle/indexing.py against the reference on random keys (every call the oracle accepts is
    replayed on the reference; refused ones would exit the process there)."""
    from oracle import indexing as ix
    R = ref.Ref.get()
    rng = np.random.default_rng(77)
    n_ok = 0
    for trial in range(300):
        nd = int(rng.integers(1, 5))
        shape = tuple(int(v) for v in rng.integers(1, 7, nd))
        dt = DTS[trial % 4]
        x = rnd(rng, shape, dt)
        key = []
        for d in range(int(rng.integers(1, nd + 1))):
            if rng.random() < 0.25:
                key.append(int(rng.integers(-shape[d], shape[d])))
            else:
                f = lambda: None if rng.random() < 0.4 else int(rng.integers(-shape[d] - 1, shape[d] + 2))   # noqa: E731
                step = None if rng.random() < 0.5 else int(rng.integers(-3, 4))
                key.append(slice(f(), f(), step))
        try:
            want_get = ix.get_slice(x, *key)
        except ix.Abort:
            continue
        got = R.get_slice(x, *key)
        assert got.shape == want_get.shape and np.array_equal(got, want_get), (shape, key)
        v = rnd(rng, (1,), dt) if trial % 2 else rnd(rng, (int(rng.integers(1, 9)),), dt)
        try:
            want_set = ix.set_slice(x, v, *key)
        except ix.Abort:
            continue
        assert np.array_equal(R.set_slice(x, v, *key), want_set), (shape, key, v.shape)
        n_ok += 1
    assert n_ok > 60


def test_max_min_with_nans_restatement_equals_reference():
    """NaN behaviour of max / min follows from the reference's predicates (dsc_ops.h:318-339, dsc.h:43-44): the restatement
    must reproduce it exactly, since the GPU tests use it as the oracle for NaN inputs (tests/test_gpu_lifecycle.py)."""
    R = ref.Ref.get()
    rng = np.random.default_rng(21)
    for dt in (np.float32, np.float64, np.complex64, np.complex128):
        x = rng.standard_normal((5, 7, 9))
        if np.dtype(dt).kind == 'c':
            x = x + 1j * rng.standard_normal((5, 7, 9))
        x = x.astype(dt)
        flat = x.reshape(-1)
        flat[rng.choice(flat.size, 40, replace=False)] = np.nan
        x[:, 3, 4] = np.nan                       # an all-NaN line along axis 0
        x[2, :, -1] = np.nan                      # NaN as the last element along axis 2
        for op in (port.MAX, port.MIN):
            for axis in (0, 1, 2):
                a, b = port.reduce(x, op, axis), R.reduce(x, op, axis)
                assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=0.0), np.nan_to_num(b, nan=0.0)), (dt, op, axis)


def test_length_one_real_transforms_abort_in_the_reference_and_are_refused_by_the_restatement():
    """rfft of one sample (order 0) and irfft of one bin reach dsc_pow2_n(0) / dsc_plan_fft(0), whose DSC_ASSERT(n > 0) ends the
    process (dsc.h:122-132, dsc.cpp:2195-2200): the restatement's shape rule reports that instead of inventing a result."""
    import subprocess
    import sys
    cases = {'rfft_len1': ('rfft', 'np.ones((3, 1), np.float32)', -1), 'rfft_n1': ('rfft', 'np.ones((3, 8), np.float64)', 1),
             'irfft_1bin': ('irfft', 'np.ones((2, 1), np.complex64)', -1), 'irfft_n1': ('irfft', 'np.ones((2, 5), np.complex128)', 1)}
    for name, (op, arr, n) in cases.items():
        code = f'import numpy as np\nfrom oracle import ref\nr = ref.Ref.get(1 << 26, 1 << 24)\nr.{op}({arr}, {n}, -1)\nprint("SURVIVED")'
        p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        assert p.returncode != 0 and 'SURVIVED' not in p.stdout, (name, p.stdout, p.stderr[-300:])
        with pytest.raises(ValueError):
            getattr(port, op)(eval(arr), n, -1)
    # the shortest lengths that do work
    assert port.rfft(np.ones((3, 2), np.float32)).shape == (3, 2)
    assert port.irfft(np.ones((3, 2), np.complex64)).shape == (3, 2)
