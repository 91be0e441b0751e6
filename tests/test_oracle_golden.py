"""The CPU restatement (oracle/dsc_oracle.c) against the committed fixtures, which are
outputs of the reference itself (tests/golden/make_golden.py).  CPU only.

The restatement follows the reference's operation order and is compiled with the same
FP flags, so on the host that generated the fixtures it is bit-identical; across hosts
FMA contraction may differ, hence a tolerance of a few ulp rather than equality."""
import numpy as np
import pytest

from oracle import port
from tests.helpers import assert_close

ULPS = {np.dtype(np.float32): 2e-6, np.dtype(np.complex64): 2e-6,
        np.dtype(np.float64): 4e-15, np.dtype(np.complex128): 4e-15}
RED = {'sum': port.SUM, 'mean': port.MEAN, 'max': port.MAX, 'min': port.MIN}


def _check(got, want, what):
    assert_close(got, want, tol=ULPS[want.dtype], what=what)


@pytest.mark.parametrize('group', ['fft_small', 'fft_large'])
def test_fft_family(golden, group):
    n = 0
    for rec, xs, y in golden.cases(group):
        got = getattr(port, rec['op'])(xs[0], rec['n'], rec['axis'])
        _check(got, y, f"{rec['key']} n={rec['n']} axis={rec['axis']} in={xs[0].shape}")
        n += 1
    assert n > 0


def test_known_answers(golden):
    """Impulse / constant / tone / Nyquist: analytic spectra, independent of any code."""
    seen = 0
    for rec, xs, y in golden.cases('fft_small', 'rfft'):
        if 'kat' not in rec:
            continue
        x = xs[0]
        N = x.shape[-1]
        want = np.zeros(N // 2 + 1, dtype=np.complex128)
        if rec['kat'] == 'impulse':
            want[:] = 1
        elif rec['kat'] == 'impulse3':
            want[:] = np.exp(-2j * np.pi * 3 * np.arange(N // 2 + 1) / N)
        elif rec['kat'] == 'const':
            want[0] = N
        elif rec['kat'] == 'tone5':
            want[5] = N / 2
        elif rec['kat'] == 'nyquist':
            want[N // 2] = N
        got = port.rfft(x)
        tol = 1e-5 if x.dtype == np.float32 else 1e-12
        assert np.max(np.abs(got - want)) <= tol * N, rec['kat']
        assert np.max(np.abs(y - want)) <= tol * N, rec['kat']
        seen += 1
    assert seen == 10


def test_mul(golden):
    for rec, xs, y in golden.cases('mul'):
        got = port.mul(xs[0], xs[1])
        _check(got, y, f"{rec['key']} {xs[0].dtype}{xs[0].shape} x {xs[1].dtype}{xs[1].shape}")


def test_reductions(golden):
    for rec, xs, y in golden.cases('reduce'):
        got = port.reduce(xs[0], RED[rec['op']], rec['axis'], rec['keepdims'])
        if rec['op'] in ('max', 'min'):
            assert got.shape == y.shape and np.array_equal(got, y), rec['key']     # selection: exact
        else:
            _check(got, y, f"{rec['key']} axis={rec['axis']} keep={rec['keepdims']}")


def test_filter_pipeline(golden):
    """README filterFFT (README.md:113-135) as four reference ops."""
    for rec, xs, y in golden.cases('filter'):
        s, b = xs
        n = rec['n']
        got = port.irfft(port.mul(port.rfft(s, n), port.rfft(b, n)))
        _check(got, y, rec['key'])
        # and it IS a linear convolution
        lin = np.convolve(s.astype(np.float64), b.astype(np.float64))
        assert np.max(np.abs(got[:len(lin)] - lin)) <= 1e-4 * np.max(np.abs(lin))


def test_semantics_quirks():
    """lengths round up to a power of two; irfft(n=) counts bins."""
    x = np.random.default_rng(3).standard_normal(1000).astype(np.float32)
    assert port.rfft(x).shape == (513,)
    assert port.rfft(x, 600).shape == (513,)
    X = port.rfft(x)
    assert port.irfft(X).shape == (1024,)
    assert port.irfft(X, 513).shape == (1024,)
    assert port.irfft(X, 257).shape == (512,)
    assert port.irfft(X, 1024).shape == (2048,)
    with pytest.raises(ValueError):
        port.rfft(X)          # complex input: the reference aborts
    with pytest.raises(ValueError):
        port.irfft(x)         # real input: the reference aborts
    assert port.mul(np.ones(3, np.float64), np.ones(3, np.complex64)).dtype == np.complex64


def test_add_sub_div(golden):
    ops = {'add': port.ADD, 'sub': port.SUB, 'div': port.DIV}
    n = 0
    for rec, xs, y in golden.cases('binary'):
        got = port.binary(xs[0], xs[1], ops[rec['op']])
        _check(got, y, f"{rec['key']} {xs[0].dtype}{xs[0].shape} {rec['op']} {xs[1].dtype}{xs[1].shape}")
        n += 1
    assert n == 60


def test_unary_spectrum_consumers(golden):
    ops = {'abs': port.ABS, 'angle': port.ANGLE, 'conj': port.CONJ, 'real': port.REALPART, 'imag': port.IMAGPART}
    n = 0
    for rec, xs, y in golden.cases('unary'):
        got = port.unary(xs[0], ops[rec['op']])
        assert got.dtype == y.dtype and got.shape == y.shape, rec['key']
        assert np.allclose(got, y, rtol=2e-6 if y.dtype == np.float32 or y.dtype == np.complex64 else 4e-15, atol=1e-30), rec['key']
        n += 1
    assert n == 40


def test_indexing_and_slicing(golden):
    """oracle/indexing.py (restatement of dsc.cpp:829-1169) against the reference's outputs: exact."""
    from oracle import indexing as ix
    from tests.helpers import decode_sel
    n = 0
    for rec, xs, y in golden.cases('slice'):
        key = decode_sel(rec['sel'])
        got = getattr(ix, rec['op'])(*xs, *key)
        assert got.dtype == y.dtype and got.shape == y.shape, rec['key']
        assert np.array_equal(got, y), rec['key']
        n += 1
    assert n == 100


def test_indexing_aborts_where_the_reference_asserts():
    from oracle import indexing as ix
    x = np.zeros((3, 4), np.float32)
    for bad in ((slice(0, 9),), (slice(2, 1),), (slice(None, None, 0),), (slice(None), slice(None), 0)):
        with pytest.raises(ix.Abort):
            ix.get_slice(x, *bad)
    with pytest.raises(ix.Abort):
        ix.get_idx(x, 3)
    with pytest.raises(ix.Abort):
        ix.set_slice(x, np.zeros((2, 4), np.float32), slice(None))          # dsc.cpp:1146
    with pytest.raises(ix.Abort):
        ix.set_idx(x, np.zeros(4, np.float64), 0)                            # dtype mismatch, dsc.cpp:1050


def test_transpose_and_fftfreq(golden):
    from oracle import indexing as ix
    n = 0
    for rec, xs, y in golden.cases('layout'):
        if rec['op'] == 'transpose':
            got = ix.transpose(xs[0], rec['axes'])
        else:
            got = getattr(ix, rec['op'])(rec['n'], rec['d'], y.dtype)
        assert got.dtype == y.dtype and got.shape == y.shape, rec['key']
        assert np.array_equal(got, y), rec['key']
        n += 1
    assert n == 56
