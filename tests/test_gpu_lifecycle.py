"""GPU tests of the pieces around the kernels: the FFT plan cache behind the exported dsc_plan_fft (dsc/src/dsc.cpp:182-267),
handle lifetime (double frees, handles that outlive dsc.clear(): dsc/src/dsc.cpp:287-303), NaN semantics of max / min
(dsc_ops.h:318-339), the tall-skinny reduction and an axis-0 transform in an arena with no room for the transpose route."""
import ctypes
import gc
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import assert_close, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
REAL, COMPLEX = 0, 1


@pytest.fixture(scope='module')
def dsc():
    import dsc_amd
    try:
        dsc_amd.init(12 << 30, 4 << 30)
    except RuntimeWarning:
        pass
    yield dsc_amd


def test_plan_fft_hit_eviction_and_clear(dsc):
    """dsc_plan_fft through the C ABI: a hit returns the same handle (keyed on pow2ceil(n), type, twiddle precision), the
    17th distinct plan evicts the least recently used one, results stay right after an eviction and after dsc_ctx_clear."""
    from dsc_amd import _bindings as B
    from dsc_amd.context import _get_ctx
    D = dsc.Dtype
    dsc.clear()
    ctx = _get_ctx()
    base = dsc.used_mem()
    h = B.dsc_plan_fft(ctx, 1024, COMPLEX, D.F64.value)
    assert h and B.dsc_plan_fft(ctx, 1000, COMPLEX, D.C64.value) == h          # rounded up to 1024, C64 -> f64 twiddles
    assert B.dsc_plan_fft(ctx, 1024, COMPLEX, D.F32.value) != h and B.dsc_plan_fft(ctx, 1024, REAL, D.F64.value) != h
    dsc.clear()
    assert dsc.used_mem() == base

    # 16 slots: one big plan (65536 complex f64 = 1 MiB of roots) first, then 15 small ones
    big = B.dsc_plan_fft(ctx, 65536, COMPLEX, D.F64.value)
    with_big = dsc.used_mem()
    assert with_big - base >= 65536 * 16
    keys = [(n, t, d) for n in (8, 16, 32, 64) for t in (REAL, COMPLEX) for d in (D.F32, D.F64)][:15]
    small = [B.dsc_plan_fft(ctx, n, t, d.value) for n, t, d in keys]
    assert len(set(small + [big])) == 16
    full = dsc.used_mem()
    assert B.dsc_plan_fft(ctx, 65536, COMPLEX, D.F64.value) == big and dsc.used_mem() == full      # a hit: touched, nothing built
    # 17th plan: the least recently used is now keys[0] (the big one was just touched), so the big block must survive
    B.dsc_plan_fft(ctx, 128, REAL, D.F32.value)
    assert dsc.used_mem() > full - 65536 * 16
    assert B.dsc_plan_fft(ctx, 65536, COMPLEX, D.F64.value) == big
    # ... and touching everything but the big plan makes IT the eviction victim
    for n, t, d in keys[1:]:
        B.dsc_plan_fft(ctx, n, t, d.value)
    B.dsc_plan_fft(ctx, 128, REAL, D.F32.value)
    before = dsc.used_mem()
    B.dsc_plan_fft(ctx, 256, REAL, D.F32.value)            # 18th distinct key
    assert dsc.used_mem() < before - 65536 * 16 + 8192, 'the least recently used (big) plan was not the one evicted'

    # results after evictions, and again after a clear (plans are rebuilt on demand)
    rng = np.random.default_rng(5)
    z = rng.standard_normal((3, 65536)) + 1j * rng.standard_normal((3, 65536))
    for _ in range(2):
        got = dsc.fft(dsc.from_numpy(z)).numpy()
        assert rel_l2(got, np.fft.fft(z, axis=-1)) <= 1e-12
        x = rng.standard_normal((4, 1024)).astype(np.float32)
        assert rel_l2(dsc.rfft(dsc.from_numpy(x)).numpy(), np.fft.rfft(x.astype(np.float64), axis=-1)) <= 1e-5
        dsc.clear()
    assert dsc.used_mem() == base


def test_double_free_and_stale_handles_cannot_free_a_live_tensor(dsc):
    """a second dsc_tensor_free on an old pointer, or a Python handle collected after dsc.clear(), must never
    release a NEW tensor's block (header addresses are quarantined; Python handles remember their clear() epoch)."""
    from dsc_amd import _bindings as B
    from dsc_amd.context import _get_ctx
    ctx = _get_ctx()
    dsc.clear()
    base = dsc.used_mem()
    t = B.dsc_tensor_1d(ctx, dsc.Dtype.F32.value, 1024)
    addr = ctypes.addressof(t.contents)
    B.dsc_tensor_free(ctx, t)
    t2 = B.dsc_tensor_1d(ctx, dsc.Dtype.F32.value, 1024)
    assert ctypes.addressof(t2.contents) != addr, 'a freed header address was handed out again at once'
    used = dsc.used_mem()
    B.dsc_tensor_free(ctx, t)                              # stale second free: ignored
    assert dsc.used_mem() == used and t2.contents.data
    B.dsc_tensor_free(ctx, t2)
    assert dsc.used_mem() == base

    a = dsc.from_numpy(np.arange(4096, dtype=np.float32))
    dsc.clear()                                            # `a` is dead now (dsc.cpp:287-291) but the Python object lives on
    want = np.arange(4096, dtype=np.float32)[::-1].copy()
    b = dsc.from_numpy(want)
    used = dsc.used_mem()
    del a
    gc.collect()
    assert dsc.used_mem() == used
    c = dsc.from_numpy(np.zeros(4096, np.float32))         # would land on b's block if it had been released
    assert np.array_equal(b.numpy(), want)
    del b, c
    gc.collect()
    assert dsc.used_mem() == base


@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.complex64, np.complex128])
def test_max_min_with_nans_follow_the_reference(dsc, dtype):
    """NaN handling is a consequence of the reference's predicates (dsc_ops.h:318-339): exact agreement with the oracle —
    which tests/test_oracle_vs_ref.py pins against the reference on the same kind of input — on every kernel variant:
    sequential (inner > 1), segmented (few outputs, long axis) and the tree kernel (last axis)."""
    from oracle import port
    rng = np.random.default_rng(11)

    def data(shape):
        x = rng.standard_normal(shape)
        if np.dtype(dtype).kind == 'c':
            x = x + 1j * rng.standard_normal(shape)
        x = x.astype(dtype)
        flat = x.reshape(-1)
        flat[rng.choice(flat.size, max(3, flat.size // 50), replace=False)] = np.nan
        return x

    cases = [((6, 40, 9), (0, 1, 2)), ((4, 3000), (0, 1)), ((3000, 3), (0,)), ((2, 70000), (1,))]
    for shape, axes in cases:
        x = data(shape)
        x[..., -1] = np.where(rng.random(x[..., -1].shape) < 0.3, np.nan, x[..., -1])          # NaN as the LAST element of some rows
        if len(shape) == 2:
            x[0, :] = np.nan                                                                    # an all-NaN line
        for axis in axes:
            for name, op in (('max', port.MAX), ('min', port.MIN)):
                got = getattr(dsc, name)(dsc.from_numpy(x), axis=axis).numpy()
                want = port.reduce(x, op, axis)
                assert got.shape == want.shape
                assert np.array_equal(got.view(np.uint8), want.view(np.uint8)) or np.array_equal(np.isnan(got), np.isnan(want)) and \
                    np.array_equal(np.nan_to_num(got, nan=0.0), np.nan_to_num(want, nan=0.0)), f'{name} {np.dtype(dtype).name} {shape} axis {axis}'


def test_tall_skinny_reduction(dsc):
    """[> 2 M, 2] over axis 0 — few outputs, a very long axis: the segment count used to exceed the grid limit."""
    from oracle import port
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2_300_007, 2)).astype(np.float32)
    X = dsc.from_numpy(x)
    for name, op in (('max', port.MAX), ('min', port.MIN)):
        assert np.array_equal(getattr(dsc, name)(X, axis=0).numpy(), port.reduce(x, op, 0))
    ref_sum = x.astype(np.float64).sum(axis=0, keepdims=True)
    assert np.allclose(dsc.sum(X, axis=0).numpy(), ref_sum, rtol=1e-4, atol=1e-2)
    assert np.allclose(dsc.mean(X, axis=0).numpy(), ref_sum / x.shape[0], rtol=1e-4, atol=1e-6)
    z = (x[:1_000_003] + 1j * x[1_000_003:2_000_006]).astype(np.complex64)
    Z = dsc.from_numpy(z)
    assert np.array_equal(dsc.max(Z, axis=0).numpy(), port.reduce(z, port.MAX, 0))
    assert np.array_equal(dsc.min(Z, axis=0).numpy(), port.reduce(z, port.MIN, 0))


def test_axis0_transform_in_a_tight_arena():
    """an fft along axis 0 of >= 512 points used to take two full-size temporaries from the MAIN arena and exit
    when a context was sized for x and out only.  Own process: the arena size is fixed at init."""
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import dsc_amd as dsc
n, cols = 16384, 512                            # beyond the column kernel: the route with two temporaries would be next
x = np.random.default_rng(3).standard_normal((n, cols)).astype(np.float32)
need = x.nbytes + (n // 2 + 1) * cols * 8
dsc.init(need + (6 << 20), 64 << 20)           # x, out, the plan tables and nothing else
X = dsc.rfft(dsc.from_numpy(x), axis=0)
path = dsc.last_fft_path()
got = X.numpy()
want = np.fft.rfft(x.astype(np.float64), axis=0)
err = np.linalg.norm(got - want) / np.linalg.norm(want)
assert err <= 1e-5, err
# with room, the same call takes the transpose route: same answer
print('OK', path, err)
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    assert 'generic' in r.stdout, r.stdout             # no room for the transposes: the strided LDS kernel took it


def test_fused_l2_falls_back_when_scratch_is_small_or_switched_off():
    """fft_xcd_fused.hip needs one scratch row per possible team (29 MiB for f32 rows of 512 KiB): a context with less scratch, or
    DSC_NO_FUSED_L2 in the environment, takes the two-kernel route — same results.  Own processes: arena sizes and the switch are
    fixed at start."""
    code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
import dsc_amd as dsc
dsc.init(1 << 30, int(sys.argv[1]) << 20)
x = np.random.default_rng(4).standard_normal((9, 131072)).astype(np.float32)
X = dsc.rfft(dsc.from_numpy(x))
path = dsc.last_fft_path()
want = np.fft.rfft(x.astype(np.float64), axis=-1)
err = np.linalg.norm(X.numpy() - want) / np.linalg.norm(want)
back = dsc.irfft(X).numpy()
assert err <= 1e-6 and np.max(np.abs(back - x)) < 1e-4, err
dsc.synchronize()
print('OK', path, dsc.last_fft_path())
''' % ROOT
    for scratch_mb, env, want in ((256, {}, 'r2c_fused_l2 c2r_fused_l2'), (12, {}, 'r2c_2pass_regs c2r_2pass_regs'),
                                  (256, {'DSC_NO_FUSED_L2': '1'}, 'r2c_2pass_regs c2r_2pass_regs')):
        r = subprocess.run([sys.executable, '-c', code, str(scratch_mb)], capture_output=True, text=True, timeout=600, env={**os.environ, **env})
        assert r.returncode == 0 and ('OK ' + want) in r.stdout, (scratch_mb, env, r.stdout[-300:], r.stderr[-800:])
