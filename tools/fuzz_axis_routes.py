"""Random shapes through the four-step routes along a non-last axis (cols_4step, cols_4step_real) and the persistent f64 lines: every element
against numpy in f64.  Seeded; prints the case on failure.   usage: python tools/fuzz_axis_routes.py [cases] [seed]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
dsc.init(12 << 30, 2 << 30)
rng = np.random.default_rng(seed)
bad = 0
paths = {}
for case in range(n_cases):
    lg = int(rng.integers(12, 18))
    n = 1 << lg
    budget = 1 << 22
    inner_max = max(8, budget // n)
    nd = int(rng.integers(2, 5))
    trailing = []
    left = int(rng.integers(1, inner_max + 1))
    for _ in range(nd - 2):
        d = int(rng.integers(1, 5))
        trailing.append(d)
        left = max(1, left // d)
    trailing.append(left)
    lead = [int(rng.integers(1, 4))] if rng.integers(0, 2) and nd < 4 and n * int(np.prod(trailing)) * 3 <= budget * 2 else []       # at most 4 dimensions
    shape = tuple(lead + [n] + trailing)
    axis = len(lead)
    f64 = bool(rng.integers(0, 2))
    op = ('fft', 'ifft', 'rfft', 'irfft', 'fft_real')[int(rng.integers(0, 5))]
    rdt, cdt, tol = (np.float64, np.complex128, 1e-13) if f64 else (np.float32, np.complex64, 3e-6)
    if op in ('fft', 'ifft'):
        x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(cdt)
        got = getattr(dsc, op)(dsc.from_numpy(x), axis=axis).numpy()
        want = getattr(np.fft, op)(x.astype(np.complex128), axis=axis)
    elif op == 'fft_real':
        x = rng.standard_normal(shape).astype(rdt)
        got = dsc.fft(dsc.from_numpy(x), axis=axis).numpy()
        want = np.fft.fft(x.astype(np.float64), axis=axis)
    elif op == 'rfft':
        x = rng.standard_normal(shape).astype(rdt)
        got = dsc.rfft(dsc.from_numpy(x), axis=axis).numpy()
        want = np.fft.rfft(x.astype(np.float64), axis=axis)
    else:
        bshape = list(shape)
        bshape[axis] = n // 2 + 1
        x = (rng.standard_normal(bshape) + 1j * rng.standard_normal(bshape)).astype(cdt)
        got = dsc.irfft(dsc.from_numpy(x), axis=axis).numpy()
        xz = x.astype(np.complex128)
        idx = [slice(None)] * xz.ndim
        for e in (0, -1):
            idx[axis] = e
            xz[tuple(idx)] = xz[tuple(idx)].real            # dsc_fft.h:227-228: imaginary parts of bins 0 and n/2 are ignored
        want = np.fft.irfft(xz, axis=axis)
    path = dsc.last_fft_path()
    paths[path] = paths.get(path, 0) + 1
    l2 = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    mx = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
    ok = got.shape == want.shape and l2 <= tol and mx <= 10 * tol
    if not ok:
        bad += 1
        print(f'FAIL case {case} seed {seed}: {op} {"f64" if f64 else "f32"} {shape} axis {axis} [{path}] l2 {l2:.2e} max {mx:.2e}', flush=True)
print('paths', paths)
print('AXIS FUZZ', 'FAILED' if bad else 'OK', bad, 'of', n_cases)
sys.exit(1 if bad else 0)
