"""dsc_fft / dsc_ifft with out == x (the reference gathers a line into scratch before it scatters: in place is safe there, dsc.cpp:1990-2040)
on every route a complex transform can take.  usage: python tools/check_inplace.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc

dsc.init(8 << 30, 2 << 30)
rng = np.random.default_rng(12)
bad = 0
cases = [((300, 64), -1), ((70, 1024), -1), ((40, 16384), -1), ((6, 32768), -1), ((5, 65536), -1), ((3, 262144), -1), ((2, 1048576), -1),
         ((256, 300), 0), ((2048, 40), 0), ((4096, 72), 0), ((65536, 24), 0), ((3, 8192, 20), 1), ((16, 40), 0)]
for dt, tol in ((np.complex64, 2e-6), (np.complex128, 1e-13)):
    for shape, axis in cases:
        z = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
        for name in ('fft', 'ifft'):
            t = dsc.from_numpy(z)
            r = getattr(dsc, name)(t, out=t, axis=axis)
            path = dsc.last_fft_path()
            want = getattr(np.fft, name)(z.astype(np.complex128), axis=axis)
            got = t.numpy()
            l2 = float(np.linalg.norm(got - want) / np.linalg.norm(want))
            ok = l2 <= tol and np.array_equal(r.numpy(), got)
            bad += not ok
            print(f'{name} in place {np.dtype(dt).name} {shape} axis {axis}: [{path}] l2 {l2:.2e} {"ok" if ok else "FAIL"}', flush=True)
print('IN PLACE', 'FAILED' if bad else 'OK', bad)
sys.exit(1 if bad else 0)
