"""tools/check_transpose.py — dsc_transpose for every permutation of 2 .. 4 axes, odd extents, every dtype, against numpy (exact)."""
import sys, itertools, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(4 << 30, 1 << 28)
rng = np.random.default_rng(3)
ok = True
for dt in (np.float32, np.float64, np.complex64, np.complex128):
    for shape in ((5, 7), (33, 65), (3, 37, 41), (64, 31, 70), (2, 3, 35, 37), (5, 33, 4, 66)):
        x = rng.standard_normal(shape).astype(dt)
        if np.dtype(dt).kind == 'c':
            x = (x + 1j * rng.standard_normal(shape)).astype(dt)
        t = dsc.from_numpy(x)
        got = dsc.transpose(t).numpy()
        if not np.array_equal(got, x.transpose()): ok = False; print('BAD default', dt.__name__, shape)
        for perm in itertools.permutations(range(len(shape))):
            got = dsc.transpose(t, perm).numpy()
            if got.shape != x.transpose(perm).shape or not np.array_equal(got, x.transpose(perm)):
                ok = False; print('BAD', dt.__name__, shape, perm)
dsc.synchronize()
print('TRANSPOSE', 'OK' if ok else 'FAILED')
