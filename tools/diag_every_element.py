"""tools/diag_every_element.py [reps] [n=<N>] [--f32] — where are the wrong elements?  rfft / irfft / fft at N = 4096 by default (2048-point lines), every element against numpy,
repeated; prints the positions of mismatches (row, bin, bin mod 64, lane quad) so that a race shows its pattern."""
import sys, collections
sys.path.insert(0, '.')
import numpy as np
import dsc_amd as dsc
dsc.init(8 << 30, 1 << 30)
rng = np.random.default_rng(64)
F32 = '--f32' in sys.argv
n = int([a for a in sys.argv[1:] if a.startswith('n=')][0][2:]) if any(a.startswith('n=') for a in sys.argv[1:]) else 4096
rows = (1 << 23) // n
TOL = 2e-6 if F32 else 1e-12
xd = rng.standard_normal((rows, n)).astype(np.float32 if F32 else np.float64)
wd = np.fft.rfft(xd.astype(np.float64), axis=-1)
td = dsc.from_numpy(xd)
tD = dsc.from_numpy(wd.astype(np.complex64 if F32 else np.complex128))
zd = (rng.standard_normal((rows, n // 2)) + 1j * rng.standard_normal((rows, n // 2))).astype(np.complex64 if F32 else np.complex128)
wz = np.fft.fft(zd.astype(np.complex128), axis=-1)
tz = dsc.from_numpy(zd)
bad_total = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6):
    for name, f, want in (('rfft', lambda: dsc.rfft(td), wd), ('irfft', lambda: dsc.irfft(tD), xd), ('fft', lambda: dsc.fft(tz), wz)):
        got = f().numpy()
        err = np.abs(got - want) / np.max(np.abs(want), axis=1, keepdims=True)
        bad = np.argwhere(~(err <= TOL))
        if len(bad):
            bad_total += len(bad)
            rws = collections.Counter(bad[:, 0].tolist())
            print(f'rep {rep} {name}: {len(bad)} wrong elements in {len(rws)} rows; rows {sorted(rws)[:12]} ...; bins {sorted(set(bad[:, 1].tolist()))[:40]}')
            print('   bin mod 64:', sorted(collections.Counter((bad[:, 1] % 64).tolist()).items())[:64])
            print('   row mod 2 :', sorted(collections.Counter((bad[:, 0] % 2).tolist()).items()), ' max err', float(err.max()))
            r, b = bad[0]
            print('   first: row', r, 'bin', b, 'got', got[r, b], 'want', want[r, b], '; does got equal another bin of the row?', [int(k) for k in np.argwhere(np.isclose(want[r], got[r, b], rtol=1e-9, atol=0)).ravel()[:4]])
dsc.synchronize()
print('DIAG', 'CLEAN' if bad_total == 0 else f'{bad_total} wrong elements')
