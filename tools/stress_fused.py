"""tools/stress_fused.py — repeated runs of the paired-team transforms (f64, 32768-, 65536- and 131072-point rows) against numpy; prints the rows that differ."""
import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(6 << 30, 1 << 30)
rng = np.random.default_rng(11)
bad = 0
for rows, L in ((3, 131072), (17, 131072), (40, 131072), (100, 131072), (5, 65536), (90, 65536), (200, 65536), (3, 32768), (70, 32768), (300, 32768)):
    z = rng.standard_normal((rows, L)) + 1j * rng.standard_normal((rows, L))
    wf, wi = np.fft.fft(z, axis=-1), np.fft.ifft(z, axis=-1)
    x = rng.standard_normal((rows, 2 * L))
    wr = np.fft.rfft(x, axis=-1)
    tz, tx, tX = dsc.from_numpy(z), dsc.from_numpy(x), dsc.from_numpy(wr)
    for rep in range(6):
        for name, f, want in (('fft', lambda: dsc.fft(tz), wf), ('ifft', lambda: dsc.ifft(tz), wi), ('rfft', lambda: dsc.rfft(tx), wr),
                              ('irfft', lambda: dsc.irfft(tX), x)):
            got = f().numpy()
            err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
            rows_bad = np.nonzero(err > 1e-13)[0]
            if len(rows_bad):
                bad += 1
                r = rows_bad[0]
                d = np.abs(got[r] - want[r])
                idx = np.nonzero(d > 1e-9 * np.abs(want[r]).max())[0]
                print(name, 'L', L, 'rows', rows, 'rep', rep, 'bad rows', rows_bad.tolist(), 'first bad row: n wrong', len(idx), 'first idx', idx[:8].tolist(), 'last', idx[-3:].tolist(), dsc.last_fft_path(), flush=True)
# f32: 65536-point rows (six teams per XCD) and 131072-point rows (512-thread tasks, paired teams)
for rows, L in ((3, 131072), (50, 131072), (120, 131072), (160, 65536)):
    z = (rng.standard_normal((rows, L)) + 1j * rng.standard_normal((rows, L))).astype(np.complex64)
    x = rng.standard_normal((rows, 2 * L)).astype(np.float32)
    wf, wi = np.fft.fft(z.astype(np.complex128), axis=-1), np.fft.ifft(z.astype(np.complex128), axis=-1)
    wr = np.fft.rfft(x.astype(np.float64), axis=-1)
    tz, tx, tX = dsc.from_numpy(z), dsc.from_numpy(x), dsc.from_numpy(wr.astype(np.complex64))
    wx = np.fft.irfft(wr.astype(np.complex64).astype(np.complex128), axis=-1)
    for rep in range(4):
        for name, f, want in (('fft', lambda: dsc.fft(tz), wf), ('ifft', lambda: dsc.ifft(tz), wi), ('rfft', lambda: dsc.rfft(tx), wr),
                              ('irfft', lambda: dsc.irfft(tX), wx)):
            got = f().numpy()
            err = np.max(np.abs(got - want), axis=1) / np.max(np.abs(want), axis=1)
            rows_bad = np.nonzero(err > 3e-5)[0]
            if len(rows_bad):
                bad += 1
                r = rows_bad[0]
                d = np.abs(got[r] - want[r])
                idx = np.nonzero(d > 1e-3 * np.abs(want[r]).max())[0]
                print('f32', name, 'L', L, 'rows', rows, 'rep', rep, 'bad rows', rows_bad[:10].tolist(), 'worst', float(err.max()), 'n wrong', len(idx),
                      'idx', idx[:12].tolist(), '..', idx[-4:].tolist(), dsc.last_fft_path(), flush=True)
print('STRESS', 'FAILED' if bad else 'OK', bad)
