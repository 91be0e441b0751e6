"""tools/stress_fused.py — repeated runs of the paired-team transforms (f64, 131072-point rows) against numpy; prints the rows that differ."""
import sys, numpy as np
sys.path.insert(0, '.')
import dsc_amd as dsc
dsc.init(6 << 30, 1 << 30)
rng = np.random.default_rng(11)
bad = 0
for rows in (3, 17, 40, 100):
    z = rng.standard_normal((rows, 131072)) + 1j * rng.standard_normal((rows, 131072))
    wf, wi = np.fft.fft(z, axis=-1), np.fft.ifft(z, axis=-1)
    x = rng.standard_normal((rows, 262144))
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
                print(name, 'rows', rows, 'rep', rep, 'bad rows', rows_bad.tolist(), 'first bad row: n wrong', len(idx), 'first idx', idx[:8].tolist(), 'last', idx[-3:].tolist(), dsc.last_fft_path(), flush=True)
print('STRESS', 'FAILED' if bad else 'OK', bad)
