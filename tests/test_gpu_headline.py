"""GPU tests of the register-resident 65536-point kernels (the hot path of BASELINE.json):
parity against the CPU oracle on seeded rows, and — at BASELINE's full sizes, where the oracle
would take minutes — size-independent properties: round trip, linearity, Parseval, impulse /
tone known answers, and agreement of a random sample of rows with the oracle.  All through the
C ABI.  Tolerance: rel-L2 <= 1e-5 (north_star)."""
import numpy as np
import pytest

from tests.helpers import assert_close, rel_l2

pytestmark = pytest.mark.gpu
N = 65536


@pytest.fixture(scope='module')
def dsc():
    import dsc_amd
    try:
        dsc_amd.init(14 << 30, 5 << 30)
    except RuntimeWarning:
        pass
    yield dsc_amd
    dsc_amd.synchronize()


def test_paths_are_the_hand_written_kernels(dsc):
    x = dsc.from_numpy(np.ones((2, N), np.float32))
    X = dsc.rfft(x)
    assert dsc.last_fft_path() == 'r2c_64k_regs'
    dsc.irfft(X)
    assert dsc.last_fft_path() == 'c2r_64k_regs'
    H = dsc.from_numpy(np.ones(N // 2 + 1, np.complex64))
    dsc.filter_fft(x, H)
    assert dsc.last_fft_path() == 'filter_64k_regs'
    # zero-padded / cropped rows stay on the register kernel (the row's buffer descriptor ends at the last valid sample)
    dsc.rfft(dsc.from_numpy(np.ones((2, 60000), np.float32)))
    assert dsc.last_fft_path() == 'r2c_64k_regs'
    # anything the fast kernels do not cover must still be right through the generic path
    dsc.rfft(dsc.from_numpy(np.ones((65536, 2), np.float32)), axis=0)       # strided lines: transposed to the back first
    assert dsc.last_fft_path() == 'r2c_64k_regs'
    dsc.rfft(dsc.from_numpy(np.ones((2, 1 << 21), np.float32)))             # the longest row of the two-kernel route
    assert dsc.last_fft_path() == 'r2c_2pass_regs'
    dsc.rfft(dsc.from_numpy(np.ones((1, 1 << 22), np.float32)))             # beyond it
    assert dsc.last_fft_path() == 'generic_4step'
    dsc.rfft(dsc.from_numpy(np.ones((256, 8), np.float32)), axis=0)         # strided lines of 32 .. 2048 complex points: the column kernel
    assert dsc.last_fft_path() == 'regs_cols'
    dsc.rfft(dsc.from_numpy(np.ones((32, 8), np.float32)), axis=0)          # shorter strided lines: one thread per line
    assert dsc.last_fft_path() == 'regs_tiny_cols'
    dsc.rfft(dsc.from_numpy(np.ones((8, 32), np.float32)))                  # ... and along the last axis
    assert dsc.last_fft_path() == 'regs_tiny'
    dsc.rfft(dsc.from_numpy(np.ones((8, 2), np.float32)))                   # one complex point: the LDS line kernel
    assert dsc.last_fft_path() == 'generic_lds'


@pytest.mark.parametrize('rows', [1, 2, 17, 255, 257, 600])
def test_rfft_irfft_vs_oracle_all_row_skews(dsc, rows):
    """Output rows are 32769 bins long: every row has a different alignment (row mod 16) in the
    aligned-store staging, and grids smaller / larger than the 256 CUs take different loops."""
    from oracle import port
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, N)).astype(np.float32)
    X = dsc.rfft(dsc.from_numpy(x))
    got = X.numpy()
    pick = sorted(set([0, rows - 1] + list(rng.integers(0, rows, 6))))
    for r in pick:
        assert_close(got[r], port.rfft(x[r]), what=f'rfft row {r}/{rows}')
    truth = np.fft.rfft(x.astype(np.float64), axis=-1)
    assert rel_l2(got, truth) <= 1e-6                      # every row, against float64 numpy
    assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)     # dsc_fft.h:221-225
    back = dsc.irfft(X).numpy()
    for r in pick:
        assert_close(back[r], port.irfft(got[r]), what=f'irfft row {r}/{rows}')
    assert rel_l2(back, x) <= 1e-6


def test_irfft_ignores_imag_of_dc_and_nyquist(dsc):
    """dsc_fft.h:227-228 reads only the real parts of bins 0 and n."""
    from oracle import port
    rng = np.random.default_rng(3)
    X = (rng.standard_normal((3, N // 2 + 1)) + 1j * rng.standard_normal((3, N // 2 + 1))).astype(np.complex64)
    assert_close(dsc.irfft(dsc.from_numpy(X)).numpy(), port.irfft(X))


def test_known_answers(dsc):
    t = np.arange(N)
    x = np.zeros((5, N), np.float32)
    x[0, 0] = 1                                            # impulse -> all ones
    x[1, 7] = 1                                            # shifted impulse -> exp(-2 pi i 7 k / N)
    x[2] = 1                                               # constant -> N at DC
    x[3] = np.cos(2 * np.pi * 1234 * t / N)                # tone -> N/2 at bin 1234
    x[4] = np.cos(np.pi * t)                               # Nyquist -> N at bin N/2
    X = dsc.rfft(dsc.from_numpy(x)).numpy().astype(np.complex128)
    k = np.arange(N // 2 + 1)
    assert np.max(np.abs(X[0] - 1)) < 1e-5
    assert np.max(np.abs(X[1] - np.exp(-2j * np.pi * 7 * k / N))) < 1e-5
    want = np.zeros(N // 2 + 1); want[0] = N
    assert np.max(np.abs(X[2] - want)) < 1e-5 * N
    want = np.zeros(N // 2 + 1); want[1234] = N / 2
    assert np.max(np.abs(X[3] - want)) < 1e-5 * N
    want = np.zeros(N // 2 + 1); want[N // 2] = N
    assert np.max(np.abs(X[4] - want)) < 1e-5 * N


def test_fused_filter_vs_composition(dsc, golden):
    from oracle import port
    rng = np.random.default_rng(8)
    taps = 537
    tt = np.arange(taps) - (taps - 1) / 2
    b = np.zeros(N, np.float32)
    b[:taps] = (np.sinc(0.2 * tt) * np.hamming(taps) * 0.2).astype(np.float32)
    Hh = port.rfft(b)
    Hh[0] += 0.25j                                         # the product's imaginary DC part must be ignored
    for rows in (1, 3, 300):
        s = rng.standard_normal((rows, N)).astype(np.float32)
        got = dsc.filter_fft(dsc.from_numpy(s), dsc.from_numpy(Hh)).numpy()
        assert dsc.last_fft_path() == 'filter_64k_regs'
        want = port.irfft(port.mul(port.rfft(s[:3]), Hh))
        assert_close(got[:3], want, what=f'fused filter rows={rows}')
        comp = dsc.irfft(dsc.rfft(dsc.from_numpy(s)) * dsc.from_numpy(Hh)).numpy()
        assert rel_l2(got, comp) <= 2e-6
    # the README pipeline's golden case (65000-sample signal, 537 taps), zero-padded by hand
    for rec, xs, y in golden.cases('filter'):
        if rec['n'] != N:
            continue
        s = np.zeros(N, np.float32); s[:len(xs[0])] = xs[0]
        bb = np.zeros(N, np.float32); bb[:len(xs[1])] = xs[1]
        got = dsc.filter_fft(dsc.from_numpy(s), dsc.rfft(dsc.from_numpy(bb))).numpy()
        assert_close(got, y, what='README filterFFT golden through the fused kernel')


def test_full_size_properties(dsc):
    """BASELINE configs[1]: [8192, 65536] f32 (2 GiB in, 2 GiB out)."""
    from oracle import port
    B = 8192
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((B, N), dtype=np.float32)
    xd = dsc.from_numpy(x)
    X = dsc.rfft(xd)
    assert dsc.last_fft_path() == 'r2c_64k_regs'
    Xh = X.numpy()
    # (1) a sample of rows against the oracle
    for r in [0, 1, 255, 256, 4095, 8191] + list(rng.integers(0, B, 10)):
        assert_close(Xh[r], port.rfft(x[r]), what=f'full-size row {r}')
    # (2) Parseval on every row: sum |x|^2 = (|X0|^2 + |XM|^2 + 2 sum |Xk|^2) / N
    e_t = np.sum(x.astype(np.float64) ** 2, axis=1)
    p = np.abs(Xh.astype(np.complex128)) ** 2
    e_f = (p[:, 0] + p[:, -1] + 2 * np.sum(p[:, 1:-1], axis=1)) / N
    assert np.max(np.abs(e_f - e_t) / e_t) < 1e-5
    # (3) round trip on every row
    back = dsc.irfft(X)
    assert dsc.last_fft_path() == 'c2r_64k_regs'
    bh = back.numpy()
    assert rel_l2(bh, x) <= 1e-6 and np.max(np.abs(bh - x)) < 1e-4
    del back, bh
    # (4) linearity: rfft(2 x + roll(x)) = 2 X + rfft(roll(x)) on a slice of the batch
    a = x[:512]
    r = np.roll(a, 1, axis=0)
    lhs = dsc.rfft(dsc.from_numpy(2 * a + r)).numpy()
    rhs = 2 * Xh[:512] + dsc.rfft(dsc.from_numpy(r)).numpy()
    assert rel_l2(lhs, rhs) <= 2e-6


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 flow with real device work: two ranks launched as the driver launches them,
    sharing the box's single GPU (gloo for the barrier / max-reduce; RCCL needs distinct devices)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
           '--batch', '1024', '--backend', 'gloo', '--share-device']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch'] == 2048 and j['config']['kernel_path'] == 'r2c_64k_regs'
    assert j['parity']['rel_l2_vs_cpu_oracle'] <= 1e-5
    assert j['roofline']['achieved'] > 0 and 'cpu_baseline' not in j
    # reassembly of the two shards of real dsc_rfft output: through host staging for the gloo collectives, and directly
    # on the device (HIP IPC between the two processes) for the hand-rolled pushes; each proven against the owners' shards
    g = j['allgather']
    assert set(g['variants']) == {'allgather', 'p2p', 'ipc'}, g
    for name, v in g['variants'].items():
        assert v.get('verified') is True, (name, v)
    assert g['variants']['ipc']['memory'] == 'device' and g['verified'] is True


def test_f64_262144_register_path(dsc):
    """BASELINE config 5 (f64 N=262144) on the team kernel of fft_xcd_fused.hip (one cooperative launch: 512-point row tasks, 256-point
    column tasks with the packed-real pass fused, the four-step intermediate in the XCD-local L2).  Parity 1e-12 vs the oracle, round
    trip, path check, row counts that leave teams without a row."""
    from oracle import port
    rng = np.random.default_rng(55)
    for rows in (1, 3, 37):
        x = rng.standard_normal((rows, 262144))
        X = dsc.rfft(dsc.from_numpy(x))
        assert dsc.last_fft_path() == 'r2c_fused_l2'       # round 2: one launch, intermediate in the XCD-local L2
        got = X.numpy()
        for r in sorted({0, rows - 1}):
            assert_close(got[r], port.rfft(x[r]), what=f'f64 rfft row {r}/{rows}')
        assert rel_l2(got, np.fft.rfft(x, axis=-1)) <= 1e-14
        assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)
        Xq = got.copy()
        Xq[:, 0] += 2j                                     # imaginary parts of bins 0 and n are ignored (dsc_fft.h:227-228)
        back = dsc.irfft(dsc.from_numpy(Xq))
        assert dsc.last_fft_path() == 'c2r_fused_l2'
        bh = back.numpy()
        assert_close(bh[0], port.irfft(Xq[0]), what='f64 irfft')
        assert rel_l2(bh, x) <= 1e-14


@pytest.mark.parametrize('n', [1024, 2048, 4096, 8192, 16384, 32768])
def test_mid_size_register_path(dsc, n):
    """Real lengths 512 .. 32768 (complex 256 .. 16384) of contiguous full rows run in
    fft_regs_mid.hip; row counts that are not a multiple of the lines-per-workgroup exercise
    the partially filled last group.  rfft / irfft / fft / ifft against the oracle."""
    from oracle import port
    rng = np.random.default_rng(n)
    for rows in (1, 5, 67, 131):
        x = rng.standard_normal((rows, n)).astype(np.float32)
        X = dsc.rfft(dsc.from_numpy(x))
        assert dsc.last_fft_path() == 'regs_mid'
        got = X.numpy()
        for r in sorted({0, rows // 2, rows - 1}):
            assert_close(got[r], port.rfft(x[r]), what=f'rfft n={n} row {r}/{rows}')
        assert rel_l2(got, np.fft.rfft(x.astype(np.float64), axis=-1)) <= 1e-6
        assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)
        Xq = got.copy()
        Xq[:, 0] += 2j                                     # ignored by dsc_fft.h:227-228
        Xq[:, -1] -= 3j
        back = dsc.irfft(dsc.from_numpy(Xq))
        assert dsc.last_fft_path() == 'regs_mid'
        bh = back.numpy()
        for r in sorted({0, rows - 1}):
            assert_close(bh[r], port.irfft(Xq[r]), what=f'irfft n={n} row {r}/{rows}')
        assert rel_l2(bh, x) <= 1e-6
        # complex transforms of length n/2
        z = (rng.standard_normal((rows, n // 2)) + 1j * rng.standard_normal((rows, n // 2))).astype(np.complex64)
        Z = dsc.fft(dsc.from_numpy(z))
        assert dsc.last_fft_path() == 'regs_mid'
        zh = Z.numpy()
        assert_close(zh[rows - 1], port.fft(z[rows - 1]), what=f'fft n={n // 2}')
        assert rel_l2(zh, np.fft.fft(z.astype(np.complex128), axis=-1)) <= 1e-6
        zb = dsc.ifft(Z)
        assert dsc.last_fft_path() == 'regs_mid'
        assert_close(zb.numpy()[0], port.ifft(zh[0]), what=f'ifft n={n // 2}')
        assert rel_l2(zb.numpy(), z) <= 1e-6
        # complex transform of REAL input (cast first: dsc.cpp:1984-1988), forward and inverse
        xr = x[:, :n // 2].copy()
        F = dsc.fft(dsc.from_numpy(xr))
        assert dsc.last_fft_path() == 'regs_mid'
        assert_close(F.numpy()[rows - 1], port.fft(xr[rows - 1]), what=f'fft(real) n={n // 2}')
        Fi = dsc.ifft(dsc.from_numpy(xr))
        assert dsc.last_fft_path() == 'regs_mid'
        assert_close(Fi.numpy()[0], port.ifft(xr[0]), what=f'ifft(real) n={n // 2}')


@pytest.mark.parametrize('n', [1024, 2048, 4096, 8192, 16384, 32768])
def test_mid_size_register_path_f64(dsc, n):
    """The same kernels in f64 (tolerance 1e-12 against the oracle)."""
    from oracle import port
    rng = np.random.default_rng(n + 1)
    for rows in (1, 37):
        x = rng.standard_normal((rows, n))
        X = dsc.rfft(dsc.from_numpy(x))
        assert dsc.last_fft_path() == 'regs_mid'
        got = X.numpy()
        for r in sorted({0, rows - 1}):
            assert_close(got[r], port.rfft(x[r]), what=f'f64 rfft n={n} row {r}/{rows}')
        assert rel_l2(got, np.fft.rfft(x, axis=-1)) <= 1e-14
        assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)
        Xq = got.copy()
        Xq[:, 0] += 2j
        Xq[:, -1] -= 3j
        back = dsc.irfft(dsc.from_numpy(Xq))
        assert dsc.last_fft_path() == 'regs_mid'
        bh = back.numpy()
        assert_close(bh[rows - 1], port.irfft(Xq[rows - 1]), what=f'f64 irfft n={n}')
        assert rel_l2(bh, x) <= 1e-14
        z = rng.standard_normal((rows, n // 2)) + 1j * rng.standard_normal((rows, n // 2))
        Z = dsc.fft(dsc.from_numpy(z))
        assert dsc.last_fft_path() == 'regs_mid'
        zh = Z.numpy()
        assert_close(zh[rows - 1], port.fft(z[rows - 1]), what=f'f64 fft n={n // 2}')
        zb = dsc.ifft(Z)
        assert dsc.last_fft_path() == 'regs_mid'
        assert rel_l2(zb.numpy(), z) <= 1e-14


def test_full_size_filter_config3(dsc):
    """BASELINE config 3 at full size ([4096, 65536] f32, fused rfft * H -> irfft): known answers that hold for
    every row (identity filter, phase ramp = circular shift), linearity, and sampled rows against the
    three-operator composition and the CPU oracle."""
    from oracle import port
    rows = 4096
    rng = np.random.default_rng(33)
    blk = rng.standard_normal((256, N)).astype(np.float32)
    scale = (1.0 + np.arange(rows) % 7).astype(np.float32)
    s = np.tile(blk, (rows // 256, 1)) * scale[:, None]
    ts = dsc.from_numpy(s)
    k = np.arange(N // 2 + 1)
    # identity
    y = dsc.filter_fft(ts, dsc.from_numpy(np.ones(N // 2 + 1, np.complex64)))
    assert dsc.last_fft_path() == 'filter_64k_regs'
    assert rel_l2(y.numpy(), s) <= 1e-6
    # phase ramp: H[k] = exp(-2 pi i k d / N)  ->  y[n] = s[(n - d) mod N] on every row
    d = 1234
    H = np.exp(-2j * np.pi * k * d / N).astype(np.complex64)
    yh = dsc.filter_fft(ts, dsc.from_numpy(H)).numpy()
    assert rel_l2(yh, np.roll(s, d, axis=1)) <= 2e-6
    # a real filter (low-pass taps): sampled rows against the composition and the oracle
    taps = np.zeros(N, np.float32)
    taps[:537] = np.hamming(537) * np.sinc((np.arange(537) - 268) * 0.2) * 0.2
    Hl = port.rfft(taps)
    tH = dsc.from_numpy(Hl)
    yl = dsc.filter_fft(ts, tH).numpy()
    comp = dsc.irfft(dsc.rfft(ts) * tH).numpy()
    assert rel_l2(yl, comp) <= 2e-6
    for r in (0, 1777, rows - 1):
        want = port.irfft(port.mul(port.rfft(s[r]), Hl))
        assert_close(yl[r], want, what=f'filter row {r}')
    # homogeneity across the batch: row r is scale[r] / scale[r % 256] times row r % 256
    base = yl[np.arange(rows) % 256]
    ratio = (scale / scale[np.arange(rows) % 256])[:, None]
    assert rel_l2(yl, base * ratio) <= 2e-6


def test_full_size_f64_config5(dsc):
    """BASELINE config 5 at full size ([2048, 262144] f64): sampled rows against the oracle (1e-12), Parseval and
    round trip on every row, homogeneity across the batch."""
    from oracle import port
    n, rows = 262144, 2048
    rng = np.random.default_rng(99)
    blk = rng.standard_normal((64, n))
    scale = 1.0 + (np.arange(rows) % 5) * 0.25
    x = np.tile(blk, (rows // 64, 1)) * scale[:, None]
    tx = dsc.from_numpy(x)
    X = dsc.rfft(tx)
    assert dsc.last_fft_path() == 'r2c_fused_l2'
    Xh = X.numpy()
    assert Xh.shape == (rows, n // 2 + 1)
    for r in (0, 1031, rows - 1):
        assert_close(Xh[r], port.rfft(x[r]), what=f'f64 full-size row {r}')
    assert np.all(Xh[:, 0].imag == 0) and np.all(Xh[:, -1].imag == 0)
    e_t = np.sum(x * x, axis=1)
    p = Xh.real ** 2 + Xh.imag ** 2
    e_f = (p[:, 0] + p[:, -1] + 2 * np.sum(p[:, 1:-1], axis=1)) / n
    assert np.max(np.abs(e_f - e_t) / e_t) < 1e-13
    idx = np.arange(rows) % 64
    assert rel_l2(Xh, Xh[idx] * (scale / scale[idx])[:, None]) <= 1e-15
    del p, tx                                          # three 4 GiB tensors do not fit the 12 GiB test arena next to cached plans
    back = dsc.irfft(X)
    assert dsc.last_fft_path() == 'c2r_fused_l2'
    bh = back.numpy()
    assert rel_l2(bh, x) <= 1e-14 and np.max(np.abs(bh - x)) < 1e-12


@pytest.mark.parametrize('dt,n', [(np.float32, 131072), (np.float32, 262144), (np.float32, 524288), (np.float32, 1048576), (np.float32, 2097152),
                                  (np.float64, 65536), (np.float64, 131072), (np.float64, 524288), (np.float64, 1048576), (np.float64, 2097152)])
def test_two_pass_long_transforms(dsc, dt, n):
    """Real lengths beyond one CU's registers (fft_r2c_2pass.hip: rows kernel + column kernel with the real pass fused),
    every L1 = n / 2048 in {32, 64, 128, 256}, f32 and f64 (f64 262144 = config 5 has its own tests)."""
    from oracle import port
    rng = np.random.default_rng(n + (1 if dt == np.float64 else 0))
    exact = 1e-6 if dt == np.float32 else 1e-14
    for rows in (1, 5):
        x = rng.standard_normal((rows, n)).astype(dt)
        X = dsc.rfft(dsc.from_numpy(x))
        fused = n <= 262144 and not (dt == np.float32 and n == 65536)      # fft_xcd_fused.hip: one launch, intermediate in the XCD-local L2
        assert dsc.last_fft_path() == ('r2c_fused_l2' if fused else 'r2c_2pass_regs')
        got = X.numpy()
        assert_close(got[rows - 1], port.rfft(x[rows - 1]), what=f'rfft {np.dtype(dt).name} n={n}')
        assert rel_l2(got, np.fft.rfft(x.astype(np.float64), axis=-1)) <= exact
        assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)
        Xq = got.copy()
        Xq[:, 0] += 2j                                     # ignored by dsc_fft.h:227-228
        Xq[:, -1] -= 3j
        back = dsc.irfft(dsc.from_numpy(Xq))
        assert dsc.last_fft_path() == ('c2r_fused_l2' if fused else 'c2r_2pass_regs')
        bh = back.numpy()
        assert_close(bh[0], port.irfft(Xq[0]), what=f'irfft {np.dtype(dt).name} n={n}')
        assert rel_l2(bh, x) <= exact


@pytest.mark.parametrize('ls', [60001, 65000, 32769, 1, 70000])
def test_padded_and_cropped_rows_on_the_register_kernels(dsc, ls):
    """dsc_rfft(x, n) with rows shorter (zero padding, dsc.cpp:2125-2133; odd lengths split a sample pair) or longer
    (crop) than the transform, and the README pipeline on such rows (README.md:113-135: s is padded to the transform
    length inside rfft), against the oracle."""
    from oracle import port
    rng = np.random.default_rng(ls)
    rows = 5
    x = rng.standard_normal((rows, ls)).astype(np.float32)
    X = dsc.rfft(dsc.from_numpy(x), n=N)
    assert dsc.last_fft_path() == 'r2c_64k_regs'
    got = X.numpy()
    for r in (0, rows - 1):
        assert_close(got[r], port.rfft(x[r], N), what=f'padded rfft ls={ls} row {r}')
    xp = np.zeros((rows, N), np.float32)
    xp[:, :min(ls, N)] = x[:, :N]
    assert rel_l2(got, np.fft.rfft(xp.astype(np.float64), axis=-1)) <= 1e-6
    taps = np.zeros(N, np.float32)
    taps[:97] = rng.standard_normal(97).astype(np.float32)
    H = port.rfft(taps)
    y = dsc.filter_fft(dsc.from_numpy(x), dsc.from_numpy(H))
    assert dsc.last_fft_path() == 'filter_64k_regs' and y.shape == (rows, N)
    want = port.irfft(port.mul(port.rfft(x[1], N), H))
    assert_close(y.numpy()[1], want, what=f'padded filter ls={ls}')
    # irfft with fewer / more bins than n/2 + 1 (`n` counts bins, dsc.cpp:2199-2200)
    lb = min(max(ls // 2, 2), 40000)
    Y = (rng.standard_normal((rows, lb)) + 1j * rng.standard_normal((rows, lb))).astype(np.complex64)
    b = dsc.irfft(dsc.from_numpy(Y), n=N // 2 + 1)
    assert dsc.last_fft_path() == 'c2r_64k_regs'
    assert_close(b.numpy()[rows - 1], port.irfft(Y[rows - 1], N // 2 + 1), what=f'padded irfft lb={lb}')


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('n', [1024, 8192, 32768])
def test_mid_sizes_padded_and_cropped(dsc, dt, n):
    """Zero padding / cropping on the register kernels of fft_regs_mid.hip: every transform (rfft, irfft, fft, ifft, fft of
    real input) with axis lengths below (odd ones cut a sample pair) and above the transform length, rows that do not fill
    the last workgroup, against the oracle."""
    from oracle import port
    cdt = np.complex64 if dt == np.float32 else np.complex128
    rng = np.random.default_rng(n)
    tol = 1e-6 if dt == np.float32 else 1e-14
    for rows, ls in ((3, n - 1), (11, n // 2 + 7), (5, n + 100), (2, 1)):
        x = rng.standard_normal((rows, ls)).astype(dt)
        got = dsc.rfft(dsc.from_numpy(x), n=n)
        assert dsc.last_fft_path() == 'regs_mid', (n, ls)
        assert_close(got.numpy()[rows - 1], port.rfft(x[rows - 1], n), what=f'rfft n={n} ls={ls}')
        xp = np.zeros((rows, n), dt)
        xp[:, :min(ls, n)] = x[:, :n]
        assert rel_l2(got.numpy(), np.fft.rfft(xp.astype(np.float64), axis=-1)) <= tol
        c = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(cdt)
        for name in ('fft', 'ifft'):
            g = getattr(dsc, name)(dsc.from_numpy(c), n=n // 2)
            assert dsc.last_fft_path() == ('regs_mid' if n // 2 >= 256 else 'generic_lds')
            assert_close(g.numpy()[0], getattr(port, name)(c[0], n // 2), what=f'{name} n={n // 2} ls={ls}')
        g = dsc.fft(dsc.from_numpy(x), n=n // 2)              # real input, cast on load
        assert_close(g.numpy()[rows - 1], port.fft(x[rows - 1], n // 2), what=f'fft(real) n={n // 2} ls={ls}')
        # irfft: `n` counts bins (dsc.cpp:2199-2200); fewer bins than n/2+1 are zero filled, more are cropped
        bins = n // 2 + 1
        Y = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(cdt)
        if ls >= 2:
            b = dsc.irfft(dsc.from_numpy(Y), n=bins)
            assert dsc.last_fft_path() == 'regs_mid'
            assert_close(b.numpy()[0], port.irfft(Y[0], bins), what=f'irfft bins={bins} ls={ls}')


@pytest.mark.parametrize('dt', [np.complex64, np.complex128])
def test_long_complex_transforms_along_a_non_last_axis(dsc, dt):
    """dsc_fft / dsc_ifft along a non-last axis (dsc.cpp:1977-1978: any axis, through the strided iterator) at 4096 points and more:
    the four-step route of the column kernel (two passes: lines over j2 with the W_n^{j1 k2} twiddle, then lines over j1 written to
    rows n2 k1 + k2).  Complex and real input, 2-D and 3-D, inner sizes that leave the last tile ragged; every element against
    numpy in f64 and sampled columns against the oracle; a padded call (n != axis length) must leave the route."""
    from oracle import port
    rng = np.random.default_rng(4)
    tol = 1e-6 if dt == np.complex64 else 1e-14
    for shape, axis in (((4096, 72), 0), ((8192, 33), 0), ((3, 16384, 17), 1), ((65536, 24), 0), ((2, 131072, 9), 1)):
        z = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)
        for name in ('fft', 'ifft'):
            got = getattr(dsc, name)(dsc.from_numpy(z), axis=axis).numpy()
            assert dsc.last_fft_path() == 'cols_4step', (shape, name, dsc.last_fft_path())
            want = getattr(np.fft, name)(z.astype(np.complex128), axis=axis)
            assert rel_l2(got, want) <= tol, (shape, name)
            assert float(np.max(np.abs(got - want)) / np.max(np.abs(want))) <= 8 * tol, (shape, name)
            col = (slice(None), shape[1] - 1) if axis == 0 else (shape[0] - 1, slice(None), 3)
            assert_close(got[col], getattr(port, name)(np.ascontiguousarray(z[col])), what=f'{name} {shape} one column against the oracle')
        x = np.ascontiguousarray(z.real)
        got = dsc.fft(dsc.from_numpy(x), axis=axis).numpy()                 # real input, widened while pass 1 loads
        assert dsc.last_fft_path() == 'cols_4step'
        assert rel_l2(got, np.fft.fft(x.astype(np.float64), axis=axis)) <= tol, shape
    z = (rng.standard_normal((5000, 16)) + 1j * rng.standard_normal((5000, 16))).astype(dt)
    got = dsc.fft(dsc.from_numpy(z), n=4096, axis=0).numpy()                 # cropped: not a full line
    assert dsc.last_fft_path() != 'cols_4step'
    assert rel_l2(got, np.fft.fft(z.astype(np.complex128), n=4096, axis=0)) <= tol
    z = z[:4096, :]                                                          # 4096 points with fewer than 64 columns: the one-pass kernel's
    got = dsc.fft(dsc.from_numpy(np.ascontiguousarray(z)), axis=0).numpy()   # 8-column tiles hold whole rows and win (f64: the transposes)
    assert dsc.last_fft_path() != 'cols_4step'
    assert rel_l2(got, np.fft.fft(z.astype(np.complex128), axis=0)) <= tol
    z = (rng.standard_normal((65536, 8)) + 1j * rng.standard_normal((65536, 8))).astype(dt)    # few columns, long lines: n1 grows (1024 x 64)
    got = dsc.fft(dsc.from_numpy(z), axis=0).numpy()
    assert dsc.last_fft_path() == 'cols_4step'
    assert rel_l2(got, np.fft.fft(z.astype(np.complex128), axis=0)) <= tol


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_long_real_transforms_along_a_non_last_axis(dsc, dt):
    """dsc_rfft / dsc_irfft along a non-last axis at 8192 points and more, an even number of columns: two neighbouring real columns travel
    as one complex column through the four-step of the column kernel; the two spectra are separated in its second pass (slice pairs
    (k2, n2 - k2)) and merged in the first pass of the inverse (block pairs (j1, n1 - j1)).  Every element against numpy in f64, one
    column per case against the oracle; exact zeros in the imaginary parts of bins 0 and n/2; irfft ignores those of its input
    (dsc_fft.h:227-228); an odd number of columns and padded calls leave the route."""
    from oracle import port
    rng = np.random.default_rng(8)
    tol = 1e-6 if dt == np.float32 else 1e-14
    cdt = np.complex64 if dt == np.float32 else np.complex128
    for shape, axis in (((8192, 40), 0), ((16384, 34), 0), ((3, 32768, 18), 1), ((65536, 24), 0), ((2, 131072, 16), 1)):
        x = rng.standard_normal(shape).astype(dt)
        got = dsc.rfft(dsc.from_numpy(x), axis=axis).numpy()
        assert dsc.last_fft_path() == 'cols_4step_real', (shape, dsc.last_fft_path())
        want = np.fft.rfft(x.astype(np.float64), axis=axis)
        assert rel_l2(got, want) <= tol, shape
        assert float(np.max(np.abs(got - want)) / np.max(np.abs(want))) <= 8 * tol, shape
        assert not np.take(got, 0, axis=axis).imag.any() and not np.take(got, -1, axis=axis).imag.any()
        col = (slice(None), shape[1] - 1) if axis == 0 else (shape[0] - 1, slice(None), 3)
        assert_close(got[col], port.rfft(np.ascontiguousarray(x[col])), what=f'rfft {shape} one column against the oracle')
        Y = want.astype(cdt)
        Yq = Y.copy()
        edge0 = [slice(None)] * Y.ndim
        edge0[axis] = 0
        edge1 = list(edge0)
        edge1[axis] = -1
        Yq[tuple(edge0)] += 2j
        Yq[tuple(edge1)] -= 3j
        back = dsc.irfft(dsc.from_numpy(Yq), axis=axis).numpy()
        assert dsc.last_fft_path() == 'cols_4step_real', (shape, dsc.last_fft_path())
        assert rel_l2(back, np.fft.irfft(Y.astype(np.complex128), axis=axis)) <= tol, shape
        assert_close(back[col], port.irfft(np.ascontiguousarray(Yq[col])), what=f'irfft {shape} one column against the oracle')
    x = rng.standard_normal((8192, 33)).astype(dt)                           # odd number of columns: no column pairs
    got = dsc.rfft(dsc.from_numpy(x), axis=0).numpy()
    assert dsc.last_fft_path() != 'cols_4step_real'
    assert rel_l2(got, np.fft.rfft(x.astype(np.float64), axis=0)) <= tol
    x = rng.standard_normal((8000, 32)).astype(dt)                           # zero padded to 8192
    got = dsc.rfft(dsc.from_numpy(x), n=8192, axis=0).numpy()
    assert dsc.last_fft_path() != 'cols_4step_real'
    assert rel_l2(got, np.fft.rfft(x.astype(np.float64), n=8192, axis=0)) <= tol


def test_f64_lines_of_16384_points_more_lines_than_workgroups(dsc):
    """f64 lines of 16384 complex points run in a PERSISTENT form of fft_mid_kernel: one group per CU walks lines blockIdx.x,
    + gridDim.x, ... and requests its next line while it stores the current one.  600 lines on 256 CUs: every group walks two or
    three lines (a ragged last round); every element of every line against numpy (f64, 1e-12 of the row's largest value), full
    lines and padded / cropped ones (the descriptor of a single line ends with its valid bytes), one row of each against the oracle."""
    from oracle import port
    rng = np.random.default_rng(16384)
    rows, n = 600, 32768

    def worst(got, want):
        return float(np.max(np.abs(got - want) / np.max(np.abs(want), axis=1, keepdims=True)))

    x = rng.standard_normal((rows, n))
    X = dsc.rfft(dsc.from_numpy(x))
    assert dsc.last_fft_path() == 'regs_mid'
    wX = np.fft.rfft(x, axis=-1)
    assert worst(X.numpy(), wX) <= 1e-12
    assert_close(X.numpy()[rows - 1], port.rfft(x[rows - 1]), what='f64 rfft, last line')
    back = dsc.irfft(X)
    assert dsc.last_fft_path() == 'regs_mid'
    assert worst(back.numpy(), x) <= 1e-12
    z = rng.standard_normal((rows, n // 2)) + 1j * rng.standard_normal((rows, n // 2))
    Z = dsc.fft(dsc.from_numpy(z))
    assert dsc.last_fft_path() == 'regs_mid'
    assert worst(Z.numpy(), np.fft.fft(z, axis=-1)) <= 1e-12
    zb = dsc.ifft(Z)
    assert worst(zb.numpy(), z) <= 1e-12
    F = dsc.fft(dsc.from_numpy(x[:, :n // 2].copy()))        # real input, cast on load
    assert worst(F.numpy(), np.fft.fft(x[:, :n // 2], axis=-1)) <= 1e-12
    del X, back, Z, zb, F
    for ls in (n - 3001, n + 64):                            # shorter (odd: cuts a sample pair) and longer axis than the transform
        xs = rng.standard_normal((rows, ls))
        got = dsc.rfft(dsc.from_numpy(xs), n=n)
        assert dsc.last_fft_path() == 'regs_mid'
        assert worst(got.numpy(), np.fft.rfft(xs, n=n, axis=-1)) <= 1e-12
        assert_close(got.numpy()[257], port.rfft(xs[257], n), what=f'f64 rfft n={n} ls={ls}')
        cs = xs[:, :ls // 2] + 1j * xs[:, ls // 2:2 * (ls // 2)]
        gc = dsc.fft(dsc.from_numpy(cs), n=n // 2)
        assert worst(gc.numpy(), np.fft.fft(cs, n=n // 2, axis=-1)) <= 1e-12
        bins = n // 2 + 1
        Y = cs[:, :min(cs.shape[1], bins + 5)]
        b = dsc.irfft(dsc.from_numpy(np.ascontiguousarray(Y)), n=bins)
        assert dsc.last_fft_path() == 'regs_mid'
        assert_close(b.numpy()[511], port.irfft(Y[511], bins), what=f'f64 irfft bins={bins} ls={Y.shape[1]}')
        assert_close(b.numpy()[3], port.irfft(Y[3], bins), what=f'f64 irfft bins={bins} ls={Y.shape[1]}')


def test_fused_l2_team_kernel_many_rows(dsc):
    """fft_xcd_fused.hip (65536-point complex rows / real length 131072, f32 and f64): more rows than teams, so that every team walks
    several rows (claimed from the global counter, published at a team barrier) and the scratch rows are reused; all four
    operators, every row compared.  A barrier that does not complete aborts at the synchronise."""
    from oracle import port
    rng = np.random.default_rng(131)
    for rows in (47, 160):
        x = rng.standard_normal((rows, 131072)).astype(np.float32)
        X = dsc.rfft(dsc.from_numpy(x))
        assert dsc.last_fft_path() == 'r2c_fused_l2'
        want = port.rfft(x)
        got = X.numpy()
        for r in range(rows):
            assert rel_l2(got[r], want[r]) <= 1e-6, ('rfft row', r, rows)
        assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)
        back = dsc.irfft(X).numpy()
        assert dsc.last_fft_path() == 'c2r_fused_l2'
        wb = port.irfft(want)
        for r in range(rows):
            assert rel_l2(back[r], wb[r]) <= 1e-6, ('irfft row', r, rows)
        z = (rng.standard_normal((rows, 65536)) + 1j * rng.standard_normal((rows, 65536))).astype(np.complex64)
        Z = dsc.fft(dsc.from_numpy(z))
        assert dsc.last_fft_path() == 'c2c_fused_l2'
        wz = port.fft(z)
        zh = Z.numpy()
        for r in range(rows):
            assert rel_l2(zh[r], wz[r]) <= 1e-6, ('fft row', r, rows)
        assert rel_l2(dsc.ifft(Z).numpy(), z) <= 1e-6
    # f64: two teams per XCD, 1 MiB of intermediate per row
    x = rng.standard_normal((90, 131072))
    X = dsc.rfft(dsc.from_numpy(x))
    assert dsc.last_fft_path() == 'r2c_fused_l2'
    got, want = X.numpy(), port.rfft(x)
    for r in range(90):
        assert rel_l2(got[r], want[r]) <= 1e-14, ('f64 rfft row', r)
    back = dsc.irfft(X).numpy()
    assert dsc.last_fft_path() == 'c2r_fused_l2' and rel_l2(back, x) <= 1e-14
    z = rng.standard_normal((70, 65536)) + 1j * rng.standard_normal((70, 65536))
    Z = dsc.fft(dsc.from_numpy(z))
    assert dsc.last_fft_path() == 'c2c_fused_l2'
    assert rel_l2(Z.numpy(), port.fft(z)) <= 1e-14 and rel_l2(dsc.ifft(Z).numpy(), z) <= 1e-14
    dsc.synchronize()


def test_every_element_of_the_f64_paths_with_128_bit_stores(dsc):
    """Every element of every row, repeatedly (an L2-norm check does not see a handful of wrong elements in one row): the f64
    kernels that store c64 values 16 bytes per lane at two waves per SIMD — the mid-size register kernel, the two-pass kernels, the
    column kernel.  tools/stress_64k.py is the long form (it also covers the 65536-point f32 kernels)."""
    rng = np.random.default_rng(64)

    def worst(got, want):
        err = np.abs(got - want) / np.max(np.abs(want), axis=1, keepdims=True)
        w = float(np.max(err))
        if not w <= 1e-12:                                   # say WHERE: a race or a hazard shows in the pattern
            bad = np.argwhere(~(err <= 1e-12))
            print(f'{len(bad)} wrong elements; rows {sorted(set(bad[:, 0].tolist()))[:16]}; columns {sorted(set(bad[:, 1].tolist()))[:48]}; '
                  f'first got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}')
        return w

    for n, rows in ((4096, 2048), (1024, 4096), (8192, 1024), (16384, 512), (524288, 12)):      # every f64 configuration of fft_mid_kernel that stores 128 bits
        xd = rng.standard_normal((rows, n))
        wd = np.fft.rfft(xd, axis=-1)
        zd = rng.standard_normal((rows, n // 2)) + 1j * rng.standard_normal((rows, n // 2))
        wz = np.fft.fft(zd, axis=-1)
        td, tD, tz = dsc.from_numpy(xd), dsc.from_numpy(wd), dsc.from_numpy(zd)
        for rep in range(12 if n == 4096 else 3):            # the store-data hazard showed in one row of a few thousand
            assert worst(dsc.rfft(td).numpy(), wd) <= 1e-12, (n, rep, 'rfft', dsc.last_fft_path())
            assert worst(dsc.irfft(tD).numpy(), xd) <= 1e-12, (n, rep, 'irfft', dsc.last_fft_path())
            assert worst(dsc.fft(tz).numpy(), wz) <= 1e-12, (n, rep, 'fft', dsc.last_fft_path())
        del td, tD, tz
    xd = rng.standard_normal((1024, 2048))
    zd = xd + 1j * rng.standard_normal((1024, 2048))
    td, tz = dsc.from_numpy(xd), dsc.from_numpy(zd)
    wd, wz = np.fft.rfft(xd, axis=0), np.fft.fft(zd, axis=0)
    for rep in range(3):
        assert worst(dsc.rfft(td, axis=0).numpy().T, wd.T) <= 1e-12 and dsc.last_fft_path() == 'regs_cols'
        assert worst(dsc.fft(tz, axis=0).numpy().T, wz.T) <= 1e-12


def test_fused_l2_paired_teams_every_element(dsc):
    """The paired teams of the team kernel (f64: config 5's 131072-point rows — two teams per XCD take turns on ONE scratch row —
    and 65536-point rows, two pairs per XCD): repeated
    launches, every element of every row against numpy — a wrong store shows up as 16 elements of one row (tools/stress_fused.py
    found two such bugs: a team overwriting its own intermediate after its partner had left, and 128-bit store data rewritten by
    the next multiply).  3 rows: teams with and without work; 40: several rows per team."""
    rng = np.random.default_rng(262)
    for rows, L in ((3, 131072), (40, 131072), (70, 65536), (150, 32768)):
        z = rng.standard_normal((rows, L)) + 1j * rng.standard_normal((rows, L))
        x = rng.standard_normal((rows, 2 * L))
        wf, wi, wr = np.fft.fft(z, axis=-1), np.fft.ifft(z, axis=-1), np.fft.rfft(x, axis=-1)
        tz, tx = dsc.from_numpy(z), dsc.from_numpy(x)
        for rep in range(4):
            tX = dsc.from_numpy(wr)
            for name, f, want, path in (('fft', lambda: dsc.fft(tz), wf, 'c2c_fused_l2'), ('ifft', lambda: dsc.ifft(tz), wi, 'c2c_fused_l2'),
                                        ('rfft', lambda: dsc.rfft(tx), wr, 'r2c_fused_l2'), ('irfft', lambda: dsc.irfft(tX), x, 'c2r_fused_l2')):
                got = f().numpy()
                assert dsc.last_fft_path() == path
                scale = np.max(np.abs(want), axis=1, keepdims=True)
                worst = np.max(np.abs(got - want) / scale)
                assert worst <= 1e-12, (name, rows, rep, worst)
    dsc.synchronize()


@pytest.mark.parametrize('dt,L,path', [(np.float32, 32768, 'c2c_32k_regs'), (np.float32, 65536, 'c2c_fused_l2'), (np.float32, 131072, 'c2c_fused_l2'),
                                       (np.float32, 262144, 'c2c_2pass_regs'), (np.float64, 32768, 'c2c_fused_l2'), (np.float64, 131072, 'c2c_fused_l2'),
                                       (np.float64, 262144, 'c2c_2pass_regs')])
def test_fft_and_ifft_of_real_tensors_on_the_long_row_kernels(dsc, dt, L, path):
    """dsc_fft / dsc_ifft of a REAL tensor cast while gathering (dsc.cpp:1984-1988): on the long-row kernels the samples are widened
    on the way in (before round 2 these lengths fell to the generic four-step path at 7-9 % of the roofline).  Full, zero-padded
    and cropped rows against the oracle."""
    from oracle import port
    rng = np.random.default_rng(L + 3)
    rows = 3 if L >= 262144 else 9
    for ls in (L, L - 300, L + 40):
        x = rng.standard_normal((rows, ls)).astype(dt)
        F = dsc.fft(dsc.from_numpy(x), n=L)
        assert dsc.last_fft_path() == path, (dsc.last_fft_path(), path)
        assert_close(F.numpy(), port.fft(x, L), what=f'fft(real) {np.dtype(dt).name} L={L} ls={ls}')
        G = dsc.ifft(dsc.from_numpy(x), n=L)
        assert dsc.last_fft_path() == path
        assert_close(G.numpy(), port.ifft(x, L), what=f'ifft(real) {np.dtype(dt).name} L={L} ls={ls}')


def test_fused_l2_f32_every_element(dsc):
    """The f32 form of the team kernel (512-thread tasks; 65536- and 131072-point rows), every element of every row, repeatedly:
    the test that exposed a missing LDS barrier between rows (a wave starting the next row's exchange while another still read
    the packed-real partners of the current one) as a few hundred wrong bins in one row out of a few hundred."""
    rng = np.random.default_rng(32)
    for rows, L in ((120, 131072), (160, 65536)):
        z = (rng.standard_normal((rows, L)) + 1j * rng.standard_normal((rows, L))).astype(np.complex64)
        x = rng.standard_normal((rows, 2 * L)).astype(np.float32)
        wf = np.fft.fft(z.astype(np.complex128), axis=-1)
        wr = np.fft.rfft(x.astype(np.float64), axis=-1)
        Xc = wr.astype(np.complex64)
        wx = np.fft.irfft(Xc.astype(np.complex128), axis=-1)
        tz, tx, tX = dsc.from_numpy(z), dsc.from_numpy(x), dsc.from_numpy(Xc)
        for rep in range(3):
            for name, f, want, path in (('fft', lambda: dsc.fft(tz), wf, 'c2c_fused_l2'), ('rfft', lambda: dsc.rfft(tx), wr, 'r2c_fused_l2'),
                                        ('irfft', lambda: dsc.irfft(tX), wx, 'c2r_fused_l2')):
                got = f().numpy()
                assert dsc.last_fft_path() == path
                worst = float(np.max(np.max(np.abs(got - want), axis=1) / np.max(np.abs(want), axis=1)))
                assert worst <= 3e-5, (name, L, rep, worst)
    dsc.synchronize()


@pytest.mark.parametrize('dt,n', [(np.float32, 131072), (np.float32, 262144), (np.float64, 65536), (np.float64, 131072), (np.float64, 262144)])
def test_two_pass_padded_rows(dsc, dt, n):
    """Zero padded / cropped rows on the long-row kernels — every form of the team kernel (fft_xcd_fused.hip), which replaced the two-pass
    kernels at these lengths: the row descriptors end at the last valid sample / bin."""
    from oracle import port
    cdt = np.complex64 if dt == np.float32 else np.complex128
    rng = np.random.default_rng(n + 5)
    for rows, ls in ((3, n - 1), (2, n // 2 + 3), (2, n + 64)):
        x = rng.standard_normal((rows, ls)).astype(dt)
        got = dsc.rfft(dsc.from_numpy(x), n=n)
        assert dsc.last_fft_path() == 'r2c_fused_l2'           # every length of this test is on the team kernel by now
        assert_close(got.numpy()[rows - 1], port.rfft(x[rows - 1], n), what=f'padded 2-pass rfft n={n} ls={ls}')
    bins = n // 2 + 1
    for rows, lb in ((2, bins - 5), (3, bins + 9)):
        Y = (rng.standard_normal((rows, lb)) + 1j * rng.standard_normal((rows, lb))).astype(cdt)
        b = dsc.irfft(dsc.from_numpy(Y), n=bins)
        assert dsc.last_fft_path() == 'c2r_fused_l2'
        assert_close(b.numpy()[0], port.irfft(Y[0], bins), what=f'padded 2-pass irfft bins={bins} lb={lb}')


@pytest.mark.parametrize('dt,L', [(np.complex64, 65536), (np.complex64, 262144), (np.complex64, 524288), (np.complex64, 1048576), (np.complex128, 32768),
                                  (np.complex128, 131072), (np.complex128, 524288), (np.complex128, 1048576)])
def test_two_pass_complex_transforms(dsc, dt, L):
    """dsc_fft / dsc_ifft of complex rows longer than one CU's registers: the two-pass kernels without the real pass,
    full and zero-padded rows."""
    from oracle import port
    rng = np.random.default_rng(L)
    tol = 1e-6 if dt == np.complex64 else 1e-14
    for rows, ls in ((1, L), (3, L), (2, L - 77), (2, L + 5)):
        z = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(dt)
        Z = dsc.fft(dsc.from_numpy(z), n=L)
        path = 'c2c_fused_l2' if L <= 131072 else 'c2c_2pass_regs'
        assert dsc.last_fft_path() == path
        zh = Z.numpy()
        assert_close(zh[rows - 1], port.fft(z[rows - 1], L), what=f'fft L={L} ls={ls}')
        zp = np.zeros((rows, L), np.complex128)
        zp[:, :min(ls, L)] = z[:, :L]
        assert rel_l2(zh, np.fft.fft(zp, axis=-1)) <= tol
        back = dsc.ifft(Z)
        assert dsc.last_fft_path() == path
        assert_close(back.numpy()[0], port.ifft(zh[0]), what=f'ifft L={L}')
        assert rel_l2(back.numpy(), zp) <= tol


@pytest.mark.parametrize('shape,axis', [((1024, 37), 0), ((3, 4096, 5), 1), ((2, 2, 2048, 3), 2), ((131072, 3), 0)])
def test_strided_axes_via_transpose(dsc, shape, axis):
    """Transforms along a non-last axis: complex lengths 128 .. 4096 on the column kernel (lanes = neighbouring lines), longer
    ones as transpose -> register kernel -> transpose, shorter ones on the strided LDS kernel; every transform kind, padded
    and cropped, against the oracle.  The inner extents here (37, 5, 3) are narrower than a column tile: masked lanes."""
    from oracle import port
    rng = np.random.default_rng(shape[axis])
    n = shape[axis]
    x = rng.standard_normal(shape).astype(np.float32)
    for nn in (-1, n // 2, 2 * n):
        if 2 * n > 524288 and nn == 2 * n:
            continue
        got = dsc.rfft(dsc.from_numpy(x), n=nn, axis=axis)
        L = (1 << int(np.ceil(np.log2(nn if nn > 0 else n)))) // 2
        want_path = 'regs_cols' if 32 <= L <= 2048 else None
        assert (dsc.last_fft_path() == 'regs_cols') == (want_path == 'regs_cols'), (shape, nn, dsc.last_fft_path())
        assert (dsc.last_fft_path() in ('generic_lds', 'generic_4step')) == (L < 32), (shape, nn, dsc.last_fft_path())
        assert_close(got.numpy(), port.rfft(x, nn, axis), what=f'rfft {shape} axis {axis} n={nn}')
    X = port.rfft(x, -1, axis)
    assert_close(dsc.irfft(dsc.from_numpy(X), axis=axis).numpy(), port.irfft(X, -1, axis), what=f'irfft {shape} axis {axis}')
    z = (x + 1j * rng.standard_normal(shape)).astype(np.complex64)
    assert_close(dsc.fft(dsc.from_numpy(z), axis=axis).numpy(), port.fft(z, -1, axis), what=f'fft {shape} axis {axis}')
    assert_close(dsc.ifft(dsc.from_numpy(z), axis=axis).numpy(), port.ifft(z, -1, axis), what=f'ifft {shape} axis {axis}')
    xd = x.astype(np.float64)
    assert_close(dsc.rfft(dsc.from_numpy(xd), axis=axis).numpy(), port.rfft(xd, -1, axis), what=f'f64 rfft {shape} axis {axis}')


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('L', [32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_column_kernel_every_length_and_mode(dsc, dt, L):
    """fft_regs_cols.hip: every complex length it serves (f64: up to 2048), every mode, several column tiles with a ragged last
    one, a leading axis (slices), zero-padded and cropped axes — against the oracle (which is pinned on the reference)."""
    from oracle import port
    cdt = np.complex64 if dt == np.float32 else np.complex128
    rng = np.random.default_rng(L)
    inner = 300 if L <= 64 else 70 if L <= 1024 else 21   # > one tile of 256 / 128 / 64 / 32 / 16 / 8 columns, not a multiple
    on_cols = L <= 2048 or dt == np.float32               # complex data; the real modes stop at 2048
    on_cols_real = L <= 2048
    # complex transform of length L along axis 1 of [2, L, inner]
    z = (rng.standard_normal((2, L, inner)) + 1j * rng.standard_normal((2, L, inner))).astype(cdt)
    Z = dsc.fft(dsc.from_numpy(z), axis=1)
    # (full lines of 4096 points with 64 columns or more take the four-step route, two passes of this kernel at 64 points:
    # test_long_complex_transforms_along_a_non_last_axis; with the 21 columns here the one-pass form stays)
    assert (dsc.last_fft_path() == 'regs_cols') == on_cols, dsc.last_fft_path()
    assert_close(Z.numpy(), port.fft(z, -1, 1), what=f'fft L={L}')
    assert_close(dsc.ifft(dsc.from_numpy(z), axis=1).numpy(), port.ifft(z, -1, 1), what=f'ifft L={L}')
    zs = z[:, :L - 11]                                     # zero padded: axis shorter than the transform
    Zp = dsc.fft(dsc.from_numpy(np.ascontiguousarray(zs)), n=L, axis=1)
    assert (dsc.last_fft_path() == 'regs_cols') == on_cols, dsc.last_fft_path()
    assert_close(Zp.numpy(), port.fft(zs, L, 1), what=f'fft padded L={L}')
    # real transform of 2L points along axis 0 of [2L, inner]: full, zero padded (axis shorter than n), cropped (axis longer)
    x = rng.standard_normal((2 * L, inner)).astype(dt)
    X = dsc.rfft(dsc.from_numpy(x), axis=0)
    assert (dsc.last_fft_path() == 'regs_cols') == on_cols_real
    assert X.shape == (L + 1, inner)
    want = port.rfft(x, -1, 0)
    assert_close(X.numpy(), want, what=f'rfft L={L}')
    assert np.all(X.numpy()[0].imag == 0) and np.all(X.numpy()[-1].imag == 0)
    xs = x[:2 * L - 37]                                    # odd valid length: the last sample pair is cut
    assert_close(dsc.rfft(dsc.from_numpy(xs), n=2 * L, axis=0).numpy(), port.rfft(xs, 2 * L, 0), what=f'rfft padded L={L}')
    xl = np.concatenate([x, x[:50]])
    assert_close(dsc.rfft(dsc.from_numpy(xl), n=2 * L, axis=0).numpy(), port.rfft(xl, 2 * L, 0), what=f'rfft cropped L={L}')
    # inverse: full bins, fewer bins than L + 1 (missing ones are zero), imaginary parts of bins 0 and L ignored
    back = dsc.irfft(dsc.from_numpy(want), axis=0)
    assert (dsc.last_fft_path() == 'regs_cols') == on_cols_real
    assert_close(back.numpy(), port.irfft(want, -1, 0), what=f'irfft L={L}')
    assert rel_l2(back.numpy(), x) <= (1e-5 if dt == np.float32 else 1e-12)
    Xq = want.copy()
    Xq[0] += 3j
    Xq[-1] -= 2j
    assert_close(dsc.irfft(dsc.from_numpy(Xq), axis=0).numpy(), port.irfft(want, -1, 0), what=f'irfft ignores imag of bins 0, L (L={L})')
    fewer = want[:L // 2 + 3]
    assert_close(dsc.irfft(dsc.from_numpy(fewer), n=L + 1, axis=0).numpy(), port.irfft(fewer, L + 1, 0), what=f'irfft padded bins L={L}')


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('n', [512, 2048, 8192, 32768])
def test_fused_filter_mid_sizes(dsc, dt, n):
    """dsc_filter_fft (README.md:113-135 as one call) on the fused mid-size kernel: y = irfft(rfft(s, n) * H) against the
    oracle's three-operator composition, full rows, zero-padded rows (the README pads s inside rfft) and ragged row counts."""
    from oracle import port
    rng = np.random.default_rng(n)
    taps = np.zeros(n, dt)
    taps[:61] = rng.standard_normal(61)
    H = port.rfft(taps)
    tH = dsc.from_numpy(H)
    for rows, ls in ((1, n), (7, n), (5, n - 61), (3, n // 2 + 1), (2, n + 10)):
        s = rng.standard_normal((rows, ls)).astype(dt)
        y = dsc.filter_fft(dsc.from_numpy(s), tH)
        assert dsc.last_fft_path() == 'filter_mid_regs', (n, ls)
        assert y.shape == (rows, n)
        yh = y.numpy()
        for r in (0, rows - 1):
            want = port.irfft(port.mul(port.rfft(s[r], n), H))
            assert_close(yh[r], want, what=f'filter n={n} ls={ls} row {r}')
    # linear convolution check (the README's use): y[:ls + lb - 1] == convolve(s, b)
    s = rng.standard_normal((2, n - 61 + 1)).astype(dt)
    y = dsc.filter_fft(dsc.from_numpy(s), tH)[:, :n].numpy()
    want = np.stack([np.convolve(r.astype(np.float64), taps[:61].astype(np.float64)) for r in s])
    assert np.abs(y - want).max() <= (2e-3 if dt == np.float32 else 1e-10)


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_short_padded_and_cropped_lines_on_the_staged_register_kernel(dsc, dt):
    """Zero padded / cropped lines of 32 .. 256 complex points (frames of 200 samples transformed at 256, ...) stay on the LDS-staged
    register kernel, which gathers line by line (before: the generic kernel at 21-38 % of the roofline, now 60-77 %): every
    transform, odd lengths (a packed-real pair straddling the end of its line keeps the first sample only), many lines."""
    from oracle import port
    rng = np.random.default_rng(256)
    cdt = np.complex64 if dt == np.float32 else np.complex128
    for n in (64, 128, 256, 512):
        for ls in (n - 1, n - 37, 3, n + 5):
            for rows in (1, 700):
                x = rng.standard_normal((rows, ls)).astype(dt)
                X = dsc.rfft(dsc.from_numpy(x), n=n)
                assert dsc.last_fft_path() == 'regs_small', (n, ls, dsc.last_fft_path())
                assert_close(X.numpy(), port.rfft(x, n), what=f'rfft n={n} ls={ls} rows={rows}')
                if n <= 256:
                    z = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(cdt)
                    Z = dsc.fft(dsc.from_numpy(z), n=n)
                    assert dsc.last_fft_path() == 'regs_small'
                    assert_close(Z.numpy(), port.fft(z, n), what=f'fft n={n} ls={ls}')
                    assert_close(dsc.ifft(dsc.from_numpy(x), n=n).numpy(), port.ifft(x, n), what=f'ifft(real) n={n} ls={ls}')
                bins = n // 2 + 1
                lb = max(2, min(ls, bins + 3))
                Y = (rng.standard_normal((rows, lb)) + 1j * rng.standard_normal((rows, lb))).astype(cdt)
                b = dsc.irfft(dsc.from_numpy(Y), n=bins)
                assert dsc.last_fft_path() == 'regs_small'
                assert_close(b.numpy(), port.irfft(Y, bins), what=f'irfft bins={bins} lb={lb}')


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_fft_of_real_tensors_along_strided_axes(dsc, dt):
    """dsc_fft / dsc_ifft of a REAL tensor along a non-last axis (the cast of dsc.cpp:1984-1988 happens in the gather): the column kernel
    widens while loading (32 .. 2048 points); longer FULL lines take the four-step route of the same kernel (pass 1 widens), padded
    ones go through the transposes to the last-axis kernels, which widen too."""
    from oracle import port
    rng = np.random.default_rng(77)
    for n, want in ((64, 'regs_cols'), (1024, 'regs_cols'), (8192, 'regs_mid'), (65536, 'c2c_fused_l2')):
        cols = 24 if n >= 8192 else 120
        for ls in (n, n - 5):
            x = rng.standard_normal((ls, cols)).astype(dt)
            F = dsc.fft(dsc.from_numpy(x), n=n, axis=0)
            # full lines of 4096 points and more: the four-step route of the column kernel; padded ones keep the transposes
            assert dsc.last_fft_path() == ('cols_4step' if n >= 8192 and ls == n else want), (n, dsc.last_fft_path())
            assert_close(F.numpy(), port.fft(x, n, 0), what=f'fft(real) axis 0 n={n} ls={ls}')
            G = dsc.ifft(dsc.from_numpy(x), n=n, axis=0)
            assert_close(G.numpy(), port.ifft(x, n, 0), what=f'ifft(real) axis 0 n={n} ls={ls}')
    x3 = rng.standard_normal((5, 256, 37)).astype(dt)
    assert_close(dsc.fft(dsc.from_numpy(x3), axis=1).numpy(), port.fft(x3, -1, 1), what='fft(real) middle axis')
    assert dsc.last_fft_path() == 'regs_cols'


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_tiny_lengths_one_thread_per_line(dsc, dt):
    """Complex lengths 2 .. 16 (real 4 .. 32): fft_tiny.hip, one thread per line, the packed-real pass on the thread's own registers.
    Every transform, full / zero-padded / cropped lines, one line and many, against the oracle."""
    from oracle import port
    rng = np.random.default_rng(16)
    cdt = np.complex64 if dt == np.float32 else np.complex128
    for n in (4, 8, 16, 32):
        for ls in sorted({n, n - 1, 3, n + 5}):
            for rows in (1, 1500):
                x = rng.standard_normal((rows, ls)).astype(dt)
                X = dsc.rfft(dsc.from_numpy(x), n=n)
                assert dsc.last_fft_path() == 'regs_tiny', (n, ls, dsc.last_fft_path())
                assert_close(X.numpy(), port.rfft(x, n), what=f'rfft n={n} ls={ls} rows={rows}')
                bins = n // 2 + 1
                lb = max(2, min(ls, bins + 2))
                Y = (rng.standard_normal((rows, lb)) + 1j * rng.standard_normal((rows, lb))).astype(cdt)
                b = dsc.irfft(dsc.from_numpy(Y), n=bins)
                assert dsc.last_fft_path() == 'regs_tiny'
                assert_close(b.numpy(), port.irfft(Y, bins), what=f'irfft bins={bins} lb={lb}')
                if n <= 16:
                    z = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(cdt)
                    assert_close(dsc.fft(dsc.from_numpy(z), n=n).numpy(), port.fft(z, n), what=f'fft n={n} ls={ls}')
                    assert dsc.last_fft_path() == 'regs_tiny'
                    assert_close(dsc.ifft(dsc.from_numpy(z), n=n).numpy(), port.ifft(z, n), what=f'ifft n={n} ls={ls}')
                    assert_close(dsc.fft(dsc.from_numpy(x), n=n).numpy(), port.fft(x, n), what=f'fft(real) n={n} ls={ls}')
    # along a non-last axis: lane = column, no staging (fft_tiny_cols_kernel)
    for n, ls, cols in ((8, 8, 1000), (16, 13, 37), (32, 35, 5)):
        x = rng.standard_normal((ls, cols)).astype(dt)
        assert_close(dsc.rfft(dsc.from_numpy(x), n=n, axis=0).numpy(), port.rfft(x, n, 0), what=f'rfft axis 0 n={n}')
        assert dsc.last_fft_path() == 'regs_tiny_cols'
        z = (rng.standard_normal((ls, cols)) + 1j * rng.standard_normal((ls, cols))).astype(cdt)
        if n <= 16:
            assert_close(dsc.ifft(dsc.from_numpy(z), n=n, axis=0).numpy(), port.ifft(z, n, 0), what=f'ifft axis 0 n={n}')
            assert_close(dsc.fft(dsc.from_numpy(x), n=n, axis=0).numpy(), port.fft(x, n, 0), what=f'fft(real) axis 0 n={n}')
        bins = n // 2 + 1
        Y = z[:min(ls, bins)]
        assert_close(dsc.irfft(dsc.from_numpy(np.ascontiguousarray(Y)), n=bins, axis=0).numpy(), port.irfft(np.ascontiguousarray(Y), bins, 0), what=f'irfft axis 0 n={n}')
    x3 = rng.standard_normal((5, 16, 37)).astype(dt)
    assert_close(dsc.rfft(dsc.from_numpy(x3), axis=1).numpy(), port.rfft(x3, -1, 1), what='rfft middle axis')
    assert dsc.last_fft_path() == 'regs_tiny_cols'
    # a batch with leading dimensions
    x4 = rng.standard_normal((3, 5, 7, 16)).astype(dt)
    assert_close(dsc.rfft(dsc.from_numpy(x4)).numpy(), port.rfft(x4), what='rfft 4-d')
    assert dsc.last_fft_path() == 'regs_tiny'


def test_generic_four_step_beyond_the_two_pass_lengths(dsc):
    """Rows longer than the two-kernel route covers (complex length above 2^20) still go through the generic four-step path."""
    rng = np.random.default_rng(22)
    x = rng.standard_normal((1, 1 << 22)).astype(np.float32)
    X = dsc.rfft(dsc.from_numpy(x))
    assert dsc.last_fft_path() == 'generic_4step'
    assert rel_l2(X.numpy(), np.fft.rfft(x.astype(np.float64), axis=-1)) <= 2e-6
    assert rel_l2(dsc.irfft(X).numpy(), x) <= 2e-6


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_fft_of_real_tensors_beyond_262144_points_widen_then_two_pass(dsc, dt):
    """dsc_fft / dsc_ifft of REAL rows of 524288 and 1048576 points: widened into a complex temporary, then the complex two-pass
    route (the generic four-step path ran at 6 % of the roofline); full, zero-padded and cropped rows."""
    rng = np.random.default_rng(77)
    tol = 2e-6 if dt == np.float32 else 1e-12
    for L in (524288, 1048576):
        x = rng.standard_normal((3, L)).astype(dt)
        t = dsc.from_numpy(x)
        assert rel_l2(dsc.fft(t).numpy(), np.fft.fft(x.astype(np.float64), axis=-1)) <= tol and dsc.last_fft_path() == 'c2c_2pass_regs'
        assert rel_l2(dsc.ifft(t).numpy(), np.fft.ifft(x.astype(np.float64), axis=-1)) <= tol and dsc.last_fft_path() == 'c2c_2pass_regs'
    short = rng.standard_normal((2, 400000)).astype(dt)                     # n= pads to 524288
    assert rel_l2(dsc.fft(dsc.from_numpy(short), n=524288).numpy(), np.fft.fft(short.astype(np.float64), n=524288, axis=-1)) <= tol
    assert dsc.last_fft_path() == 'c2c_2pass_regs'
    long_ = rng.standard_normal((2, 600000)).astype(dt)                     # n= crops to 524288
    assert rel_l2(dsc.fft(dsc.from_numpy(long_), n=524288).numpy(), np.fft.fft(long_.astype(np.float64), n=524288, axis=-1)) <= tol


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_trailing_unit_dimensions_after_the_axis(dsc, dt):
    """x of shape [B, N, 1] (or [B, N, 1, 1]) transformed along axis 1: the lines are contiguous (inner == 1) although the axis is
    not the last slot.  Every contiguous-row route must count the lines of x and of out the same way — the route that widens
    real rows of 524288 points into a complex temporary once rewrote the slot of x only and wrote past the end of `out`."""
    rng = np.random.default_rng(5)
    cdt = np.complex64 if dt == np.float32 else np.complex128
    tol = 2e-6 if dt == np.float32 else 1e-12
    for n, want_path in ((64, None), (1024, None), (8192, None), (65536, None), (131072, None), (524288, 'c2c_2pass_regs')):
        for tail in ((1,), (1, 1)):
            x = rng.standard_normal((3, n) + tail).astype(dt)
            ref = np.fft.fft(x.astype(np.float64), axis=1)
            guard = dsc.from_numpy(np.full((3, n) + tail, 7 + 7j, cdt))     # allocated right behind `out` by the best-fit arena
            got = dsc.fft(dsc.from_numpy(x), axis=1)
            assert got.shape == x.shape and rel_l2(got.numpy(), ref) <= tol, (n, tail, dsc.last_fft_path())
            if want_path:
                assert dsc.last_fft_path() == want_path
            assert rel_l2(dsc.ifft(dsc.from_numpy(x), axis=1).numpy(), np.fft.ifft(x.astype(np.float64), axis=1)) <= tol
            assert rel_l2(dsc.rfft(dsc.from_numpy(x), axis=1).numpy(), np.fft.rfft(x.astype(np.float64), axis=1)) <= tol, (n, tail)
            X = np.fft.rfft(x.astype(np.float64), axis=1).astype(cdt)
            assert rel_l2(dsc.irfft(dsc.from_numpy(X), axis=1).numpy(), x) <= 2 * tol, (n, tail)
            assert np.all(guard.numpy() == 7 + 7j), (n, tail)
            del guard, got


@pytest.mark.parametrize('dt', [np.float32, np.float64])
def test_every_power_of_two_length(dsc, dt):
    """One sweep over every transform length 2 .. 2^20, all four transforms, full and zero-padded rows,
    against float64 numpy: whatever kernel path a length takes, the result must be within the precision's tolerance."""
    cdt = np.complex64 if dt == np.float32 else np.complex128
    tol = 2e-6 if dt == np.float32 else 2e-14
    rng = np.random.default_rng(11)
    paths = {}
    for p in range(1, 21):
        n = 1 << p
        rows = 3 if n <= 1 << 17 else 1
        for ls in (n, max(1, n - 3)):
            x = rng.standard_normal((rows, ls)).astype(dt)
            xp = np.zeros((rows, n), np.float64)
            xp[:, :ls] = x
            X = dsc.rfft(dsc.from_numpy(x), n=n)
            paths.setdefault(dsc.last_fft_path(), []).append(n)
            want = np.fft.rfft(xp, axis=-1)
            assert rel_l2(X.numpy(), want) <= tol, ('rfft', n, ls, dsc.last_fft_path())
            if n >= 2:
                back = dsc.irfft(X).numpy()
                assert back.shape == (rows, n) and rel_l2(back, xp) <= tol, ('irfft', n, ls, dsc.last_fft_path())
            z = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(cdt)
            zp = np.zeros((rows, n), np.complex128)
            zp[:, :ls] = z
            Z = dsc.fft(dsc.from_numpy(z), n=n)
            paths.setdefault(dsc.last_fft_path(), []).append(n)
            assert rel_l2(Z.numpy(), np.fft.fft(zp, axis=-1)) <= tol, ('fft', n, ls, dsc.last_fft_path())
            assert rel_l2(dsc.ifft(Z).numpy(), zp) <= tol, ('ifft', n, ls, dsc.last_fft_path())
    # every kernel family must have been exercised by the sweep
    need = {'generic_lds', 'regs_mid', 'r2c_2pass_regs', 'c2c_2pass_regs'} | ({'r2c_64k_regs'} if dt == np.float32 else set())
    assert need <= set(paths), paths.keys()


def test_shard_gather_through_the_c_collectives_one_rank():
    """dsc_amd/shard.py's 'allgather_c' and 'p2p_c' (the library's own RCCL entry points, include/dsc_mi355x.h section C) with a
    one-rank group: the transform writes its slot in place, the C all-gather runs for real, verify() passes
    (tools/check_c_collectives.py, its own process like the other torch.cuda users)."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_c_collectives.py')], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_rccl_sees_arena_memory_one_rank():
    """bench.py's all-gather phase hands RCCL zero-copy views of arena memory: with a one-rank NCCL (= RCCL) group on the
    test box's single GPU the view goes through all_gather_into_tensor, all_reduce and barrier and comes back intact
    (tools/check_rccl_view.py).  More ranks need more GPUs."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        free_port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_rccl_view.py')], capture_output=True, text=True, timeout=300,
                       cwd=root, env=env)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('n', [64, 128, 256, 512])
def test_small_size_register_path(dsc, dt, n):
    """Real lengths 64 .. 512 (complex 32 .. 256): the LDS-staged register kernel of fft_regs_mid.hip; all transforms, row
    counts that leave the last workgroup partially filled, against the oracle."""
    from oracle import port
    cdt = np.complex64 if dt == np.float32 else np.complex128
    rng = np.random.default_rng(n + 3)
    tol = 1e-6 if dt == np.float32 else 1e-14
    for rows in (1, 7, 300, 1031):
        x = rng.standard_normal((rows, n)).astype(dt)
        X = dsc.rfft(dsc.from_numpy(x))
        assert dsc.last_fft_path() == 'regs_small'
        got = X.numpy()
        for r in sorted({0, rows // 2, rows - 1}):
            assert_close(got[r], port.rfft(x[r]), what=f'rfft n={n} row {r}/{rows}')
        assert rel_l2(got, np.fft.rfft(x.astype(np.float64), axis=-1)) <= tol
        assert np.all(got[:, 0].imag == 0) and np.all(got[:, -1].imag == 0)
        Xq = got.copy()
        Xq[:, 0] += 2j
        Xq[:, -1] -= 3j
        back = dsc.irfft(dsc.from_numpy(Xq))
        assert dsc.last_fft_path() == 'regs_small'
        assert_close(back.numpy()[rows - 1], port.irfft(Xq[rows - 1]), what=f'irfft n={n}')
        assert rel_l2(back.numpy(), x) <= tol
        z = (rng.standard_normal((rows, n // 2)) + 1j * rng.standard_normal((rows, n // 2))).astype(cdt)
        Z = dsc.fft(dsc.from_numpy(z))
        assert dsc.last_fft_path() == 'regs_small'
        assert_close(Z.numpy()[rows - 1], port.fft(z[rows - 1]), what=f'fft n={n // 2}')
        assert rel_l2(dsc.ifft(Z).numpy(), z) <= tol
        xr = x[:, :n // 2].copy()
        assert_close(dsc.fft(dsc.from_numpy(xr)).numpy()[0], port.fft(xr[0]), what=f'fft(real) n={n // 2}')
        assert_close(dsc.ifft(dsc.from_numpy(xr)).numpy()[rows - 1], port.ifft(xr[rows - 1]), what=f'ifft(real) n={n // 2}')


def test_complex_32768_register_path(dsc):
    """dsc_fft / dsc_ifft of c32 rows of 32768 samples: the persistent register kernel of fft_c2c_32k.hip (full and
    zero-padded rows, ragged batch sizes)."""
    from oracle import port
    rng = np.random.default_rng(321)
    L = 32768
    for rows, ls in ((1, L), (3, L), (300, L), (2, L - 9), (2, L + 4)):
        z = (rng.standard_normal((rows, ls)) + 1j * rng.standard_normal((rows, ls))).astype(np.complex64)
        Z = dsc.fft(dsc.from_numpy(z), n=L)
        assert dsc.last_fft_path() == 'c2c_32k_regs'
        zh = Z.numpy()
        for r in sorted({0, rows - 1}):
            assert_close(zh[r], port.fft(z[r], L), what=f'fft 32768 ls={ls} row {r}')
        zp = np.zeros((rows, L), np.complex128)
        zp[:, :min(ls, L)] = z[:, :L]
        assert rel_l2(zh, np.fft.fft(zp, axis=-1)) <= 1e-6
        back = dsc.ifft(Z)
        assert dsc.last_fft_path() == 'c2c_32k_regs'
        assert_close(back.numpy()[rows - 1], port.ifft(zh[rows - 1]), what='ifft 32768')
        assert rel_l2(back.numpy(), zp) <= 1e-6
