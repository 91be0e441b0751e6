#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference itself.

Runs only where oracle/_ref/libdsc_ref.so exists (built by `make -C oracle ref` from
the sources under /root/reference, i.e. in the build container).  The outputs below are
what dspcraft/dsc's own CPU code returns for the stored inputs; they are data, not code.
The reference ships no golden vectors (python/tests/test_ops.py compares against numpy
on unseeded random input), so these fixtures are the pin.

    python tests/golden/make_golden.py          # rewrites the .npz files + manifest.json

Large inputs are produced by `tests/helpers.py:lcg_signal`, an integer recurrence that is bit-reproducible
on any numpy, so only their outputs are stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))

from oracle import ref  # noqa: E402
from tests.helpers import lcg_signal  # noqa: E402

NP = {'f32': np.float32, 'f64': np.float64, 'c32': np.complex64, 'c64': np.complex128}


def rnd(rng, shape, dtype):
    x = rng.standard_normal(shape)
    if np.dtype(dtype).kind == 'c':
        x = x + 1j * rng.standard_normal(shape)
    return x.astype(dtype)


def main():
    R = ref.Ref.get()
    rng = np.random.default_rng(20241004)
    manifest = []
    arrays = {}

    def add(group, op, x, y, **kw):
        i = len([m for m in manifest if m['group'] == group])
        key = f'{op}_{i}'
        rec = dict(group=group, key=key, op=op, **kw)
        g = arrays.setdefault(group, {})
        if isinstance(x, dict):           # generator spec instead of stored input
            rec['gen'] = x
        elif isinstance(x, tuple):
            for j, xi in enumerate(x):
                g[f'{key}_x{j}'] = xi
            rec['n_in'] = len(x)
        else:
            g[f'{key}_x0'] = x
            rec['n_in'] = 1
        g[f'{key}_y'] = y
        manifest.append(rec)

    # ---- rfft / irfft / fft / ifft, small sizes, every layout rule of SURVEY 8c -----
    for rdt, cdt in (('f32', 'c32'), ('f64', 'c64')):
        for N in (2, 4, 8, 16, 64, 1024, 4096):
            for shape, axis in (((N,), -1), ((3, N), -1), ((2, 3, N), -1), ((N, 3), 0)):
                if N >= 1024 and (len(shape) == 3 or shape[0] == N and len(shape) == 2):
                    continue
                x = rnd(rng, shape, NP[rdt])
                X = R.rfft(x, -1, axis)
                add('fft_small', 'rfft', x, X, n=-1, axis=axis)
                add('fft_small', 'irfft', X, R.irfft(X, -1, axis), n=-1, axis=axis)
        # pad / crop through n=, non power of two lengths and n
        for (shape, axis, n) in (((3, 64), -1, 32), ((3, 64), -1, 128), ((1000,), -1, -1),
                                 ((2, 1000), -1, 600), ((5, 48, 2), 1, -1), ((2, 2, 2, 16), -1, -1),
                                 ((4, 4, 16, 4), 2, 8), ((16, 4, 4, 4), 0, 32)):
            x = rnd(rng, shape, NP[rdt])
            add('fft_small', 'rfft', x, R.rfft(x, n, axis), n=n, axis=axis)
            xc = rnd(rng, shape, NP[cdt])
            add('fft_small', 'fft', xc, R.fft(xc, n, axis), n=n, axis=axis)
            add('fft_small', 'ifft', xc, R.ifft(xc, n, axis), n=n, axis=axis)
            add('fft_small', 'fft', x, R.fft(x, n, axis), n=n, axis=axis)      # real in -> complex
        # irfft(n=) is a BIN count in the reference (dsc.cpp:2199-2200)
        X = rnd(rng, (2, 513), NP[cdt])
        for n in (-1, 513, 257, 1024, 100):
            add('fft_small', 'irfft', X, R.irfft(X, n, -1), n=n, axis=-1)
        for N in (8, 1024):
            xc = rnd(rng, (2, N), NP[cdt])
            add('fft_small', 'fft', xc, R.fft(xc), n=-1, axis=-1)
            add('fft_small', 'ifft', xc, R.ifft(xc), n=-1, axis=-1)
        # known-answer signals: impulse, constant, single tone, Nyquist tone
        N = 256
        t = np.arange(N)
        for name, sig in (('impulse', (t == 0) * 1.0), ('impulse3', (t == 3) * 1.0), ('const', np.ones(N)),
                          ('tone5', np.cos(2 * np.pi * 5 * t / N)), ('nyquist', np.cos(np.pi * t))):
            x = sig.astype(NP[rdt])
            add('fft_small', 'rfft', x, R.rfft(x), n=-1, axis=-1, kat=name)

    # ---- headline sizes: generated input, stored output --------------------------------
    spec = dict(kind='lcg', shape=[2, 65536], seed=11, dtype='f32')
    x = lcg_signal(spec['shape'], spec['seed'], NP[spec['dtype']])
    X = R.rfft(x)
    add('fft_large', 'rfft', spec, X, n=-1, axis=-1)
    add('fft_large', 'irfft', dict(kind='prev_output'), R.irfft(X), n=-1, axis=-1)
    spec = dict(kind='lcg', shape=[1, 65536], seed=12, dtype='c32')
    xc = lcg_signal(spec['shape'], spec['seed'], NP[spec['dtype']])
    add('fft_large', 'fft', spec, R.fft(xc), n=-1, axis=-1)
    spec = dict(kind='lcg', shape=[1, 262144], seed=13, dtype='f64')
    x = lcg_signal(spec['shape'], spec['seed'], NP[spec['dtype']])
    add('fft_large', 'rfft', spec, R.rfft(x), n=-1, axis=-1)

    # ---- mul: equal shapes, row broadcast, scalar, promotion (F64 x C32 -> C32) ----------
    for da, db in (('c32', 'c32'), ('c64', 'c64'), ('f32', 'f32'), ('f64', 'c32'), ('c32', 'f64'),
                   ('f32', 'c64'), ('f64', 'f32')):
        for sa, sb in (((6, 33), (6, 33)), ((6, 33), (33,)), ((6, 33), (1,)), ((1,), (6, 33)),
                       ((3, 1, 5), (1, 4, 1)), ((2, 3, 4, 5), (4, 1))):
            a, b = rnd(rng, sa, NP[da]), rnd(rng, sb, NP[db])
            add('mul', 'mul', (a, b), R.mul(a, b))

    # ---- reductions: all axes x keepdims, all dtypes; ties on the real part for max/min --
    for dt in ('f32', 'f64', 'c32', 'c64'):
        for shape in ((7, 5, 3), (2, 3, 4, 5), (9,)):
            for axis in range(-len(shape), len(shape)):
                for keep in (True, False):
                    if len(shape) == 1 and not keep:
                        continue
                    for op, name in enumerate(('sum', 'mean', 'max', 'min')):
                        x = rnd(rng, shape, NP[dt])
                        if op >= 2:       # quantise the real part so that ties occur
                            q = np.round(x.real * 2) / 2
                            x = (q + 1j * x.imag).astype(NP[dt]) if np.dtype(NP[dt]).kind == 'c' else q.astype(NP[dt])
                        add('reduce', name, x, R.reduce(x, op, axis, keep), axis=axis, keepdims=keep)

    # ---- README filterFFT pipeline (README.md:113-135): rfft, rfft, mul, irfft ------------
    for ls, lb in ((1000, 31), (65000, 537)):
        if ls > 5000:
            spec = dict(kind='lcg', shape=[ls], seed=21, dtype='f32')
            s = lcg_signal(spec['shape'], spec['seed'], np.float32)
        else:
            s = rnd(rng, (ls,), np.float32)
        tt = np.arange(lb) - (lb - 1) / 2
        b = (np.sinc(0.2 * tt) * np.hamming(lb) * 0.2).astype(np.float32)
        n = int(2 ** np.ceil(np.log2(ls + lb - 1)))
        S, B = R.rfft(s, n), R.rfft(b, n)
        y = R.irfft(R.mul(S, B))
        if ls > 5000:
            add('filter', 'filter', dict(spec, taps=lb), y, n=n)
            arrays['filter'][f'filter_{len([m for m in manifest if m["group"] == "filter"]) - 1}_x1'] = b
        else:
            add('filter', 'filter', (s, b), y, n=n)

    # ---- add / sub / div: same skeleton as mul; appended last so that the
    # random stream of the groups above is unchanged
    for op, name in ((0, 'add'), (1, 'sub'), (3, 'div')):
        for da, db in (('c32', 'c32'), ('f32', 'f32'), ('f64', 'c32'), ('c64', 'f32')):
            for sa, sb in (((6, 33), (6, 33)), ((6, 33), (33,)), ((6, 33), (1,)), ((1,), (6, 33)), ((3, 1, 5), (1, 4, 1))):
                a, b = rnd(rng, sa, NP[da]), rnd(rng, sb, NP[db])
                add('binary', name, (a, b), R.binary(a, b, op))

    # ---- abs / angle / conj / real / imag of spectra and of real tensors
    for dt in ('f32', 'f64', 'c32', 'c64'):
        for shape in ((9,), (6, 33)):
            x = rnd(rng, shape, NP[dt])
            for op, name in enumerate(('abs', 'angle', 'conj', 'real', 'imag')):
                add('unary', name, x, R.unary(x, op))

    # ---- indexing / slicing: `sel` is the key, ints or [start, stop, step] with null = None
    def enc(k):
        return [None if v is None else int(v) for v in (k.start, k.stop, k.step)] if isinstance(k, slice) else int(k)

    S = slice
    for dt in ('f32', 'c32', 'f64', 'c64'):
        x = rnd(rng, (3, 4, 5, 6), NP[dt])
        x2 = rnd(rng, (7, 40), NP[dt])
        for k in ((S(None),), (1,), (S(None), 2), (S(1, 3), S(None, None, -1), S(0, 5, 2), S(-4, -1)), (-1, S(None), -2, S(None, None, 3)),
                  (S(2, 0, -1),), (S(None, None, -2), S(None), S(None), S(5, None, -2)), (S(None), S(None), S(None), 0)):
            add('slice', 'get_slice', x, R.get_slice(x, *k), sel=[enc(i) for i in k])
        for k in ((S(None), S(None, 17)), (S(None), S(3, None, 4)), (S(None, None, -1), S(None)), (-1, S(None, None, -1))):
            add('slice', 'get_slice', x2, R.get_slice(x2, *k), sel=[enc(i) for i in k])
        for idx in ((0,), (2, 3), (-1, -1, -1), (1, 2, 3, 4)):
            add('slice', 'get_idx', x, R.get_idx(x, *idx), sel=list(idx))
        one = rnd(rng, (1,), NP[dt])
        for k, v in (((S(0, 2), 1), rnd(rng, (2, 1, 5, 6), NP[dt])), ((S(None), S(None), S(None), S(0, 3)), one),
                     ((1, S(None, None, 2)), rnd(rng, (1, 2, 5, 6), NP[dt])), ((S(None), S(None), S(1, 4), S(None, None, -1)), rnd(rng, (3,), NP[dt])),
                     ((S(None, None, -1),), rnd(rng, (3, 4, 5, 6), NP[dt]))):
            add('slice', 'set_slice', (x, v), R.set_slice(x, v, *k), sel=[enc(i) for i in k])
        for idx, v in (((1, 2, 3, 4), one), ((2,), one), ((0, 1), one), ((1,), rnd(rng, (3, 4, 5), NP[dt]))):
            add('slice', 'set_idx', (x, v), R.set_idx(x, v, *idx), sel=list(idx))

    # ---- transpose / fftfreq / rfftfreq
    for dt in ('f32', 'c32', 'f64', 'c64'):
        for shape, axes in (((5, 7), None), ((40, 70), (1, 0)), ((2, 3, 4), None), ((2, 3, 4), (0, 2, 1)), ((2, 3, 4), (1, 2, 0)),
                            ((2, 3, 4, 5), None), ((2, 3, 4, 5), (0, 1, 3, 2)), ((2, 3, 4, 5), (3, 0, 2, 1)), ((9,), None)):
            x = rnd(rng, shape, NP[dt])
            add('layout', 'transpose', x, R.transpose(x, axes), axes=list(axes) if axes else None)
    for dt in ('f32', 'f64'):
        for n, d in ((8, 1.0), (9, 0.5), (1024, 1.0 / 44100), (1000, 0.001), (1, 2.0)):
            add('layout', 'fftfreq', np.zeros(1, NP[dt]), R.fftfreq(n, d, NP[dt]), n=n, d=d)
            add('layout', 'rfftfreq', np.zeros(1, NP[dt]), R.fftfreq(n, d, NP[dt], True), n=n, d=d)

    for group, g in arrays.items():
        np.savez_compressed(os.path.join(HERE, f'{group}.npz'), **g)
        print(group, len(g), 'arrays', os.path.getsize(os.path.join(HERE, f'{group}.npz')) // 1024, 'KiB')
    with open(os.path.join(HERE, 'manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=0)
    print(len(manifest), 'cases')


if __name__ == '__main__':
    main()
