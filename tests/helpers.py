"""Shared test helpers: deterministic signals, golden-fixture access, error metrics."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden')
NP = {'f32': np.float32, 'f64': np.float64, 'c32': np.complex64, 'c64': np.complex128}

# Tolerances of BASELINE.json's north_star ("within 1e-5 rel of CPU reference"; 1e-12 for
# the f64 config) — relative L2 error against the reference/oracle output.
TOL = {np.dtype(np.float32): 1e-5, np.dtype(np.complex64): 1e-5,
       np.dtype(np.float64): 1e-12, np.dtype(np.complex128): 1e-12}


def lcg_signal(shape, seed, dtype):
    """Uniform(-1, 1) samples from a 64-bit LCG (Knuth MMIX constants), top 24 bits.

    Pure integer recurrence: bit-reproducible on any numpy, so fixtures for large inputs
    store only this spec and the reference's output."""
    n = int(np.prod(shape))
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    lanes = 4096            # 4096 independent streams, interleaved
    state = np.arange(lanes, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    out = np.empty((-(-n // lanes), lanes), dtype=np.float64)
    with np.errstate(over='ignore'):
        for i in range(out.shape[0]):
            state = state * a + c
            out[i] = (state >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0
    flat = out.reshape(-1)[:n]
    if np.dtype(dtype).kind == 'c':
        return (flat + 1j * (np.roll(flat, 1) * 0.5)).reshape(shape).astype(dtype)
    return flat.reshape(shape).astype(dtype)


def rel_l2(a, b):
    a = np.asarray(a).ravel()
    b = np.asarray(b).ravel()
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


def max_rel(a, b):
    """max |a-b| / max |b| — the element-wise form of SURVEY 8c's tolerance."""
    b = np.asarray(b)
    m = np.max(np.abs(b)) if b.size else 0.0
    d = np.max(np.abs(np.asarray(a) - b)) if b.size else 0.0
    return float(d / m) if m > 0 else float(d)


def assert_close(actual, expected, tol=None, what=''):
    actual = np.asarray(actual)
    expected = np.asarray(expected)
    assert actual.shape == expected.shape, f'{what}: shape {actual.shape} != {expected.shape}'
    assert actual.dtype == expected.dtype, f'{what}: dtype {actual.dtype} != {expected.dtype}'
    tol = TOL[expected.dtype] if tol is None else tol
    e2, em = rel_l2(actual, expected), max_rel(actual, expected)
    assert e2 <= tol and em <= tol * 4, f'{what}: rel_l2={e2:.3e} max_rel={em:.3e} tol={tol:g}'


class Golden:
    """Committed fixtures: outputs of the reference itself (tests/golden/make_golden.py)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, 'manifest.json')) as f:
            self.manifest = json.load(f)
        self._npz = {}

    def group(self, name):
        if name not in self._npz:
            self._npz[name] = np.load(os.path.join(GOLDEN, f'{name}.npz'))
        return self._npz[name]

    def cases(self, group, op=None):
        prev_y = None
        for rec in self.manifest:
            if rec['group'] != group:
                continue
            g = self.group(group)
            y = g[rec['key'] + '_y']
            if 'gen' in rec:
                spec = rec['gen']
                if spec['kind'] == 'prev_output':
                    xs = [prev_y]
                else:
                    xs = [lcg_signal(spec['shape'], spec['seed'], NP[spec['dtype']])]
                    if rec['key'] + '_x1' in g:
                        xs.append(g[rec['key'] + '_x1'])
            else:
                xs = [g[f"{rec['key']}_x{j}"] for j in range(rec['n_in'])]
            prev_y = y
            if op is None or rec['op'] == op:
                yield rec, xs, y


def decode_sel(sel):
    """`sel` of a 'slice' golden record -> tuple of ints / slices (tests/golden/make_golden.py)."""
    return tuple(slice(*k) if isinstance(k, list) else int(k) for k in sel)
