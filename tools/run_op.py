#!/usr/bin/env python3
"""tools/run_op.py <case> [--iters N] — run ONE operator case of the hot path N times (for rocprofv3) and print one JSON line
with its algorithmic bytes and the HIP-event time per launch.  `--list` prints the case names.
Used by tools/profile_families.sh; the summaries land in profiles/."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

CASES = {}


def case(name):
    def deco(f):
        CASES[name] = f
        return f
    return deco


def _rfft_case(n, batch, f64=False, inverse=False):
    def make(dsc, B, ctx, np):
        rdt, cdt = (dsc.Dtype.F64, dsc.Dtype.C64) if f64 else (dsc.Dtype.F32, dsc.Dtype.C32)
        npr = np.float64 if f64 else np.float32
        rng = np.random.default_rng(3)
        blk = rng.standard_normal((min(batch, 64), n)).astype(npr)
        x = dsc.from_numpy(np.tile(blk, (batch // blk.shape[0], 1)))
        X = dsc.empty((batch, n // 2 + 1), cdt)
        B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, -1)
        es = 8 if f64 else 4
        nbytes = batch * (n * es + (n // 2 + 1) * 2 * es)
        if inverse:
            return (lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, -1)), nbytes, (x, X)
        return (lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, -1)), nbytes, (x, X)
    return make


def _fft_case(n, batch, f64=False, inverse=False):
    def make(dsc, B, ctx, np):
        cdt = dsc.Dtype.C64 if f64 else dsc.Dtype.C32
        npc = np.complex128 if f64 else np.complex64
        rng = np.random.default_rng(4)
        blk = (rng.standard_normal((min(batch, 64), n)) + 1j * rng.standard_normal((min(batch, 64), n))).astype(npc)
        x = dsc.from_numpy(np.tile(blk, (batch // blk.shape[0], 1)))
        X = dsc.empty((batch, n), cdt)
        es = 16 if f64 else 8
        if inverse:
            return (lambda: B.dsc_ifft(ctx, x._c_ptr, X._c_ptr, -1, -1)), 2 * batch * n * es, (x, X)
        return (lambda: B.dsc_fft(ctx, x._c_ptr, X._c_ptr, -1, -1)), 2 * batch * n * es, (x, X)
    return make


CASES['rfft64k'] = _rfft_case(65536, 8192)
CASES['irfft64k'] = _rfft_case(65536, 8192, inverse=True)
CASES['rfft_c5_f64_262144'] = _rfft_case(262144, 2048, f64=True)
CASES['irfft_c5_f64_262144'] = _rfft_case(262144, 2048, f64=True, inverse=True)
CASES['rfft_f32_131072'] = _rfft_case(131072, 4096)
CASES['rfft_f32_524288'] = _rfft_case(524288, 1024)
CASES['rfft_f32_1048576'] = _rfft_case(1048576, 512)
CASES['rfft_f64_1048576'] = _rfft_case(1048576, 256, f64=True)
CASES['fft_c32_524288'] = _fft_case(524288, 512)
CASES['rfft_f32_262144'] = _rfft_case(262144, 2048)
CASES['irfft_f32_262144'] = _rfft_case(262144, 2048, inverse=True)
CASES['fft_c32_131072'] = _fft_case(131072, 2048)
CASES['irfft_f32_131072'] = _rfft_case(131072, 4096, inverse=True)
CASES['irfft_f32_524288'] = _rfft_case(524288, 1024, inverse=True)
CASES['rfft_f32_1024'] = _rfft_case(1024, 524288)
CASES['irfft_f32_1024'] = _rfft_case(1024, 524288, inverse=True)
CASES['rfft_f32_4096'] = _rfft_case(4096, 131072)
CASES['irfft_f32_4096'] = _rfft_case(4096, 131072, inverse=True)
CASES['rfft_f32_16384'] = _rfft_case(16384, 32768)
CASES['rfft_f32_256'] = _rfft_case(256, 2097152)
CASES['rfft_f64_4096'] = _rfft_case(4096, 65536, f64=True)
CASES['rfft_f32_32'] = _rfft_case(32, 16777216)          # one thread per line (fft_tiny.hip)
CASES['rfft_f32_8'] = _rfft_case(8, 67108864)
CASES['fft_c32_32768'] = _fft_case(32768, 8192)
CASES['fft_c32_4096'] = _fft_case(4096, 65536)
CASES['fft_c32_65536'] = _fft_case(65536, 4096)
CASES['fft_c64_65536'] = _fft_case(65536, 2048, f64=True)
CASES['fft_c64_131072'] = _fft_case(131072, 2048, f64=True)
CASES['ifft_c64_131072'] = _fft_case(131072, 2048, f64=True, inverse=True)
CASES['ifft_c32_65536'] = _fft_case(65536, 4096, inverse=True)
CASES['rfft_f64_131072'] = _rfft_case(131072, 2048, f64=True)
CASES['rfft_f64_65536'] = _rfft_case(65536, 4096, f64=True)
CASES['irfft_f64_65536'] = _rfft_case(65536, 4096, f64=True, inverse=True)
CASES['fft_c64_32768'] = _fft_case(32768, 4096, f64=True)
CASES['rfft_f64_32768'] = _rfft_case(32768, 8192, f64=True)      # the persistent f64 lines of 16384 points (round 3)
CASES['irfft_f64_32768'] = _rfft_case(32768, 8192, f64=True, inverse=True)
CASES['fft_c64_16384'] = _fft_case(16384, 8192, f64=True)
CASES['irfft_f32_32768'] = _rfft_case(32768, 16384, inverse=True)
CASES['irfft_f64_131072'] = _rfft_case(131072, 2048, f64=True, inverse=True)


@case('filter64k')
def _filter(dsc, B, ctx, np):
    rng = np.random.default_rng(5)
    blk = rng.standard_normal((64, 65536)).astype(np.float32)
    s = dsc.from_numpy(np.tile(blk, (64, 1)))
    y = dsc.empty((4096, 65536), dsc.Dtype.F32)
    H = dsc.from_numpy((rng.standard_normal(32769) + 1j * rng.standard_normal(32769)).astype(np.complex64))
    return (lambda: B.dsc_filter_fft(ctx, s._c_ptr, H._c_ptr, y._c_ptr)), 4096 * 65536 * 8, (s, y, H)


@case('filter_f32_4096')
def _filter_mid(dsc, B, ctx, np):
    rng = np.random.default_rng(5)
    n, batch = 4096, 65536
    blk = rng.standard_normal((64, n)).astype(np.float32)
    s = dsc.from_numpy(np.tile(blk, (batch // 64, 1)))
    y = dsc.empty((batch, n), dsc.Dtype.F32)
    H = dsc.from_numpy((rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)).astype(np.complex64))
    return (lambda: B.dsc_filter_fft(ctx, s._c_ptr, H._c_ptr, y._c_ptr)), batch * n * 8, (s, y, H)


def _filter_mid_case(n, batch):
    def make(dsc, B, ctx, np):
        rng = np.random.default_rng(5)
        blk = rng.standard_normal((64, n)).astype(np.float32)
        s = dsc.from_numpy(np.tile(blk, (batch // 64, 1)))
        y = dsc.empty((batch, n), dsc.Dtype.F32)
        H = dsc.from_numpy((rng.standard_normal(n // 2 + 1) + 1j * rng.standard_normal(n // 2 + 1)).astype(np.complex64))
        return (lambda: B.dsc_filter_fft(ctx, s._c_ptr, H._c_ptr, y._c_ptr)), batch * n * 8, (s, y, H)
    return make


CASES['filter_f32_32768'] = _filter_mid_case(32768, 8192)
CASES['filter_f32_16384'] = _filter_mid_case(16384, 16384)


@case('rfft_axis0_4096x8192')
def _axis0(dsc, B, ctx, np):
    rng = np.random.default_rng(6)
    n, cols = 4096, 8192
    x = dsc.from_numpy(rng.standard_normal((n, cols)).astype(np.float32))
    X = dsc.empty((n // 2 + 1, cols), dsc.Dtype.C32)
    return (lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0)), cols * (n * 4 + (n // 2 + 1) * 8), (x, X)


@case('rfft_axis0_256x131072')
def _axis0_small(dsc, B, ctx, np):
    rng = np.random.default_rng(6)
    n, cols = 256, 131072
    x = dsc.from_numpy(rng.standard_normal((n, cols)).astype(np.float32))
    X = dsc.empty((n // 2 + 1, cols), dsc.Dtype.C32)
    return (lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0)), cols * (n * 4 + (n // 2 + 1) * 8), (x, X)


def _axis0_case(n, cols, kind):
    def make(dsc, B, ctx, np):
        rng = np.random.default_rng(6)
        blk = rng.standard_normal((n, 4096)).astype(np.float32)
        if kind == 'fft':
            zc = cols // 2
            z = dsc.from_numpy(np.tile((blk[:, :2048] + 1j * blk[:, 2048:]).astype(np.complex64), (1, zc // 2048)))
            Z = dsc.empty((n, zc), dsc.Dtype.C32)
            return (lambda: B.dsc_fft(ctx, z._c_ptr, Z._c_ptr, -1, 0)), 2 * n * zc * 8, (z, Z)
        x = dsc.from_numpy(np.tile(blk, (1, cols // 4096)))
        X = dsc.empty((n // 2 + 1, cols), dsc.Dtype.C32)
        B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0)
        nbytes = n * cols * 4 + (n // 2 + 1) * cols * 8
        if kind == 'irfft':
            return (lambda: B.dsc_irfft(ctx, X._c_ptr, x._c_ptr, -1, 0)), nbytes, (x, X)
        return (lambda: B.dsc_rfft(ctx, x._c_ptr, X._c_ptr, -1, 0)), nbytes, (x, X)
    return make


# 8192 and 65536 points along axis 0: the four-step routes of the column kernel (round 3)
for _n, _cols in ((256, 1 << 20), (1024, 1 << 18), (2048, 1 << 17), (8192, 1 << 15), (65536, 1 << 12)):
    for _kind in ('rfft', 'irfft', 'fft'):
        CASES[f'{_kind}_axis0_{_n}x{_cols if _kind != "fft" else _cols // 2}'] = _axis0_case(_n, _cols, _kind)


@case('mul_c32_bcast')
def _mul(dsc, B, ctx, np):
    S = dsc.empty((4096, 32769), dsc.Dtype.C32)
    P = dsc.empty((4096, 32769), dsc.Dtype.C32)
    H = dsc.from_numpy(np.ones(32769, np.complex64))
    return (lambda: B.dsc_mul(ctx, S._c_ptr, H._c_ptr, P._c_ptr)), 4096 * 32769 * 16, (S, P, H)


@case('sum_c32_axis0')
def _sum0(dsc, B, ctx, np):
    S = dsc.empty((4096, 32769), dsc.Dtype.C32)
    o = dsc.empty((1, 32769), dsc.Dtype.C32)
    return (lambda: B.dsc_sum(ctx, S._c_ptr, o._c_ptr, 0, True)), 4096 * 32769 * 8, (S, o)


@case('sum_c32_axis1')
def _sum1(dsc, B, ctx, np):
    S = dsc.empty((4096, 32769), dsc.Dtype.C32)
    o = dsc.empty((4096, 1), dsc.Dtype.C32)
    return (lambda: B.dsc_sum(ctx, S._c_ptr, o._c_ptr, 1, True)), 4096 * 32769 * 8, (S, o)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('case', nargs='?')
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--ramp-ms', type=float, default=100.0, help='untimed launches for this long before the warm-up, as bench.py does (0 disables)')
    ap.add_argument('--list', action='store_true')
    args = ap.parse_args()
    if args.list:
        print(' '.join(CASES))
        return
    fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import dsc_amd as dsc
    from dsc_amd import _bindings as B
    from dsc_amd.context import _get_ctx
    dsc.init(20 << 30, 6 << 30)
    ctx = _get_ctx()
    f, nbytes, keep = CASES[args.case](dsc, B, ctx, np)
    # the same untimed clock ramp as bench.py (the uploads above left the GPU idle): --ramp-ms of launches, then the warm-up
    import time
    t_end = time.perf_counter() + args.ramp_ms / 1e3
    while time.perf_counter() < t_end:
        for _ in range(5):
            f()
        dsc.synchronize()
    for _ in range(max(5, args.iters // 3)):
        f()
    dsc.synchronize()
    best = 1e30
    for _ in range(3):
        B.dsc_timer_start(ctx)
        for _ in range(args.iters):
            f()
        best = min(best, B.dsc_timer_stop(ctx) / args.iters)
    path = dsc.last_fft_path()
    sys.stdout.flush()
    os.dup2(fd, 1)
    print(json.dumps({'case': args.case, 'algorithmic_bytes': nbytes, 'hip_event_ms': round(best, 4), 'path': path,
                      'frac_of_8TBps': round(nbytes / best / 1e6 / 8000, 4)}), flush=True)


if __name__ == '__main__':
    main()
